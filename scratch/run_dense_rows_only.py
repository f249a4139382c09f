"""Ten 672 -> 256 Linears over a slot-major buffer of N rows (argv[1], default 100000) with the rows kernel: the process PMC passes run on."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import ops
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
mid = torch.randn(21, N, 32, device=dev); w = torch.randn(256, 672, device=dev) * 0.05; b = torch.randn(256, device=dev)
parts = ops.new_partials(256, dev)
for _ in range(10):
    ops.dense_act(mid, w, b, 1, slot_major=True, f16_split=True, partials=parts)
torch.cuda.synchronize()
