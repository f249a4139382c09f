import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops
dev = torch.device('cuda:0')
N = 100_000
mid = torch.randn(21, N, 32, device=dev); w = torch.randn(256, 672, device=dev) * 0.05; b = torch.randn(256, device=dev)
parts = ops.new_partials(256, dev)
for _ in range(5): ops.dense_act(mid, w, b, 1, partials=parts, slot_major=True)
torch.cuda.synchronize()
