"""Where do the rows kernel's outputs differ from the block-tile kernel's / fp64?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
for N in (4099, 100_000, 300_000, 300_000):
    mid = torch.randn(21, N, 32, device=dev); w = torch.randn(256, 672, device=dev) * 0.05; b = torch.randn(256, device=dev)
    parts = ops.new_partials(256, dev)
    o1, n1 = ops.dense_act(mid, w, b, 1, slot_major=True, f16_split="tile", partials=parts)
    for rep in range(3):
        o2, n2 = ops.dense_act(mid, w, b, 1, slot_major=True, f16_split=True, partials=parts)
        bad = ~torch.isfinite(o2)
        d = (o1 - o2).abs()
        d[bad] = 1e30
        rows = (d > 1e-3 * o1.abs().max()).any(1).nonzero().flatten()
        cols = (d > 1e-3 * o1.abs().max()).any(0).nonzero().flatten()
        print(N, rep, "nonfinite", int(bad.sum()), "tile nonfinite", int((~torch.isfinite(o1)).sum()), "bad rows", rows.numel(), rows[:12].tolist(),
              "bad cols", cols.numel(), cols[:12].tolist(), "max diff", float(d[~bad].max()))
