"""Isolated NNConv column kernel at 100k nodes vs the number of CUs it is given (TGNN_RESERVE_CUS)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import ops
from tilingnn_amd.synth import make_super_graph
n = 100000
sg = make_super_graph(n, 10 * n, int(12.5 * n), tile_count=2, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch("cuda:0")
g = ops.prepare_graph(n, adj, attr, col)
h = torch.randn(n, 32, device="cuda")
wtab = torch.randn(g.n_types, 32, 32, device="cuda") * 0.1
root, bias = torch.randn(32, 32, device="cuda") * 0.1, torch.randn(32, device="cuda")
parts = ops.new_partials(32, "cuda:0")
f = lambda: ops.nnconv_mean(h, g, wtab, root, bias, act=1, partials=parts)
for _ in range(5): f()
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("TGNN_RESERVE_CUS"), "us per call (image + kernel):", e0.elapsed_time(e1) / 50 * 1e3)
