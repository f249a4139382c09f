"""The bf16 width-64 NNConv (BASELINE config 3) over edge groups and over type columns, alone: 100 000 nodes / 1 M edges, T = 13;
run under rocprofv3 --kernel-trace --stats."""
import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops, ops_bf16
from tilingnn_amd.synth import make_super_graph
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dev = torch.device('cuda:0')
sg = make_super_graph(n, 10 * n, 12 * n, tile_count=4, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
g = ops.prepare_graph(n, adj, attr, col)
h = torch.randn(n, 64, device=dev).to(torch.bfloat16)
wtab = torch.rand(g.n_types, 64, 64, device=dev)
root = torch.randn(64, 64, device=dev) * 0.3
bias = torch.randn(64, device=dev)
for k in ("cols", "eg"):
    for _ in range(50):
        ops_bf16.nnconv64(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, ops.new_partials(64, dev), kernel=k)
torch.cuda.synchronize()
