"""Runs only NNConv launches of the LDS-streaming kernel (for PMC passes): N=100k/Ea=1M synthetic graph. argv[1]: ps | cols_f16"""
import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops
from tilingnn_amd.synth import make_super_graph
dev = torch.device('cuda:0')
n = 100_000
sg = make_super_graph(n, 1_000_000, 1_250_000, tile_count=2, n_edge_types=13, seed=2)
x, adj, adj_attr, col, _ = sg.to_torch(dev)
g = ops.prepare_graph(n, adj, adj_attr, col)
torch.manual_seed(0)
h = torch.randn(n, 32, device=dev) * 3
wtab = torch.rand(g.n_types, 32, 32, device=dev)
root = torch.randn(32, 32, device=dev) * 0.2
bias = torch.randn(32, device=dev)
part = ops.new_partials(32, dev)
kern = sys.argv[1] if len(sys.argv) > 1 else "ps"
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, part, kernel=kern)
torch.cuda.synchronize()
