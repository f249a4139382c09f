"""A few launches of the stream NNConv kernel alone (N=100k/Ea=1M synthetic graph), for rocprofv3 passes.  argv[1]: cols|stream, argv[2]: n"""
import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops
from tilingnn_amd.synth import make_super_graph
dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else 'stream'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
sg = make_super_graph(n, 10 * n, 10 * n // 4 * 5, tile_count=2, n_edge_types=13, seed=2)
x, adj, adj_attr, col, _ = sg.to_torch(dev)
g = ops.prepare_graph(n, adj, adj_attr, col)
torch.manual_seed(0)
h = torch.randn(n, 32, device=dev)
wtab = torch.rand(g.n_types, 32, 32, device=dev); root = torch.randn(32, 32, device=dev) * 0.2; bias = torch.randn(32, device=dev)
part = ops.new_partials(32, dev)
for _ in range(6):
    ops.nnconv_mean(h, g, wtab, root, bias, 1, part, kernel=which)
torch.cuda.synchronize()
