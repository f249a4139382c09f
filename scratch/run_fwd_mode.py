"""argv: split mode (0 / 1), forwards (default 10): cached-layout forwards at the benchmark shape for a rocprofv3 kernel trace."""
import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
_lib.lib.tgnn_set_split_precision(int(sys.argv[1]))
sg = make_super_graph(100_000, 1_000_000, 1_250_000, tile_count=2, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    net(x, adj, attr, col)
torch.cuda.synchronize()
