"""Runs only NNConv launches (for PMC passes): N=100k/Ea=1M synthetic graph."""
import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, ops
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
sg = make_super_graph(100_000, 1_000_000, 1_250_000, tile_count=2, n_edge_types=13, seed=2)
x, adj, adj_attr, col, _ = sg.to_torch(dev)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev)
h = torch.randn(100_000, 32, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else 'nnconv'
for _ in range(5):
    if which == 'nnconv':
        net.brch_1_graph_conv_layers[0](h, adj, adj_attr)
    else:
        net.brch_2_coll_conv_layers[0](h, col)
torch.cuda.synchronize()
