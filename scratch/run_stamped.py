"""Production (two-stream) forwards with the NNConv launches stamped on the device clock (tgnn_forward_stamped); run under
rocprofv3 --kernel-trace --stats the two measurements of the same launches can be compared."""
import sys, torch
sys.path.insert(0, '.')
from bench import stamped_nnconv_us
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
sg = make_super_graph(100_000, 1_000_000, 1_250_000, tile_count=2, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
us, cnt = stamped_nnconv_us(net, x, adj, attr, col, 20)
print(f"stamped: {us:.2f} us average over {cnt} launches")
