import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN, train
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
T = int(sys.argv[1]) if len(sys.argv) > 1 else 25
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sg = make_super_graph(600, 6000, 7500, tile_count=2, n_edge_types=T, seed=9)
fe = 2 + T
net = TilinGNN(adj_edge_features_dim=fe, network_depth=depth, network_width=32, node_features_dim=3)
net.load_state_dict(make_state_dict(fe, depth, 32, 1, 3, seed=4)); net = net.cuda().train()
x, adj, attr, col, _ = sg.to_torch("cuda:0")
probs, sv = train.forward_train(net, x, adj, attr, col)
dp = torch.randn(600, 1, device="cuda") * 1e-2
a = train.backward_library(net, sv, dp); torch.cuda.synchronize()
b = train.backward_train(net, sv, dp); torch.cuda.synchronize()
bad = [(k, float((a[k].reshape(-1) - b[k].reshape(-1)).abs().max()), float(b[k].abs().max())) for k in a if not torch.equal(a[k].reshape(-1), b[k].reshape(-1))]
print("T", T, "depth", depth, "types", sv.tg.g.n_types, "cols", sv.tg.g.cols is not None, "mismatching", len(bad), "of", len(a))
for r in bad[:12]: print("  ", r)
