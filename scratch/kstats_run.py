"""A few cached-layout forwards for rocprofv3 --kernel-trace --stats: scratch/kstats_run.py <n> <gin_fused 0|1> <mid 0|1>"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device("cuda:0")
n = int(sys.argv[1]); _lib.lib.tgnn_set_gin_fused(2 * int(sys.argv[2]))
if len(sys.argv) > 3 and sys.argv[3] == "0": _lib.lib.tgnn_set_mid_layout_limit(0)
sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0), strict=True)
net = net.to(dev).train()
for _ in range(10):
    net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
torch.cuda.synchronize()
