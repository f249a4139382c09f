"""Five bf16-storage forwards at BASELINE config 3 (100k / 1M / 1.25M, tile_count 4, width 64), cached layout: the process
rocprofv3 --pmc passes are taken on."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
sg = make_super_graph(100_000, 1_000_000, 1_250_000, tile_count=4, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=64, node_features_dim=5)
net.load_state_dict(make_state_dict(15, 20, 64, 1, 5, seed=0))
net = net.to(dev).train()
net.activation_dtype = torch.bfloat16
for _ in range(5):
    net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
torch.cuda.synchronize()
