"""40 cached-layout forwards at 100 000 nodes (argv[1]: groups | columns; argv[3] = 1: the collision MLP on fp16 pairs) for a kernel trace."""
import os, sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, ops
from tilingnn_amd._lib import lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
mode = sys.argv[1] if len(sys.argv) > 1 else "groups"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
ops.GROUPS = mode == "groups"
lib.tgnn_set_nnconv_eg(1 if mode == "groups" else 0)
if len(sys.argv) > 3: lib.tgnn_set_gin_mlp_f16(int(sys.argv[3]))
dev = torch.device('cuda:0')
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
for _ in range(40): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
torch.cuda.synchronize()
