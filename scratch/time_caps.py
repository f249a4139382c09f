"""Cached-layout forward at 100 000 nodes against upper bounds on the grids of the two whole-CU kernels (tgnn_debug_set_block_caps)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
n = 100_000
sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
for nn_cap, gin_cap in ((0, 0), (0, 192), (0, 160), (0, 128), (0, 96), (256, 0), (256, 160), (240, 160)):
    _lib.lib.tgnn_debug_set_block_caps(nn_cap, gin_cap)
    for _ in range(5): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"NNConv cap {nn_cap or 'default (CUs - 32)'}, GIN MLP cap {gin_cap or 'default (CUs - 32)'}: cached forward {sorted(ts)[15]:.3f} ms")
_lib.lib.tgnn_debug_set_block_caps(0, 0)
