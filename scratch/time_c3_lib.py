"""Config 3 cached / with-preparation step with the library TGNN_LIB_PATH names (HIP events, 3 x 20 forwards each)."""
import os, sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
n = 100_000
sg = make_super_graph(n, 10 * n, 10 * n // 4 * 5, tile_count=4, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=64, node_features_dim=5)
net.load_state_dict(make_state_dict(15, 20, 64, 1, 5, seed=0))
net = net.to(dev).train()
net.activation_dtype = torch.bfloat16
res = []
for cache in (True, False):
    net.cache_graph = cache
    for _ in range(4): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20)
print(os.path.basename(os.environ.get("TGNN_LIB_PATH", "default")), "cached", " ".join(f"{t:.4f}" for t in res[:3]), " with preparation", " ".join(f"{t:.4f}" for t in res[3:]), flush=True)
