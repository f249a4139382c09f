"""Forward latency on small layouts (the sizes the greedy loop actually sees)."""
import sys, time, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev)
for n in (300, 1254, 5000, 20000):
    sg = make_super_graph(n, int(n * 6.8), int(n * 8.35), tile_count=2, n_edge_types=13, seed=2)
    x, adj, adj_attr, col, _ = sg.to_torch(dev)
    for cached in (False, True):
        net.cache_graph = cached
        for _ in range(5): net(x=x, adj_e_index=adj, adj_e_features=adj_attr, col_e_idx=col)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): net(x=x, adj_e_index=adj, adj_e_features=adj_attr, col_e_idx=col)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50 * 1e3
        print(f"N={n:6d} cached_layout={cached}: {dt:.3f} ms per forward")
