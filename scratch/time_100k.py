"""Cached-layout and with-preparation forward at the benchmark shape (and 50 000 / 300 000 nodes): median of 30."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
from tilingnn_amd.graph_networks import _graph_cache
dev = torch.device('cuda:0')
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
for n in (50_000, 100_000, 300_000):
    sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
    x, adj, attr, col, _ = sg.to_torch(dev)
    out = []
    for cached in (True, False):
        net.cache_graph = cached
        _graph_cache.clear()
        for _ in range(5): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        out.append(sorted(ts)[15])
    print(f"n {n:7d}: cached {out[0]:.3f} ms, with prep {out[1]:.3f} ms")
