"""Cached-layout / with-preparation forward time at mid sizes, persistent layer loop against the general schedule."""
import sys, time, torch
sys.path.insert(0, ".")
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.graph_networks import _graph_cache
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [5000, 10000, 20000, 50000]
for n in sizes:
    ea, ec = (8 * n, 10 * n) if n == 10000 else (10 * n, 12 * n + n // 2)
    sg = make_super_graph(n, ea, ec, tile_count=2, n_edge_types=13, seed=1)
    x, adj, attr, col, _ = sg.to_torch(dev)
    net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
    net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0), strict=True)
    net = net.to(dev).train()
    out = {}
    for name, limit in (("general", 0), ("mid", 65536)):
        _lib.lib.tgnn_set_mid_layout_limit(limit)
        _graph_cache.clear()
        for cache in (True, False):
            net.cache_graph = cache
            for _ in range(5):
                net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
            torch.cuda.synchronize()
            ts = []
            for rep in range(5):
                t0 = time.perf_counter()
                for _ in range(20):
                    net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 20 * 1e3)
            out[(name, cache)] = min(ts)
    print(f"n {n:6d}: general cached {out[('general', True)]:.3f} ms, with prep {out[('general', False)]:.3f} | "
          f"mid cached {out[('mid', True)]:.3f} ms, with prep {out[('mid', False)]:.3f}", flush=True)
