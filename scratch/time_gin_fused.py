import sys, time, torch
sys.path.insert(0, ".")
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device("cuda:0")
_lib.lib.tgnn_set_mid_layout_limit(0)
for n in [int(a) for a in sys.argv[1:]] or [20000, 50000, 100000, 300000]:
    sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
    x, adj, attr, col, _ = sg.to_torch(dev)
    net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
    net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0), strict=True)
    net = net.to(dev).train()
    res = {}
    for rep in range(2):
        for on in (0, 1):
            _lib.lib.tgnn_set_gin_fused(2 * on)
            for _ in range(5):
                net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(20):
                    net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 20 * 1e3)
            res[on] = min(ts) if on not in res else min(res[on], min(ts))
    print(f"n {n:7d}: cached-layout forward two-kernel GIN {res[0]:.3f} ms, fused {res[1]:.3f} ms", flush=True)
