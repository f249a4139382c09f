"""Forward at 50 000 / 100 000 / 300 000 nodes with the NNConv on edge groups (default) and on type columns (TGNN_GROUPS=0 semantics:
tgnn_set_nnconv_eg(0) + graphs prepared with columns): cached layout and with preparation, median of 30; output difference."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN, ops
from tilingnn_amd._lib import lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
from tilingnn_amd.graph_networks import _graph_cache
dev = torch.device('cuda:0')
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
sizes = [int(a) for a in sys.argv[1:]] or [50_000, 100_000, 300_000]
for n in sizes:
    sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
    x, adj, attr, col, _ = sg.to_torch(dev)
    res = {}
    for mode in ("columns", "groups"):
        ops.GROUPS = mode == "groups"
        lib.tgnn_set_nnconv_eg(1 if mode == "groups" else 0)
        out = []
        for cached in (True, False):
            net.cache_graph = cached
            _graph_cache.clear()
            for _ in range(5): p = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0]
            torch.cuda.synchronize()
            ts = []
            for _ in range(30):
                t0 = time.perf_counter(); net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col); torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            out.append(sorted(ts)[15])
        res[mode] = (out, p.clone())
        print(f"n {n:7d} {mode:8s}: cached {out[0]:.3f} ms, with prep {out[1]:.3f} ms", flush=True)
    print(f"          max |p_groups - p_columns| = {float((res['groups'][1] - res['columns'][1]).abs().max()):.3e}")
