"""BASELINE config 3 (100 000 nodes, width 64, bf16 storage): NNConv on edge groups (default) against type columns
(tgnn_set_nnconv_eg(0) + TGNN_GROUPS off), same box: step time with preparation / cached, median of 20."""
import sys, time, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, ops
from tilingnn_amd._lib import lib
from tilingnn_amd.graph_networks import _graph_cache
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
sg = make_super_graph(n, 10 * n, 10 * n // 4 * 5, tile_count=4, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=64, node_features_dim=5)
net.load_state_dict(make_state_dict(15, 20, 64, 1, 5, seed=0))
net = net.to(dev).train()
net.activation_dtype = torch.bfloat16
res = {}
for rep in range(2):
    for mode in ("columns", "groups"):
        ops.GROUPS = mode == "groups"
        lib.tgnn_set_nnconv_eg(1 if mode == "groups" else 0)
        for cache in (False, True):
            net.cache_graph = cache
            _graph_cache.clear()
            for _ in range(3):
                p = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0]
            torch.cuda.synchronize()
            ts = []
            for _ in range(20):
                t = time.perf_counter()
                net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t) * 1e3)
            print(f"config 3, NNConv on {mode:8s} cached_layout={cache}: median {sorted(ts)[10]:.3f} ms", flush=True)
        res[mode] = p.clone()
print("max |p_groups - p_columns| =", float((res["groups"] - res["columns"]).abs().max()))
