"""Cached-layout forward at small N with the NNConv column structure vs without (CSR / LDS-table kernel)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN, ops
from tilingnn_amd.graph_networks import _graph_cache
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0)); net = net.cuda().train()
orig = ops.build_nnconv_columns
def bench(f, n=40):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for n in (300, 1254, 2500, 5000, 10000, 20000):
    sg = make_super_graph(n, 10 * n, int(12.5 * n), tile_count=2, n_edge_types=13, seed=2)
    x, adj, attr, col, _ = sg.to_torch("cuda:0")
    out = []
    with torch.no_grad():
        for use_cols in (True, False):
            ops.build_nnconv_columns = orig if use_cols else (lambda *a, **k: None)
            _graph_cache.clear()
            net.cache_graph = True
            cached = bench(lambda: net(x, adj, attr, col))
            net.cache_graph = False
            full = bench(lambda: net(x, adj, attr, col))
            out.append((cached, full))
    print(f"N={n}: columns cached {out[0][0]:.3f} / with prep {out[0][1]:.3f} ms;  CSR cached {out[1][0]:.3f} / with prep {out[1][1]:.3f} ms")
