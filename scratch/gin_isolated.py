import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device("cuda:0")
_lib.lib.tgnn_set_mid_layout_limit(0)
for n in (100000, 300000):
    sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
    x, adj, attr, col, _ = sg.to_torch(dev)
    net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
    net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0), strict=True)
    net = net.to(dev).train()
    for on in (0, 1):
        _lib.lib.tgnn_set_gin_fused(2 * on)
        bench.profiled_classes(net, x, adj, attr, col, 3)
        cls, _ = bench.profiled_classes(net, x, adj, attr, col, 10)
        print(n, "fused" if on else "two kernels", {k: round(v["ms_per_forward"] / max(1, v["launches_per_forward"]) * 1e3, 1) for k, v in cls.items() if k in ("nnconv", "gin", "merge")}, flush=True)
