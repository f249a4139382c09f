"""Cached-layout forward at mid sizes under upper bounds on the grids of the two whole-CU kernels (column NNConv, GIN MLP): both
need a CU to themselves, so 224 + 224 blocks cannot be resident together and the two chains of the forward run one after the other
(scratch/mid_trace.sh); with bounds that add up to the device they run side by side."""
import sys, time, ctypes as C, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
f = _lib.lib.tgnn_debug_set_block_caps
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
caps = [(0, 0), (256, 224), (240, 224), (256, 192), (240, 192), (224, 192), (224, 160), (256, 256), (0, 0)]
for n in [int(a) for a in sys.argv[1:]] or [5000, 10000, 20000, 30000, 50000, 100000]:
    sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
    x, adj, attr, col, _ = sg.to_torch(dev)
    out = []
    for a, b in caps:
        f(a, b)
        for _ in range(5): net(x, adj, attr, col)
        torch.cuda.synchronize(); ts = []
        for _ in range(30):
            t = time.perf_counter(); net(x, adj, attr, col); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        out.append(f"({a},{b}) {sorted(ts)[15]:.3f}")
    print(f"n {n}: " + "  ".join(out), flush=True)
