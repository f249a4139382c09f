"""Cached-layout forward at mid sizes: wall time per forward, host time of the library call, and (under rocprofv3) the kernel time."""
import sys, time, ctypes as C, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, _lib, ops
from tilingnn_amd._lib import lib, ptr, check
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
for n in [int(a) for a in sys.argv[1:]] or [5000, 10000, 20000, 50000]:
    sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
    x, adj, attr, col, _ = sg.to_torch(dev)
    for _ in range(5): net(x, adj, attr, col)
    torch.cuda.synchronize(); ts = []
    for _ in range(30):
        t = time.perf_counter(); net(x, adj, attr, col); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    graph = ops.prepare_graph(n, adj, attr, col)
    dims = net._dims(); table, _ = net._param_table()
    ws_bytes = lib.tgnn_forward_workspace_bytes(C.byref(dims), n, graph.n_types)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev); probs = torch.empty(n, 1, device=dev)
    g = graph.c_struct(); st, side = _lib.current_stream(dev), _lib.side_stream(dev)
    hs = []
    for _ in range(20):
        torch.cuda.synchronize(); t = time.perf_counter()
        check(lib.tgnn_forward(C.byref(dims), table, ptr(x), ptr(attr), C.byref(g), 1, 0, ptr(probs), ptr(ws), ws_bytes, st, side))
        hs.append((time.perf_counter() - t) * 1e3); torch.cuda.synchronize()
    print(f"n {n}: forward (cached layout) median {sorted(ts)[15]:.3f} ms; host time of tgnn_forward (enqueue only) {sorted(hs)[10]:.3f} ms", flush=True)
