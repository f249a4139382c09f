"""The forward's two chains under stream priorities: the adjacency chain (NNConv -> merge, on the caller's stream) is the critical
path at 100k nodes and the collision chain has slack (profiles/r03_trace_100000.txt).  Cached-layout forward with the library call
made directly: (a) both streams of normal priority, (b) the main stream of high priority, (c) the side stream of low priority
(hipStreamCreateWithPriority through ctypes)."""
import sys, time, ctypes as C, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, _lib, ops
from tilingnn_amd._lib import lib, ptr, check
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
hip = C.CDLL('libamdhip64.so')
lo, hi = C.c_int(), C.c_int()
hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
print('priority range: least', lo.value, 'greatest', hi.value, flush=True)
def raw_stream(prio):
    s = C.c_void_p()
    assert hip.hipStreamCreateWithPriority(C.byref(s), 1, prio) == 0     # hipStreamNonBlocking
    return torch.cuda.ExternalStream(s.value, device=dev)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
for n in [int(a) for a in sys.argv[1:]] or [20000, 100000, 300000]:
    sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
    x, adj, attr, col, _ = sg.to_torch(dev)
    graph = ops.prepare_graph(n, adj, attr, col)
    dims = net._dims(); table, _ = net._param_table()
    ws_bytes = lib.tgnn_forward_workspace_bytes(C.byref(dims), n, graph.n_types)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev); probs = torch.empty(n, 1, device=dev)
    g = graph.c_struct()
    torch.cuda.synchronize()
    def run(main, side, reps=40):
        ts = []
        with torch.cuda.stream(main):
            for k in range(reps + 5):
                torch.cuda.synchronize(); t = time.perf_counter()
                check(lib.tgnn_forward(C.byref(dims), table, ptr(x), ptr(attr), C.byref(g), 0, 0, ptr(probs), ptr(ws), ws_bytes,
                                       C.c_void_p(main.cuda_stream), C.c_void_p(side.cuda_stream)))
                torch.cuda.synchronize()
                if k >= 5: ts.append((time.perf_counter() - t) * 1e3)
        return sorted(ts)[reps // 2], probs.clone()
    normal_a, normal_b = raw_stream(0), raw_stream(0)
    res = {}
    for name, main, side in (("normal / normal", normal_a, normal_b), ("high / normal", raw_stream(hi.value), raw_stream(0)),
                             ("normal / low", raw_stream(0), raw_stream(lo.value)), ("high / low", raw_stream(hi.value), raw_stream(lo.value)),
                             ("normal / normal again", normal_a, normal_b)):
        t, p = run(main, side)
        res[name] = p
        print(f"n {n}: main / side = {name}: {t:.3f} ms", flush=True)
    print("  bit-identical across the settings:", all(torch.equal(v, res["normal / normal"]) for v in res.values()), flush=True)
