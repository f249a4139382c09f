"""Experiment (round 6): the two chains of the forward on CU-masked streams (hipExtStreamCreateWithCUMask) -- does keeping the
collision chain (side stream) and / or the adjacency chain (main stream) on disjoint CU sets shorten the layer period?
Cached-layout forward at the benchmark shape, ms per forward (HIP events, 3 x 30 forwards) per mask pair."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict

hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits[32 * w + b]) for w in range(8)])
    h = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value)


dev = torch.device('cuda:0')
W = int(sys.argv[1]) if len(sys.argv) > 1 else 32
FX = 3 if W == 32 else 5
sg = make_super_graph(100_000, 1_000_000, 1_250_000, tile_count=2 if W == 32 else 4, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(15, 20, W, node_features_dim=FX)
net.load_state_dict(make_state_dict(15, 20, W, 1, FX)); net = net.to(dev).train()
if W == 64:
    net.activation_dtype = torch.bfloat16


def run(tag, main, side):
    if side is not None:
        _lib._side_streams[0] = side
    ctx = torch.cuda.stream(main) if main is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(5): net(x, adj, attr, col)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): net(x, adj, attr, col)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 30)
    print(f"{tag:44s}", " ".join(f"{t:.4f}" for t in ts), flush=True)


net(x, adj, attr, col)
default_side = _lib.side_stream_torch(dev)
run("default streams", None, default_side)
ALL = [1] * 256
for name, pat in [("side: every 2nd CU (128)", [i % 2 for i in range(256)]),
                  ("side: every 4th CU (64)", [int(i % 4 == 0) for i in range(256)]),
                  ("side: 3 of 4 CUs (192)", [int(i % 4 != 0) for i in range(256)]),
                  ("side: first 128", [int(i < 128) for i in range(256)]),
                  ("side: first 96", [int(i < 96) for i in range(256)]),
                  ("side: all 256 (masked stream, full mask)", ALL)]:
    run(name, None, masked_stream(pat))
for name, pm, ps in [("main 3 of 4 / side the 4th", [int(i % 4 != 0) for i in range(256)], [int(i % 4 == 0) for i in range(256)]),
                     ("main even / side odd", [int(i % 2 == 0) for i in range(256)], [i % 2 for i in range(256)]),
                     ("main first 160 / side last 96", [int(i < 160) for i in range(256)], [int(i >= 160) for i in range(256)]),
                     ("main all / side all (both masked)", ALL, ALL)]:
    run(name, masked_stream(pm), masked_stream(ps))
run("default streams again", None, default_side)
