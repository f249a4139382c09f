"""fp16 x 2 split: the result must not depend on the power-of-two scale.  The same forward with the layout's in-degree bound
inflated (x4, x64: scales 4 / 64 times smaller) against the true bound; bf16 x 3 beside it."""
import sys, ctypes as C, torch
sys.path.insert(0, '.')
from tilingnn_amd import _lib, ops
from tilingnn_amd._lib import lib, ptr, check
from tilingnn_amd.synth import make_super_graph
from tests.test_hip_parity import make_net
dev = torch.device('cuda:0')
lib.tgnn_set_small_layout_limit(0)
for n in (6000, 100_000):
    sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=8)
    x, adj, attr, col, _ = sg.to_torch(dev)
    for depth in (3, 20):
        net, _ = make_net(dev, depth=depth)
        graph = ops.prepare_graph(n, adj, attr, col)
        dims = net._dims(); table, _ = net._param_table()
        ws_bytes = lib.tgnn_forward_workspace_bytes(C.byref(dims), n, graph.n_types)
        outs = {}
        for name, mode, mul in (("bf16x3", 0, 1), ("f16 x1", 1, 1), ("f16 x4", 1, 4), ("f16 x64", 1, 64)):
            lib.tgnn_set_split_precision(mode)
            g = graph.c_struct(); g.nn_max_in_degree = graph.max_in_degree * mul
            ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev); probs = torch.empty(n, 1, device=dev)
            check(lib.tgnn_forward(C.byref(dims), table, ptr(x), ptr(attr), C.byref(g), 0, 0, ptr(probs), ptr(ws), ws_bytes,
                                   _lib.current_stream(dev), None))
            torch.cuda.synchronize(); outs[name] = probs.clone()
        ref = outs["f16 x1"]
        print(f"n {n} depth {depth} (max in-degree {graph.max_in_degree}): " +
              ", ".join(f"|{k} - f16 x1| {float((v - ref).abs().max()):.2e}" for k, v in outs.items() if k != "f16 x1"), flush=True)
lib.tgnn_set_split_precision(1)
