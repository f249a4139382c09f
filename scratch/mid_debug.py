import sys, ctypes as C, torch
sys.path.insert(0, ".")
from tests.test_mid_layout import _layout
from tests.test_hip_parity import make_net
from tilingnn_amd import _lib, ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
inputs, _ = _layout(n, dev)
def run(limit, depth=1):
    _lib.lib.tgnn_set_mid_layout_limit(limit)
    x, adj, attr, col = inputs
    graph = ops.prepare_graph(n, adj, attr, col)
    net, _ = make_net(dev, depth=depth)
    dims = net._dims(); table, _ = net._param_table()
    ws_bytes = _lib.lib.tgnn_forward_workspace_bytes(C.byref(dims), n, graph.n_types)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
    probs = torch.empty(n, 1, device=dev)
    g = graph.c_struct()
    _lib.check(_lib.lib.tgnn_forward(C.byref(dims), table, ops.ptr(x), ops.ptr(attr), C.byref(g), 0, 0, ops.ptr(probs), ops.ptr(ws), ws_bytes,
                                    _lib.current_stream(dev), _lib.side_stream(dev)))
    torch.cuda.synchronize()
    f = ws.view(torch.float32)
    al = lambda k: (k + 63) // 64 * 64
    o = 0
    mid = f[o:o + (depth + 1) * n * 32].view(depth + 1, n, 32).clone(); o = al(o + (depth + 1) * n * 32)
    a1 = f[o:o + n * 32].view(n, 32).clone(); o = al(o + n * 32)
    a20 = f[o:o + n * 32].view(n, 32).clone(); o = al(o + n * 32)
    return mid.cpu(), a1.cpu(), a20.cpu(), probs.cpu(), graph
mg, a1g, a2g, pg, _ = run(0)
mm, a1m, a2m, pm, graph = run(65536)
print("mid taken:", graph.mid is not None)
for name, a, b in (("a1 (NNConv)", a1m, a1g), ("a2 (GIN)", a2m, a2g), ("slot1", mm[1], mg[1])):
    bad = ~torch.isfinite(a)
    print(name, "non-finite:", int(bad.sum()), "rows:", bad.any(1).nonzero().flatten()[:20].tolist(), "max diff (finite):", float((a - b)[~bad].abs().max()))
    if bad.any():
        r = int(bad.any(1).nonzero()[0])
        print("  row", r, "tile", r // 16, "block", r // 16 // ((n + 15) // 16 // 256 + 1), a[r][:8].tolist())
for cap in (100, 40):
    _lib.lib.tgnn_debug_set_mid_blocks(cap)
    mm, a1m, a2m, pm, graph = run(65536)
    bad = ~torch.isfinite(a1m)
    d = (a1m - a1g).abs()
    print("blocks cap", cap, "a1 non-finite:", int(bad.sum()), "rows with |diff| > 1e-3:", (d > 1e-3).any(1).nonzero().flatten()[:20].tolist())
_lib.lib.tgnn_debug_set_mid_blocks(0)
mm, a1m, a2m, pm, graph = run(65536)
d = (a1m - a1g).abs(); d[~torch.isfinite(d)] = 1e9
rows = (d > 1e-3).any(1).nonzero().flatten().tolist()
print("default: rows with |diff| > 1e-3:", rows[:30])
# the batches of the first bad row's tile
if rows:
    import numpy as np
    r = rows[0]; t = r // 16
    nb = int(graph.mid.tile_nb[t]); ent = graph.mid.ent.cpu().numpy().view(np.uint32).reshape(-1, 24, 36)[t]
    print("tile", t, "row", r % 16, "batches", nb)
    for b in range(nb):
        ws = [int(ent[b, 4 + 4 * o + g]) for g in range(4) for o in range(8)]
        print(b, hex(int(ent[b, 0])), hex(int(ent[b, 1])), [(w & 0xffffff, (w >> 25) & 31, (w >> 30) & 1) for w in ws if w != 0x21000000])
