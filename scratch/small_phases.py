"""Phase timers of the persistent small-layout kernel (a -DTGNN_SMALL_TIMING build: scratch/build_abl.sh forward_small
SMALLTIME -DTGNN_SMALL_TIMING; TGNN_LIB_PATH=scratch/libs/libtgnn_SMALLTIME.so)."""
import ctypes as C, sys, torch
import numpy as np
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev)
names = ["stage weights", "phase A1 (own work)", "wait at sync", "A2 epilogue + row", "barrier 1", "B reduce+records", "merge", "barrier 2"]
raw = _lib.lib
for arg in sys.argv[1:] or ["laby", "300", "2500"]:
    if arg == "laby":
        from tests.golden_util import graph_tensors, load_labyrinth_graph
        inputs = graph_tensors(load_labyrinth_graph(), torch.float32, dev)[:4]
        n = 1254
    else:
        n = int(arg)
        sg = make_super_graph(n, int(n * 6.8), int(n * 8.35), tile_count=2, n_edge_types=13, seed=2)
        inputs = sg.to_torch(dev)[:4]
    for _ in range(3): net(*inputs)
    torch.cuda.synchronize()
    tiles = (n + 15) // 16
    teams, nb = 1, tiles
    buf = (C.c_ulonglong * (32 * nb))()
    rc = raw.tgnn_debug_small_timing(buf, nb)
    a = np.array(buf[:], dtype=np.float64).reshape(nb, 32)[:, :8] * 0.01 / 20      # us per layer (100 MHz ticks, 20 layers)
    print(f"N={n} blocks={nb} teams={teams} rc={rc}: per layer, us  [block 0 | mean | max over blocks]")
    for k, nm in enumerate(names):
        print(f"   {nm:22s} {a[0, k]:7.2f} {a[:, k].mean():7.2f} {a[:, k].max():7.2f}")
    print(f"   {'sum':22s} {a[0].sum():7.2f} {a.sum(1).mean():7.2f}")
    b = np.array(buf[:], dtype=np.float64).reshape(nb, 32)[:, 8:16] * 0.01 / 20
    for k, nm in enumerate(["GIN sum + z + split", "GIN layer 1", "GIN layers 2-3", "GIN out sigmoid + LDS", "GIN store", "GIN issue gathers", "GIN wait gathers"]):
        print(f"   {nm:22s} {b[0, k]:7.2f} {b[:, k].mean():7.2f} {b[:, k].max():7.2f}")
    c = np.array(buf[:], dtype=np.float64).reshape(nb, 32)[:, 16:24] * 0.01 / 20
    for k, nm in enumerate(["NN issue gathers", "NN wait gathers", "NN consume 8 cols", "NN tail + LDS write", "B issue loads", "B commit", "B wait + fold", "B LDS + sync"]):
        print(f"   {nm:22s} {c[0, k]:7.2f} {c[:, k].mean():7.2f} {c[:, k].max():7.2f}")
