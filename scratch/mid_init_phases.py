"""Stamps of the init MLP in the mid-size kernel's prologue (-DTGNN_MID_TIMING build: scratch/build_abl.sh forward_mid MIDTIME -DTGNN_MID_TIMING)."""
import ctypes as C, sys, torch
import numpy as np
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev)
raw = C.CDLL(_lib.LIB_PATH)
names = ["entry", "Linear 0 + sums", "all-reduce 0", "record 0", "Linear 1 + sums", "all-reduce 1", "record 1", "apply + barrier"]
for arg in sys.argv[1:] or ["10000"]:
    n = int(arg)
    ea, ec = (8 * n, 10 * n) if n == 10000 else (10 * n, 12 * n + n // 2)
    sg = make_super_graph(n, ea, ec, tile_count=2, n_edge_types=13, seed=1)
    inputs = sg.to_torch(dev)[:4]
    for _ in range(3): net(*inputs)
    torch.cuda.synchronize()
    tiles = (n + 15) // 16
    k = (tiles + 255) // 256
    k = next((p for p in (1, 2, 4, 8) if k <= p), k)
    nb = (tiles + k - 1) // k
    buf = (C.c_ulonglong * (64 * nb))()
    raw.tgnn_debug_mid_timing(buf, nb)
    a = np.array(buf[:], dtype=np.float64).reshape(nb, 64)[:, 48:56] * 0.01
    t0 = a[:, 0].min()
    print(f"N={n} blocks={nb} tiles/block={k}")
    for j in range(1, 8):
        d = a[:, j] - a[:, j - 1]
        print(f"   {names[j]:18s} at {np.median(a[:, j]) - t0:6.1f} us   phase: block 0 {d[0]:6.2f}  median {np.median(d):6.2f}  max {d.max():6.2f}")
