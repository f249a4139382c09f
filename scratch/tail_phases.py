"""Phase stamps of the persistent final-MLP kernel (a -DTGNN_TAIL_TIMING build: scratch/build_abl.sh forward_tail TAILTIME
-DTGNN_TAIL_TIMING; TGNN_LIB_PATH=scratch/libs/libtgnn_TAILTIME.so python scratch/tail_phases.py [n ..])."""
import ctypes as C, sys, torch
import numpy as np
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev)
raw = C.CDLL(_lib.LIB_PATH)
names = ["dense", "sums+stores", "(unused)", "level 1", "level 2", "record"]
for arg in sys.argv[1:] or ["10000", "20000"]:
    n = int(arg)
    ea, ec = (8 * n, 10 * n) if n == 10000 else (10 * n, 12 * n + n // 2)
    sg = make_super_graph(n, ea, ec, tile_count=2, n_edge_types=13, seed=1)
    inputs = sg.to_torch(dev)[:4]
    for _ in range(3): net(*inputs)
    torch.cuda.synchronize()
    tiles = (n + 15) // 16
    k = max(2, (tiles + 255) // 256)
    nb = (tiles + k - 1) // k
    buf = (C.c_ulonglong * (32 * nb))()
    rc = raw.tgnn_debug_tail_timing(buf, nb)
    a = np.array(buf[:], dtype=np.float64).reshape(nb, 32) * 0.01     # us
    t0 = a[:, 24].min()
    print(f"N={n} blocks={nb} tiles/block={k} rc={rc}: entry spread {a[:, 24].max() - t0:.1f} us, exit (last block) {a[:, 25].max() - t0:.1f} us")
    prev = a[:, 24]
    for l in range(4):
        for j, nm in enumerate(names):
            cur = a[:, l * 6 + j]
            print(f"   layer {l} {nm:16s} at {np.median(cur) - t0:7.1f} us (median)   phase: block 0 {cur[0] - prev[0]:6.2f}  median {np.median(cur - prev):6.2f}  max {np.max(cur - prev):6.2f}")
            prev = cur
    print(f"   read-out: median {np.median(a[:, 25] - prev):.2f}")
    prev = a[:, 24]
    for j, nm in enumerate(["chunk 0 rows in + split", "chunk 0 barrier", "chunk 0 multiply", "chunk 1 (barrier +) split", "chunk 1 barrier", "chunk 1 multiply"]):
        cur = a[:, 26 + j]
        print(f"   layer 0 {nm:26s} at {np.median(cur) - t0:7.1f} us   phase: block 0 {cur[0] - prev[0]:6.2f}  median {np.median(cur - prev):6.2f}  max {np.max(cur - prev):6.2f}")
        prev = cur
