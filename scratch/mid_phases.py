"""Phase timers of the mid-size persistent kernel (a -DTGNN_MID_TIMING build: scratch/build_abl.sh forward_mid MIDTIME
-DTGNN_MID_TIMING; TGNN_LIB_PATH=scratch/libs/libtgnn_MIDTIME.so)."""
import ctypes as C, sys, torch
import numpy as np
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev)
names = ["shadow tail->top", "top sync", "NNConv items (wave 0)", "wait all waves", "R: block row + level 1", "R: level 2", "R: stats",
         "merge", "arrive sync", "zero+dma", "GIN items (wave 0)", "B wait"]
raw = _lib.lib
for arg in sys.argv[1:] or ["10000", "20000", "50000"]:
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
    rc = raw.tgnn_debug_mid_timing(buf, nb)
    a = np.array(buf[:], dtype=np.float64).reshape(nb, 64) * 0.01 / 20      # us per layer (100 MHz ticks, 20 layers)
    print(f"N={n} blocks={nb} tiles/block={k} rc={rc}: per layer, us  [block 0 | mean | max over blocks]")
    for j, nm in enumerate(names):
        print(f"   {nm:26s} {a[0, j]:7.2f} {a[:, j].mean():7.2f} {a[:, j].max():7.2f}")
    print(f"   {'sum':26s} {a[0, :12].sum():7.2f} {a[:, :12].sum(1).mean():7.2f}")
    nn, gin = a[:, 16:32], a[:, 32:48]
    print("   NNConv items per wave (mean over blocks):", " ".join(f"{v:.1f}" for v in nn.mean(0)))
    print("   GIN items per wave (mean over blocks):   ", " ".join(f"{v:.1f}" for v in gin.mean(0)))
