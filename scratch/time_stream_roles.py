"""Per-role cycle split of the stream NNConv kernel (library built with -DTGNN_ST_TIMING)."""
import sys, ctypes, torch, numpy as np
sys.path.insert(0, '.')
from tilingnn_amd import ops, _lib
from tilingnn_amd.synth import make_super_graph
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ea = 10 * n
sg = make_super_graph(n, ea, ea // 4 * 5, tile_count=2, n_edge_types=13, seed=2)
x, adj, adj_attr, col, _ = sg.to_torch(dev)
g = ops.prepare_graph(n, adj, adj_attr, col)
torch.manual_seed(0)
h = torch.randn(n, 32, device=dev)
wtab = torch.rand(g.n_types, 32, 32, device=dev); root = torch.randn(32, 32, device=dev) * 0.2; bias = torch.randn(32, device=dev)
part = ops.new_partials(32, dev)
for _ in range(3):
    ops.nnconv_mean(h, g, wtab, root, bias, 1, part, kernel="stream")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (512 * 12 * 4))()
_lib.lib.tgnn_debug_stream_time(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 12, 4)[:256].astype(np.float64)
names = ["mult"] * 8 + ["load"] * 2 + ["epil"] * 2
print("cycles (100 MHz counter? s_memtime ticks) per block, mean over 256 blocks")
for w in range(12):
    print(f" wave {w:2d} {names[w]}: work {a[:, w, 0].mean():9.0f}  barrier wait {a[:, w, 1].mean():9.0f}  steps {a[:, w, 2].mean():5.1f}  total {a[:, w, 3].mean():9.0f}  (max total {a[:, w, 3].max():9.0f})")

buf2 = (ctypes.c_ulonglong * (512 * 12 * 4))()
if hasattr(_lib.lib, "tgnn_debug_stream_seg"):
    _lib.lib.tgnn_debug_stream_seg(buf2)
    b = np.frombuffer(buf2, dtype=np.uint64).reshape(512, 12, 4)[:256].astype(np.float64)
    print("segments (mult: next info | run A | run B;  epil: DMA issue | row info wait | row sums | finish + store + DMA wait)")
    for w in range(12):
        print(f" wave {w:2d} {names[w]}: " + "  ".join(f"{b[:, w, k].mean():9.0f}" for k in range(4)))
