"""Reads the in-kernel phase timers of a -DTGNN_TIMING build of nnconv_cols.hip (TGNN_LIB_PATH=scratch/libs/libtgnn_TIMING.so)."""
import sys, ctypes, torch, numpy as np
sys.path.insert(0, '.')
from tilingnn_amd import ops, _lib
from tilingnn_amd.synth import make_super_graph
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dev = torch.device('cuda:0')
sg = make_super_graph(n, 10 * n, 10 * n // 4 * 5, tile_count=2, n_edge_types=13, seed=2)
x, adj, adj_attr, col, _ = sg.to_torch(dev)
g = ops.prepare_graph(n, adj, adj_attr, col)
h = torch.randn(n, 32, device=dev)
wtab = torch.randn(g.n_types, 32, 32, device=dev) * 0.2
root = torch.randn(32, 32, device=dev) * 0.2
bias = torch.randn(32, device=dev)
part = ops.new_partials(32, dev)
for _ in range(3):
    ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, part)
torch.cuda.synchronize()
NB = 512
buf = (ctypes.c_ulonglong * (NB * 16 * 8))()
_lib.lib.tgnn_debug_col_timing.argtypes = [ctypes.c_void_p]
print("rc", _lib.lib.tgnn_debug_col_timing(buf))
t = np.frombuffer(buf, dtype=np.uint64).reshape(NB, 16, 8).astype(np.float64)[:224]
names = ["prologue", "chunk top", "gather wait + flags", "pre-add", "split + W + MFMA", "epilogue", "gather issue + idx", "tail/barrier"]
m = t.mean(axis=(0, 1))
print("clock ticks per wave (mean over waves), total", m.sum())
for k in range(8):
    print(f"  {names[k]:22s} {m[k]:10.0f}  {100 * m[k] / m.sum():5.1f}%")
tot = t.sum(-1).ravel()
print("wave total percentiles", np.percentile(tot, [0, 25, 50, 75, 100]))
