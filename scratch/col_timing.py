import sys, os, ctypes, torch, numpy as np
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, ops, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
sg = make_super_graph(100_000, 1_000_000, 1_250_000, tile_count=2, n_edge_types=13, seed=2)
x, adj, adj_attr, col, _ = sg.to_torch(dev)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev)
h = torch.randn(100_000, 32, device=dev)
g = ops.prepare_graph(100_000, adj, adj_attr, col)
l1 = net.brch_1_graph_conv_layers[0]
wtab = ops.edge_weight_table(adj_attr, g, *l1.nnConv._edge_mlp_params(), 32)
parts = ops.new_partials(32, dev)
for _ in range(3): ops.nnconv_mean(h, g, wtab, l1.nnConv.root, l1.nnConv.bias, act=1, partials=parts)
torch.cuda.synchronize()
NB = 512
buf = (ctypes.c_ulonglong * (NB*8*8))()
_lib.lib.tgnn_debug_col_timing.argtypes = [ctypes.c_void_p]
print("rc", _lib.lib.tgnn_debug_col_timing(buf))
t = np.frombuffer(buf, dtype=np.uint64).reshape(NB, 8, 8).astype(np.float64)
names = ["prologue", "consume wait+af", "B load+MFMA", "epilogue", "unpack", "gather issue", "group load", "bn final"]
m = t.mean(axis=(0, 1))
print("clock ticks per wave (mean over 2048 waves), total", m.sum())
for k in range(8): print(f"  {names[k]:22s} {m[k]:10.0f}  {100*m[k]/m.sum():5.1f}%")
print("per-wave totals: min/max", t.sum(-1).min(), t.sum(-1).max())
tot = t.sum(-1).ravel()
print("wave total percentiles", np.percentile(tot, [0, 25, 50, 75, 100]))
