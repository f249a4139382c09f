"""Single-stream forward (tgnn_forward with stream2 = the main stream: every kernel alone on the chip) n times, for rocprofv3
kernel stats of the kernels ISOLATED.  argv: n_nodes [gin_mlp_f16 0|1]"""
import sys, ctypes as C, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, ops, _lib
from tilingnn_amd._lib import check, lib, ptr
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
if len(sys.argv) > 2: lib.tgnn_set_gin_mlp_f16(int(sys.argv[2]))
sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
graph = ops.prepare_graph(n, adj, attr, col)
dims = net._dims(); table, _ = net._param_table()
ws_bytes = lib.tgnn_forward_workspace_bytes(C.byref(dims), n, graph.n_types)
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
probs = torch.empty(n, 1, dtype=torch.float32, device=dev)
g = graph.c_struct()
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for _ in range(10):
    check(lib.tgnn_forward(C.byref(dims), table, ptr(x), ptr(attr), C.byref(g), 1, 0, ptr(probs), ptr(ws), ws_bytes, stream, stream))
torch.cuda.synchronize()
print("done", float(probs.sum()))
