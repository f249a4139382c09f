"""Cached-layout forward at the benchmark shape with the library TGNN_LIB_PATH names: ms per forward (HIP events, 3 x 30 forwards)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
sg = make_super_graph(100_000, 1_000_000, 1_250_000, tile_count=2, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
for _ in range(5): net(x, adj, attr, col)
ts = []
for _ in range(3):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): net(x, adj, attr, col)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 30)
print(os.path.basename(os.environ.get("TGNN_LIB_PATH", "default")), " ".join(f"{t:.4f}" for t in ts))
