import sys, time, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
sg = make_super_graph(100_000, 1_000_000, 1_250_000, tile_count=2, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
for _ in range(5): net(x, adj, attr, col)
torch.cuda.synchronize(); ts = []
for _ in range(20):
    t = time.perf_counter(); net(x, adj, attr, col); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print(f"cached-layout forward at 100k: median {sorted(ts)[10]:.3f} ms min {min(ts):.3f}")
