"""Host time of enqueueing one forward at 100 000 nodes (cached layout / with preparation), against the device time of the same forwards."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=1)
x, adj, attr, col, _ = sg.to_torch(dev)
for cached in (True, False):
    net.cache_graph = cached
    for _ in range(10): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
        ts.append((time.perf_counter() - t0) * 1e6)
    # 20 back to back: host time until the last call returns, and until the device is done
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"cached={cached}: host time of one call on an idle device {sorted(ts)[15]:.0f} us; 20 back to back: host returns after {(t1 - t0) / 20 * 1e6:.0f} us per call, "
          f"device done after {(t2 - t0) / 20 * 1e6:.0f} us per call")
