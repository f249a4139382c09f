"""BASELINE config 3: 100k nodes / 1M + 1.25M edges, tile_count 4, width 64, depth 20, bf16 storage: step time (prep included / cached)."""
import sys, time, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
sg = make_super_graph(n, 10 * n, 10 * n // 4 * 5, tile_count=4, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=64, node_features_dim=5)
net.load_state_dict(make_state_dict(15, 20, 64, 1, 5, seed=0))
net = net.to(dev).train()
for mode in (torch.bfloat16, torch.float32):
    net.activation_dtype = mode
    for cache in (False, True):
        net.cache_graph = cache
        for _ in range(3):
            net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            t = time.perf_counter()
            net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
        print(f"width 64 storage {mode} cached_layout={cache}: median {sorted(ts)[5]:.3f} ms  ({n / sorted(ts)[5] / 1e3:.2f} M nodes/s)")
