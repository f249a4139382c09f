"""prepare_graph at the benchmark shape: wall time per call (one sync inside) and the queue-to-done time of the launches."""
import sys, time, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops
from tilingnn_amd.synth import make_super_graph
dev = torch.device('cuda:0')
for n in (10_000, 20_000, 100_000, 500_000):
    sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
    x, adj, adj_attr, col, _ = sg.to_torch(dev)
    for _ in range(5): ops.prepare_graph(n, adj, adj_attr, col)
    torch.cuda.synchronize(); ts = []
    for _ in range(20):
        t = time.perf_counter(); ops.prepare_graph(n, adj, adj_attr, col); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"n {n}: prepare_graph median {sorted(ts)[10]:.3f} ms min {min(ts):.3f}", flush=True)
