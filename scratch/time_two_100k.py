"""Two independent 100 000-node layouts side by side (each with its own pair of streams) against one after the other: does a second
forward fill the dependency gaps of the first?  GPU_MAX_HW_QUEUES=8 python scratch/time_two_100k.py [n]"""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lay, nets = [], []
for k in range(K):
    sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=1 + k)
    x, adj, attr, col, _ = sg.to_torch(dev)
    lay.append((x, adj, attr, col))
    net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
    net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0), strict=True)
    nets.append(net.to(dev).train())
lanes = _lib.concurrent_streams(dev, 2 * K)
print("distinct streams:", len({s.cuda_stream for s in lanes}), "of", 2 * K)
cur_side = [None]
_lib.side_stream = lambda device: C.c_void_p(cur_side[0].cuda_stream)
cur = torch.cuda.current_stream(dev)


def one_after_the_other(reps):
    cur_side[0] = lanes[1]
    for _ in range(reps):
        for k in range(K):
            nets[k](x=lay[k][0], adj_e_index=lay[k][1], adj_e_features=lay[k][2], col_e_idx=lay[k][3])


def side_by_side(reps):
    for k in range(K):
        lanes[2 * k].wait_stream(cur)
    for _ in range(reps):
        for k in range(K):
            cur_side[0] = lanes[2 * k + 1]
            with torch.cuda.stream(lanes[2 * k]):
                nets[k](x=lay[k][0], adj_e_index=lay[k][1], adj_e_features=lay[k][2], col_e_idx=lay[k][3])
    for k in range(K):
        cur.wait_stream(lanes[2 * k])


for name, fn in (("one after the other", one_after_the_other), ("side by side", side_by_side), ("one after the other", one_after_the_other),
                 ("side by side", side_by_side)):
    fn(3)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 20
    fn(reps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print(f"{K} x {n} nodes, {name}: {dt * 1e3:.3f} ms per {K} forwards = {K * n / dt / 1e6:.1f} M nodes/s (graph prep cached: "
          f"{getattr(nets[0], 'cache_graph', None)})")
