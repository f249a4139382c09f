"""Layout producer timings: pickle vs side-car load, host producer, device producer.
    python scratch/time_producer.py [n_tiles]   (GPU box)"""
import os, sys, time, shutil, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd.tiling.tile_graph import TileGraph, GraphArrays
from tilingnn_amd.util import data_util as du

def best(f, n=5):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts)

small = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "complete_graph_small.pkl")
d = tempfile.mkdtemp()
p = os.path.join(d, "cg.pkl"); shutil.copy(small, p)
def load(sidecar):
    g = TileGraph(2); g.load_graph_state(p, sidecar=sidecar); g.arrays; return g
print("150-tile fixture: pickle load + arrays %.2f ms" % (1e3 * best(lambda: load(False))))
load(True)
print("150-tile fixture: side-car load        %.2f ms" % (1e3 * best(lambda: load(True))))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rng = np.random.default_rng(0)
def edges(m):
    u = rng.integers(0, n, m); v = (u + rng.integers(1, 40, m)) % n
    return np.stack([np.concatenate([u, v]), np.concatenate([v, u])])
adj, col = edges(4 * n), edges(5 * n)
T = 13
adj_type = rng.integers(0, T, adj.shape[1]).astype(np.int32)
adjf = np.zeros((adj.shape[1], 2 + T)); adjf[np.arange(adj.shape[1]), 2 + adj_type] = 1; adjf[:, 1] = 0.1 + 0.05 * adj_type
colf = np.zeros((col.shape[1], 2 + T)); colf[:, 0] = 0.3
g = TileGraph(2)
g._arrays = GraphArrays(rng.integers(0, 2, n), 0.5 + 0.5 * rng.random(n), col, adj, colf, adjf, adj_type, 1.0, 1.0, 2)
tiles = np.flatnonzero(rng.random(n) < 0.5)
print(f"synthetic complete graph: {n} tiles, {adj.shape[1]} adjacency + {col.shape[1]} collision edges; super set {tiles.size}")
t_host = best(lambda: du.create_brick_layout_from_super_set(g, tiles), 3)
print("host producer (numpy, this package)      %.1f ms" % (1e3 * t_host))
def host_and_upload():
    x, ci, cf, ai, af, _ = du.create_brick_layout_from_super_set(g, tiles)
    du.to_torch_tensor("cuda:0", x, ai, af, ci, cf); torch.cuda.synchronize()
print("host producer + to_torch_tensor upload   %.1f ms" % (1e3 * best(host_and_upload, 3)))
cg = du.CompleteGraphOnDevice(g, "cuda:0")
def dev():
    cg.layout(tiles); torch.cuda.synchronize()
dev()
print("device producer (mask upload + compact)  %.2f ms" % (1e3 * best(dev, 10)))
alive = torch.zeros(n, dtype=torch.int32, device="cuda:0"); alive[torch.from_numpy(tiles).cuda()] = 1
def dev2():
    cg._builder.build(alive); torch.cuda.synchronize()
print("device producer (mask already resident)  %.2f ms" % (1e3 * best(dev2, 10)))
shutil.rmtree(d)
