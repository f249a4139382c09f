"""forward_many: three labyrinth-sized layouts side by side against one (host-synchronised wall time per call)."""
import sys, time, torch
sys.path.insert(0, '.')
from tests.golden_util import graph_tensors, load_labyrinth_graph
from tests.test_hip_parity import make_net
from tilingnn_amd.synth import make_super_graph
dev = torch.device('cuda:0')
net, _ = make_net(dev, depth=20)
g = load_labyrinth_graph()
layouts = [tuple(graph_tensors(g, torch.float32, dev)[:4])]
for n, seed in ((1254, 3), (1254, 4)):
    sg = make_super_graph(n, 8502, 10472, tile_count=2, n_edge_types=13, seed=seed)
    layouts.append(tuple(sg.to_torch(dev)[:4]))
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return sorted(ts)[reps // 2]
for cache in (True, False):
    net.cache_graph = cache
    one = timed(lambda: net(x=layouts[0][0], adj_e_index=layouts[0][1], adj_e_features=layouts[0][2], col_e_idx=layouts[0][3]))
    seq = timed(lambda: [net(x=l[0], adj_e_index=l[1], adj_e_features=l[2], col_e_idx=l[3]) for l in layouts])
    many = timed(lambda: net.forward_many(layouts))
    print(f"cached layout {cache}: one {one:.3f} ms | three one after the other {seq:.3f} ms | three side by side {many:.3f} ms ({many / one:.2f}x one)")
net.cache_graph = True
for k in (1, 2, 3, 4, 6):
    ls = (layouts * 2)[:k]
    for st in (1, 2, 3):
        t = timed(lambda: net.forward_many(ls, streams=st))
        print(f"forward_many of {k} layouts on {st} stream(s): {t:.3f} ms", flush=True)
