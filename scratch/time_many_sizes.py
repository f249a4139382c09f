"""forward_many with three layouts of n nodes each, for several n: does the third co-run?"""
import sys, time, torch
sys.path.insert(0, '.')
from tests.test_hip_parity import make_net
from tilingnn_amd.synth import make_super_graph
dev = torch.device('cuda:0')
net, _ = make_net(dev, depth=20)
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return sorted(ts)[reps // 2]
for n in (320, 640, 800, 1000, 1120, 1254):
    layouts = []
    for seed in (3, 4, 5, 6):
        sg = make_super_graph(n, 7 * n, 8 * n, tile_count=2, n_edge_types=13, seed=seed)
        layouts.append(tuple(sg.to_torch(dev)[:4]))
    one = timed(lambda: net.forward_many(layouts[:1]))
    two = timed(lambda: net.forward_many(layouts[:2]))
    three = timed(lambda: net.forward_many(layouts[:3]))
    four = timed(lambda: net.forward_many(layouts[:4], streams=4))
    print(f"n {n} ({(n + 15) // 16} CUs each): one {one:.3f} | two {two:.3f} ({two / one:.2f}x) | three {three:.3f} ({three / one:.2f}x) | four on 4 streams {four:.3f} ({four / one:.2f}x)", flush=True)
