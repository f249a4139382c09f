"""Forward latency on small layouts (the sizes the greedy loop actually sees): the persistent small-layout kernel
(csrc/forward_small.hip) against the general launch schedule, cached layout, plus the real labyrinth graph."""
import sys, time, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev)
sizes = [int(a) for a in sys.argv[1:]] or [300, 1254, 2500, 4096]


def timed(inputs, reps=100):
    for _ in range(5): net(*inputs)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): net(*inputs)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


cases = []
try:
    from tests.golden_util import graph_tensors, load_labyrinth_graph
    cases.append(("labyrinth", graph_tensors(load_labyrinth_graph(), torch.float32, dev)[:4]))
except Exception as exc:                                   # noqa: BLE001
    print("labyrinth graph not available:", exc)
for n in sizes:
    sg = make_super_graph(n, int(n * 6.8), int(n * 8.35), tile_count=2, n_edge_types=13, seed=2)
    cases.append((f"N={n}", sg.to_torch(dev)[:4]))
for name, inputs in cases:
    row = []
    for limit in (0, 4096):
        _lib.lib.tgnn_set_small_layout_limit(limit)
        row.append(timed(inputs))
    print(f"{name:>10s}: general {row[0]:.3f} ms   small-layout kernel {row[1]:.3f} ms")
