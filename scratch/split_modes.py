"""bf16 x 3 (mode 0) against fp16 x 2 (mode 1) in the general schedule: per-slot error against the float64 oracle on the
labyrinth graph (depth 4 and 20, free running), then the cached-layout forward and the per-class kernel times at 100k nodes."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from oracle import tilingnn_oracle as orc
from tests.golden_util import graph_tensors, load_labyrinth_graph
from tests.test_hip_parity import make_net
from tests.test_small_layout import _forward_with_slots, small_limit
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
g = load_labyrinth_graph()
inputs, inputs64 = graph_tensors(g, torch.float32, dev)[:4], graph_tensors(g, torch.float64)
for depth in (4, 20):
    for seed in (0, 1):
        out = {}
        for mode in (0, 1):
            _lib.lib.tgnn_set_split_precision(mode)
            net, sd = make_net(dev, depth=depth, seed=seed)
            with small_limit(0):
                probs, slots = _forward_with_slots(net, inputs, 1254, dev)
            cap = {}
            with torch.no_grad():
                want = orc.tilingnn_forward(orc.cast_sd(sd, torch.float64), *inputs64, capture=cap)[0]
            errs = [orc.rel_max_err(slots[0], cap["init"])] + [orc.rel_max_err(slots[k], cap[f"mid.{k}"]) for k in range(1, depth + 1)]
            out[mode] = (errs, float((probs.double().cpu() - want.cpu()).abs().max()))
        print(f"depth {depth} seed {seed}: max|p - p64| bf16x3 {out[0][1]:.2e} fp16x2 {out[1][1]:.2e}; slot errors (bf16x3 / fp16x2):")
        print("   " + "  ".join(f"{k}:{out[0][0][k]:.1e}/{out[1][0][k]:.1e}" for k in range(0, depth + 1, max(1, depth // 5))), flush=True)

sg = make_super_graph(100_000, 1_000_000, 1_250_000, tile_count=2, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
res = {}
for mode in (0, 1, 0, 1):
    _lib.lib.tgnn_set_split_precision(mode)
    for _ in range(5): p = net(x, adj, attr, col)[0]
    torch.cuda.synchronize(); ts = []
    for _ in range(30):
        t = time.perf_counter(); net(x, adj, attr, col); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    res[mode] = p.clone()
    print(f"mode {mode}: cached-layout forward at 100k: median {sorted(ts)[15]:.3f} ms min {min(ts):.3f}", flush=True)
print("max |p(mode 0) - p(mode 1)| at 100k:", float((res[0] - res[1]).abs().max()))
from bench import profiled_classes, in_forward_classes
for mode in (0, 1):
    _lib.lib.tgnn_set_split_precision(mode)
    profiled_classes(net, x, adj, attr, col, 2)
    cl, _ = profiled_classes(net, x, adj, attr, col, 10)
    in_forward_classes(net, x, adj, attr, col, 2)
    inf = in_forward_classes(net, x, adj, attr, col, 10)
    print(f"mode {mode} in-forward (two-stream) ms per forward: " + ", ".join(f"{k} {v['ms_per_forward']:.3f}" for k, v in inf.items()), flush=True)
    print(f"mode {mode} single-stream classes (ms per forward): " + ", ".join(f"{k} {v['ms_per_forward']:.3f}" for k, v in cl.items()), flush=True)
_lib.lib.tgnn_set_split_precision(1)
