"""Per-layer error of both schedules against the float64 oracle, free running, depth 20, real labyrinth graph:
max-norm relative error of every skip-buffer slot, for a few weight seeds."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import tilingnn_oracle as orc
from tests.golden_util import graph_tensors, load_labyrinth_graph
from tests.test_hip_parity import make_net
from tests.test_small_layout import _forward_with_slots, small_limit
dev = torch.device('cuda:0')
g = load_labyrinth_graph()
inputs, inputs64 = graph_tensors(g, torch.float32, dev)[:4], graph_tensors(g, torch.float64)
for seed in (0, 1, 2):
    rows = {}
    for name, limit in (("general", 0), ("persistent", 4096)):
        net, sd = make_net(dev, depth=20, seed=seed)
        with small_limit(limit):
            probs, slots = _forward_with_slots(net, inputs, 1254, dev)
        cap = {}
        with torch.no_grad():
            want = orc.tilingnn_forward(orc.cast_sd(sd, torch.float64), *inputs64, capture=cap)[0]
        errs = [orc.rel_max_err(slots[0], cap["init"])] + [orc.rel_max_err(slots[k], cap[f"mid.{k}"]) for k in range(1, 21)]
        rows[name] = (errs, float((probs.double() - want.cpu()).abs().max()))
    print(f"seed {seed}: max|p - p64| general {rows['general'][1]:.2e} persistent {rows['persistent'][1]:.2e}")
    print("  slot  general   persistent")
    for k in range(21):
        print(f"  {k:3d}  {rows['general'][0][k]:.2e}  {rows['persistent'][0][k]:.2e}")
