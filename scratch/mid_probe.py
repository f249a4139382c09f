import sys, torch
sys.path.insert(0, ".")
from tests.test_mid_layout import _layout
from tests.test_hip_parity import make_net
from tests.test_small_layout import _forward_with_slots
from tilingnn_amd import _lib, ops
dev = torch.device("cuda:0")
for n in (8000, 20000, 50000, 65536):
    inputs, _ = _layout(n, dev)
    g = ops.prepare_graph(n, *inputs[1:])
    nbmax = int(g.mid.tile_nb.max()) if g.mid is not None else -1
    print(n, "mid", g.mid is not None, "maxdeg", g.max_in_degree, "types", g.n_types, "max batches", nbmax, "mean", float(g.mid.tile_nb.float().mean()) if g.mid is not None else 0, flush=True)
n = 8000
inputs, _ = _layout(n, dev)
net, _ = make_net(dev, depth=3)
p, s = _forward_with_slots(net, inputs, n, dev)
_lib.lib.tgnn_set_mid_layout_limit(0)
p2, s2 = _forward_with_slots(net, inputs, n, dev)
print("equal", torch.equal(p, p2), [float((s[k] - s2[k]).abs().max()) for k in range(4)])
