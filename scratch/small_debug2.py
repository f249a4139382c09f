import sys, torch, numpy as np
sys.path.insert(0, '.')
from tests.test_small_layout import _forward_with_slots, small_limit
from tests.test_hip_parity import make_net
from tilingnn_amd import ops
from tilingnn_amd.synth import make_super_graph
dev = torch.device('cuda:0')
for n, ea, ec, nt in [(300, 1200, 900, 13), (300, 1800, 900, 13), (300, 1800, 900, 2), (300, 3000, 900, 13), (300, 3000, 3750, 13), (300, 2400, 900, 13)]:
    sg = make_super_graph(n, ea, ec, tile_count=2, n_edge_types=nt, seed=5)
    inputs = sg.to_torch(dev)[:4]
    net, _ = make_net(dev, depth=1)
    with small_limit(0):
        pg, s0 = _forward_with_slots(net, inputs, n, dev)
    ps, s1 = _forward_with_slots(net, inputs, n, dev)
    g = ops.prepare_graph(n, *inputs[1:])
    rp = g.adj_rowptr.cpu().numpy(); ty = g.adj_type.cpu().numpy()
    maxrank = max((np.bincount(ty[rp[v]:rp[v + 1]], minlength=1).max() if rp[v + 1] > rp[v] else 0) for v in range(n))
    print(f"ea {ea} ec {ec} types {g.n_types}: max in-degree {g.max_in_degree} max same-type edges per row {maxrank}: slot1 diff {float((s1[1] - s0[1]).abs().max()):.2e}")
