import sys, torch, numpy as np
sys.path.insert(0, '.')
from tests.test_small_layout import _forward_with_slots, _synthetic, small_limit
from tests.test_hip_parity import make_net
from tilingnn_amd import ops
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
inputs = _synthetic(n, dev)
net, _ = make_net(dev, depth=1)
with small_limit(0):
    pg, sg = _forward_with_slots(net, inputs, n, dev)
ps, ss = _forward_with_slots(net, inputs, n, dev)
d = (ss[1] - sg[1]).abs()
print("slot0 max diff", float((ss[0] - sg[0]).abs().max()), "slot1 max diff", float(d.max()), "scale", float(sg[1].abs().max()))
bad_rows = (d.max(dim=1).values > 1e-4).nonzero().flatten()
print("bad rows:", len(bad_rows), bad_rows[:40].tolist())
x, adj, attr, col = inputs
g = ops.prepare_graph(n, adj, attr, col)
deg = (g.adj_rowptr[1:n + 1] - g.adj_rowptr[:n]).cpu()
print("max in-degree", int(deg.max()), "graph.max_in_degree", g.max_in_degree, "T", g.n_types)
# per row: has a repeated type among its in-edges?
rp = g.adj_rowptr.cpu().numpy(); ty = g.adj_type.cpu().numpy()
rep = np.array([len(set(ty[rp[v]:rp[v + 1]])) < rp[v + 1] - rp[v] for v in range(n)])
print("rows with repeated types:", int(rep.sum()), "of which bad:", int(rep[bad_rows.numpy()].sum()) if len(bad_rows) else 0)
bad_ch = (d.max(dim=0).values > 1e-4).nonzero().flatten()
print("bad channels:", bad_ch.tolist())
