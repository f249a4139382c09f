import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops
from tilingnn_amd.synth import make_super_graph
dev = torch.device('cuda:0')
n = 5000
sg = make_super_graph(n, 10 * n, 12 * n, tile_count=2, n_edge_types=3, seed=2)
x, adj, adj_attr, col, _ = sg.to_torch(dev)
g = ops.prepare_graph(n, adj, adj_attr, col)
for c0, w0, r0 in ((1.25, 0.75, 0.0), (1.2345678, 0.75, 0.0), (1.25, 0.7123456, 0.0), (1.2345678, 0.7123456, 0.0), (1.25, 0.0, 0.5), (1.2345678, 0.0, 0.5),
                   (1.25, 0.0, 0.3123456), (1.2345678, 0.0, 0.3123456)):
    h = torch.full((n, 32), c0, device=dev)
    wtab = (torch.eye(32, device=dev) * w0).repeat(g.n_types, 1, 1).contiguous()
    root = torch.eye(32, device=dev) * r0
    bias = torch.zeros(32, device=dev)
    deg = (g.adj_rowptr[1:n + 1] - g.adj_rowptr[:n]).double()
    want = torch.where(deg > 0, torch.tensor(float(torch.tensor(c0)) * float(torch.tensor(w0)), dtype=torch.float64, device=dev), torch.zeros((), dtype=torch.float64, device=dev)) \
        + float(torch.tensor(c0)) * float(torch.tensor(r0))
    for k in ("cols", "cols_f16"):
        o, _ = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_NONE, ops.new_partials(32, dev), kernel=k)
        e = (o[:, 0].double() - want).abs().max() / want.abs().max()
        print(f"x {c0} w {w0} root {r0}: {k:9s} rel err {float(e):.2e}  (out[5,0] {float(o[5,0]):.9f} want {float(want[5]):.9f})")
