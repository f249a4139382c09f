"""Column NNConv kernel, bf16 x 3 against fp16 x 2 (true / inflated in-degree bound) against a float64 evaluation; inputs with a
heavy tail (product of two normals, as the skip buffer's rows are)."""
import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops
from tilingnn_amd.synth import make_super_graph
from scratch.test_stream import ref64
dev = torch.device('cuda:0')
for n in (6000, 100000):
    ea = 10 * n
    sg = make_super_graph(n, ea, ea // 4 * 5, tile_count=2, n_edge_types=13, seed=2)
    x, adj, adj_attr, col, _ = sg.to_torch(dev)
    g = ops.prepare_graph(n, adj, adj_attr, col)
    torch.manual_seed(0)
    for tail in (False, True):
        h = torch.randn(n, 32, device=dev)
        if tail: h = h * torch.randn(n, 32, device=dev) + 0.3 * torch.randn(n, 32, device=dev)
        wtab = torch.rand(g.n_types, 32, 32, device=dev)
        root = torch.randn(32, 32, device=dev) * 0.2
        bias = torch.randn(32, device=dev)
        want = ref64(h, adj, g.edge_type[:ea].long(), wtab, root, bias, n)
        sc = float(want.abs().max())
        res = {}
        for name, kw in (("bf16x3", dict(kernel="cols")), ("f16", dict(kernel="cols_f16")), ("f16 x4", dict(kernel="cols_f16", max_in_degree=4 * g.max_in_degree)),
                         ("f16 x64", dict(kernel="cols_f16", max_in_degree=64 * g.max_in_degree))):
            p = ops.new_partials(32, dev)
            o, _ = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, p, **kw)
            res[name] = o
        torch.cuda.synchronize()
        print(f"n {n} tail {tail} max|h| {float(h.abs().max()):.1f} maxdeg {g.max_in_degree}: " + ", ".join(
            f"{k} {float((v.double() - want).abs().max()) / sc:.2e}" for k, v in res.items()) +
            f" | f16 x4 vs f16 {float((res['f16 x4'] - res['f16']).abs().max()) / sc:.2e}", flush=True)
