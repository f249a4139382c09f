import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops
from tilingnn_amd.synth import make_super_graph
from scratch.test_stream import ref64
dev = torch.device('cuda:0')
n = 5000
for T in (3, 13):
    sg = make_super_graph(n, 10 * n, 12 * n, tile_count=2, n_edge_types=T, seed=2)
    x, adj, adj_attr, col, _ = sg.to_torch(dev)
    g = ops.prepare_graph(n, adj, adj_attr, col)
    torch.manual_seed(0)
    hs = {"const h": torch.full((n, 32), 1.2345678, device=dev), "rand h": torch.randn(n, 32, device=dev), "pos h": torch.rand(n, 32, device=dev) + 0.5}
    ws = {"diag w": (torch.eye(32, device=dev) * 0.7123456).repeat(g.n_types, 1, 1).contiguous(),
          "same full w": torch.rand(1, 32, 32, device=dev).repeat(g.n_types, 1, 1).contiguous(),
          "rand w": torch.rand(g.n_types, 32, 32, device=dev)}
    rs = {"root 0": torch.zeros(32, 32, device=dev), "rand root": torch.randn(32, 32, device=dev) * 0.2}
    bias = torch.zeros(32, device=dev)
    for hn, h in hs.items():
        for wn, w in ws.items():
            for rn, r in rs.items():
                want = ref64(h, adj, g.edge_type[:10 * n].long(), w, r, bias, n)
                sc = float(want.abs().max())
                errs = []
                for k in ("cols", "cols_f16"):
                    o, _ = ops.nnconv_mean(h, g, w, r, bias, ops.ACT_LEAKY_RELU, ops.new_partials(32, dev), kernel=k)
                    errs.append(float((o.double() - want).abs().max()) / sc)
                print(f"T {g.n_types:2d} {hn:8s} {wn:12s} {rn:10s}: bf16x3 {errs[0]:.2e}  f16 {errs[1]:.2e}", flush=True)
