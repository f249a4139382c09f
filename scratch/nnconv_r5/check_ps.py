"""The LDS-streaming NNConv (csrc/nnconv_ps.hip) against a float64 torch restatement and against the column kernel, + timings.
usage: python scratch/check_ps.py [n] [ea]"""
import sys, os, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops
from tilingnn_amd.synth import make_super_graph

KERN = os.environ.get("PS_KERN", "cols_ps")
dev = torch.device('cuda:0')


def ref64(h, adj, etype, wtab, root, bias, n):
    h = h.double().cpu(); wt = wtab.double().cpu(); adj = adj.cpu(); etype = etype.cpu().long()
    src, dst = adj[0], adj[1]
    out = torch.zeros(n, 32, dtype=torch.float64)
    cnt = torch.zeros(n, dtype=torch.float64)
    for t in range(wt.shape[0]):
        m = etype == t
        if m.any():
            out.index_add_(0, dst[m], h[src[m]] @ wt[t])
    cnt.index_add_(0, dst, torch.ones(dst.shape[0], dtype=torch.float64))
    out = out / cnt.clamp(min=1)[:, None] + h[:n] @ root.double().cpu() + bias.double().cpu()
    return torch.nn.functional.leaky_relu(out, 0.01)


def run(n, ea, T=13, time_it=True):
    sg = make_super_graph(n, ea, ea // 4 * 5, tile_count=2, n_edge_types=T, seed=2)
    x, adj, adj_attr, col, _ = sg.to_torch(dev)
    g = ops.prepare_graph(n, adj, adj_attr, col)
    torch.manual_seed(0)
    h = torch.randn(n, 32, device=dev) * 3
    wtab = torch.rand(g.n_types, 32, 32, device=dev)
    root = torch.randn(32, 32, device=dev) * 0.2
    bias = torch.randn(32, device=dev)
    part = ops.new_partials(32, dev)
    part2 = ops.new_partials(32, dev)
    o_ps, np_ps = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, part, kernel=KERN)
    o_c, np_c = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, part2, kernel="cols_f16")
    torch.cuda.synchronize()
    want = ref64(h, adj, g.edge_type, wtab, root, bias, n)
    sc = want.abs().max()
    e_ps = float((o_ps.double().cpu() - want).abs().max() / sc)
    e_c = float((o_c.double().cpu() - want).abs().max() / sc)
    s_ps = ops.bn_sums(part, np_ps, 32).cpu(); s_c = ops.bn_sums(part2, np_c, 32).cpu()
    s_ref = torch.cat([want.sum(0), (want * want).sum(0)])
    e_bn = float(((s_ps.reshape(-1) - s_ref).abs() / s_ref.abs().clamp(min=1)).max())
    print(f"n={n} ea={ea} T={g.n_types}: max-norm rel err vs fp64: ps {e_ps:.2e}  cols_f16 {e_c:.2e}   bn sums rel {e_bn:.2e}  finite {bool(torch.isfinite(o_ps).all())}")
    # repeatability
    o2, _ = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, part, kernel=KERN)
    print("  bit-repeatable:", bool(torch.equal(o_ps, o2)))
    if not time_it:
        return
    for kern in ("cols_ps", "ps", "cols_f16"):
        for _ in range(5):
            ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, part, kernel=kern)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, part, kernel=kern)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"  {kern:9s} us/call (incl. bounds + split / weight image launches): min {min(ts):.1f} median {sorted(ts)[2]:.1f}")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 10 * int(sys.argv[1]))
    else:
        run(50, 200, T=3, time_it=False)
        run(1000, 8000, time_it=False)
        run(10_000, 80_000, time_it=False)
        run(100_000, 1_000_000)
