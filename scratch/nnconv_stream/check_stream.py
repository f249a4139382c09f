"""Stream NNConv kernel vs the column kernel vs a float64 torch evaluation, then timing of both (HIP events).
argv: sizes (default 300 5000 20000 100000)."""
import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops
from tilingnn_amd.synth import make_super_graph

dev = torch.device('cuda:0')
sizes = [int(a) for a in sys.argv[1:]] or [300, 5000, 20000, 100000]


def ref64(h, adj, etype, wtab, root, bias, n):
    h = h.double(); src, dst = adj[0], adj[1]
    out = torch.zeros(n, 32, dtype=torch.float64, device=h.device)
    deg = torch.zeros(n, dtype=torch.float64, device=h.device)
    deg.index_add_(0, dst, torch.ones_like(dst, dtype=torch.float64))
    for t in range(wtab.shape[0]):
        m = etype == t
        if m.any():
            msg = h[src[m]] @ wtab[t].double()
            out.index_add_(0, dst[m], msg)
    out = out / deg.clamp(min=1).unsqueeze(1) + h[:n] @ root.double() + bias.double()
    return torch.where(out >= 0, out, out * 0.01)


for n in (sizes if __name__ == "__main__" else []):
    ea = 10 * n
    sg = make_super_graph(n, ea, ea // 4 * 5, tile_count=2, n_edge_types=13, seed=2)
    x, adj, adj_attr, col, _ = sg.to_torch(dev)
    g = ops.prepare_graph(n, adj, adj_attr, col)
    if g.stream is None:
        g.stream = ops.build_nnconv_stream(n, ea, g.n_types, g.adj_rowptr, g.adj_src, g.adj_type)
    print(f"n={n}: types {g.n_types}, stream {'yes' if g.stream is not None else 'NO'}", flush=True)
    if g.stream is None:
        continue
    torch.manual_seed(0)
    h = torch.randn(n, 32, device=dev)
    wtab = torch.rand(g.n_types, 32, 32, device=dev)
    root = torch.randn(32, 32, device=dev) * 0.2
    bias = torch.randn(32, device=dev)
    p1, p2 = ops.new_partials(32, dev), ops.new_partials(32, dev)
    o_c, np_c = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, p1, kernel="cols")
    o_s, np_s = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, p2, kernel="stream")
    torch.cuda.synchronize()
    want = ref64(h, adj, g.edge_type[:ea].long(), wtab, root, bias, n)
    sc = float(want.abs().max())
    e_c, e_s = float((o_c.double() - want).abs().max()) / sc, float((o_s.double() - want).abs().max()) / sc
    s_c = p1[:np_c * 64].view(np_c, 2, 32).sum(0)
    s_s = p2[:np_s * 64].view(np_s, 2, 32).sum(0)
    s_w = torch.stack([want.sum(0), (want * want).sum(0)])
    print(f"   rel err vs fp64: cols {e_c:.2e}  stream {e_s:.2e};  BN sums rel: cols {float(((s_c - s_w).abs() / s_w.abs().clamp(min=1)).max()):.2e} "
          f"stream {float(((s_s - s_w).abs() / s_w.abs().clamp(min=1)).max()):.2e}  partial rows {np_c}/{np_s}", flush=True)
    bad = (o_s.double() - want).abs().max(dim=1).values
    if e_s > 1e-5:
        idx = torch.nonzero(bad > 1e-5 * sc).flatten()
        print("   BAD rows:", idx[:20].tolist(), "count", idx.numel(), flush=True)
    # repeatability
    o_s2, _ = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, p2, kernel="stream")
    print("   bit-repeatable:", bool(torch.equal(o_s, o_s2)), flush=True)
    for name in ("cols", "stream"):
        for _ in range(5):
            ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, p1, kernel=name)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, p1, kernel=name)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"   {name:7s} us/call (incl. weight image ~5 us + alloc): min {min(ts):.1f} median {sorted(ts)[2]:.1f}", flush=True)
