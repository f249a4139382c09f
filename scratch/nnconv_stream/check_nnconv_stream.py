"""The stream NNConv kernel (csrc/nnconv_stream.hip: gathered rows through an LDS ring by LDS-DMA, dense per-type products,
row sums in the epilogue) against the float64 evaluation of PyG 1.3.2's NNConv(aggr="mean") formula -- the semantics of
/root/reference/graph_networks/layers/edge_conv.py:25 -- and against the column kernel.  An experiment kept for its
measurements, not on the production path; the structure builder and the kernel are library entry
points like the others."""
import numpy as np
import pytest
import torch

from tilingnn_amd.synth import make_super_graph

pytestmark = pytest.mark.gpu


def _ref64(h, adj, etype, wtab, root, bias, n):
    h = h.double()
    src, dst = adj[0], adj[1]
    out = torch.zeros(n, 32, dtype=torch.float64, device=h.device)
    deg = torch.zeros(n, dtype=torch.float64, device=h.device)
    deg.index_add_(0, dst, torch.ones_like(dst, dtype=torch.float64))
    for t in range(wtab.shape[0]):
        m = etype == t
        if m.any():
            out.index_add_(0, dst[m], h[src[m]] @ wtab[t].double())
    out = out / deg.clamp(min=1).unsqueeze(1) + h[:n] @ root.double() + bias.double()
    return torch.where(out >= 0, out, out * 0.01)


@pytest.mark.parametrize("n,scale", [(300, 1.0), (5000, 1.0), (20000, 1e-4), (20000, 3e4), (100000, 1.0)])
def test_stream_kernel_against_fp64(n, scale):
    from tilingnn_amd import ops
    dev = torch.device("cuda:0")
    ea = 10 * n
    sg = make_super_graph(n, ea, ea // 4 * 5, tile_count=2, n_edge_types=13, seed=2)
    x, adj, adj_attr, col, _ = sg.to_torch(dev)
    g = ops.prepare_graph(n, adj, adj_attr, col)
    st = ops.build_nnconv_stream(n, ea, g.n_types, g.adj_rowptr, g.adj_src, g.adj_type)
    assert st is not None, "the benchmark's graph shape must fit the kernel's rings"
    g.stream = st
    torch.manual_seed(0)
    h = torch.randn(n, 32, device=dev) * scale            # (the fp16 pairs carry a power-of-two scale: any magnitude)
    wtab = torch.rand(g.n_types, 32, 32, device=dev)
    root = torch.randn(32, 32, device=dev) * 0.2
    bias = torch.randn(32, device=dev) * scale
    p1, p2 = ops.new_partials(32, dev), ops.new_partials(32, dev)
    o_c, np_c = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, p1, kernel="cols")
    o_s, np_s = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, p2, kernel="stream")
    want = _ref64(h, adj, g.edge_type[:ea].long(), wtab, root, bias, n)
    sc = float(want.abs().max())
    e_c = float((o_c.double() - want).abs().max()) / sc
    e_s = float((o_s.double() - want).abs().max()) / sc
    assert e_s < 1e-6 and e_c < 1e-5, (e_s, e_c)            # measured 1.0-1.3e-7 (stream), 2.4-3.0e-7 (columns)
    s_s = p2[:np_s * 64].view(np_s, 2, 32).sum(0)
    s_w = torch.stack([want.sum(0), (want * want).sum(0)])
    assert float(((s_s - s_w).abs() / s_w.abs().clamp(min=sc)).max()) < 1e-5
    o_s2, _ = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, p2, kernel="stream")
    assert torch.equal(o_s, o_s2)


def test_stream_structure_rejects_what_does_not_fit():
    from tilingnn_amd import ops
    dev = torch.device("cuda:0")
    n = 2000
    # a few rows with very high in-degree: two consecutive tiles exceed the rings
    rng = np.random.default_rng(0)
    src = rng.integers(0, n, size=40000)
    dst = np.concatenate([rng.integers(0, 32, size=20000), rng.integers(0, n, size=20000)])
    adj = torch.from_numpy(np.stack([src, dst])).long().to(dev)
    attr = torch.zeros(40000, 4, device=dev)
    attr[:, 2] = 1.0
    col = torch.zeros(2, 0, dtype=torch.long, device=dev)
    g = ops.prepare_graph(n, adj, attr, col)
    assert ops.build_nnconv_stream(n, 40000, g.n_types, g.adj_rowptr, g.adj_src, g.adj_type) is None
