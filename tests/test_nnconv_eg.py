"""NNConv over edge groups (csrc/nnconv_eg.hip, graph_prep.hip: nnconv_eg_kernel) -- the kernel the general schedule runs at
the benchmark's size.  Reference semantics: GraphConv.forward, /root/reference/graph_networks/layers/edge_conv.py:24-27 (PyG
NNConv, aggr="mean", root weight, bias) -- restated in fp64 below exactly as oracle/tilingnn_oracle.py: nnconv_mean does
(index_add of per-edge messages, mean by the in-degree clamped at 1, root term, bias).

* the structure against a numpy restatement of its definition (every in-edge exactly once, sorted by (type, row, original
  order), groups of 16, the root group last, selection masks, in-degrees), ragged sizes and rows without in-edges included;
* the op against fp64 at 37 ... 100 000 nodes, 1 ... 22 edge types, rows with more than 16 in-edges of ONE type, magnitudes
  from 1e-3 to 1e3, with and without LeakyReLU, BatchNorm partial sums included;
* tgnn_forward on groups against tgnn_forward on type columns, and the fall-back when the kernel is switched off.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def random_layout(n, ea, n_types, seed, max_type_run=0):
    """adjacency [2, ea] (int64), attribute rows with n_types distinct values; max_type_run > 0: node 3 receives that many
    in-edges of ONE type (more than a group holds), node 5 none at all."""
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (ea,), generator=g)
    dst = torch.randint(0, n, (ea,), generator=g)
    typ = torch.randint(0, n_types, (ea,), generator=g)
    if n > 8:
        dst[dst == 5] = 6                                           # a row without in-edges
        if max_type_run:
            dst[:max_type_run] = 3
            typ[:max_type_run] = typ[0]
    table = torch.rand(n_types, 4, generator=g)
    return torch.stack([src, dst]), table[typ].contiguous()


def fp64_nnconv(h, adj, edge_type, wtab, root, bias, n, leaky):
    """oracle/tilingnn_oracle.py: nnconv_mean in fp64 on the device (edge_conv.py:25; LeakyReLU of :26-27 behind it)."""
    src, dst = adj[0], adj[1]
    msg = torch.einsum("ek,eko->eo", h.double()[src], wtab.double()[edge_type])
    agg = torch.zeros(n, 32, dtype=torch.float64, device=h.device).index_add_(0, dst, msg)
    deg = torch.zeros(n, dtype=torch.float64, device=h.device).index_add_(0, dst, torch.ones_like(dst, dtype=torch.float64))
    out = agg / deg.clamp(min=1).unsqueeze(1) + h.double()[:n] @ root.double() + bias.double()
    return torch.where(out >= 0, out, out * 0.01) if leaky else out


@pytest.mark.parametrize("n,ea,n_types,run", [(37, 200, 3, 0), (1254, 9000, 13, 0), (1000, 12000, 2, 40), (4097, 30000, 22, 0)])
def test_group_structure_is_its_definition(dev, n, ea, n_types, run):
    from tilingnn_amd import ops
    adj, attr = random_layout(n, ea, n_types, seed=n, max_type_run=run)
    col = torch.zeros(2, 0, dtype=torch.int64)
    g = ops.prepare_graph(n, adj.to(dev), attr.to(dev), col.to(dev), groups=True)
    assert g.groups is not None and g.cols is None
    rowptr = g.adj_rowptr.cpu().numpy()
    csr_src = g.adj_src.cpu().numpy()
    csr_type = g.adj_type.cpu().numpy()
    tptr = g.groups.tile_grp_ptr.cpu().numpy()
    grp = g.groups.grp.cpu().numpy()
    ntiles = (n + 15) // 16
    assert tptr[0] == 0 and len(tptr) == ntiles + 1
    for tile in range(ntiles):
        rows = range(tile * 16, min(tile * 16 + 16, n))
        want = []                                                  # (type, row in tile, CSR position, source)
        for r in rows:
            for e in range(rowptr[r], rowptr[r + 1]):
                want.append((int(csr_type[e]), r - tile * 16, e, int(csr_src[e])))
        want.sort()
        groups = []
        for t in sorted({w[0] for w in want}):
            es = [w for w in want if w[0] == t]
            groups += [(t, es[i:i + 16]) for i in range(0, len(es), 16)]
        g0, g1 = int(tptr[tile]), int(tptr[tile + 1])
        assert g1 - g0 == len(groups) + 1, (tile, g1 - g0, len(groups))
        for k, (t, es) in enumerate(groups):
            rec = grp[(g0 + k) * 16:(g0 + k + 1) * 16]
            assert [int(v) for v in rec[:len(es), 0]] == [e[3] for e in es] and (rec[len(es):, 0] == -1).all()
            assert ((rec[:, 1] >> 16) == t).all()
            for j in range(16):
                mask = sum(1 << i for i, e in enumerate(es) if e[1] == j)
                assert int(rec[j, 1]) & 0xffff == mask
        root = grp[(g1 - 1) * 16:g1 * 16]
        assert ((root[:, 1] >> 16) == (g.n_types | 1 << 8)).all()
        for j in range(16):
            r = tile * 16 + j
            if r < n:
                deg = max(int(rowptr[r + 1] - rowptr[r]), 1)
                assert root[j, 0] == np.float32(deg).view(np.int32) and int(root[j, 1]) & 0xffff == 1 << j
            else:
                assert root[j, 0] == -1


@pytest.mark.parametrize("n,ea,n_types,run,scale", [(37, 200, 3, 0, 1.0), (1254, 9000, 13, 0, 1.0), (1000, 12000, 1, 40, 1.0),
                                                     (5000, 60000, 22, 0, 1.0), (10_000, 80_000, 13, 0, 1e3),
                                                     (10_000, 80_000, 13, 0, 1e-3), (10_000, 80_000, 13, 0, 8.0),
                                                     (100_000, 1_000_000, 13, 20, 1.0)])
@pytest.mark.parametrize("leaky", [False, True])
def test_op_against_fp64(dev, n, ea, n_types, run, scale, leaky):
    """Tolerance: 2e-6 of the largest output (measured 1.5e-7 .. 3e-7: three fp16-pair splits of 2^-22 each and fp32 sums), the
    bound the column kernel's fp16-pair form is held to (tests/test_hip_parity.py: nnconv_f16_pair); magnitudes 1e-3, 1 and 1e3
    take the scaled branch of the row split, 8 the unscaled one."""
    from tilingnn_amd import ops
    import oracle.tilingnn_oracle as orc
    adj, attr = random_layout(n, ea, n_types, seed=n + n_types, max_type_run=run)
    adj, attr = adj.to(dev), attr.to(dev)
    g = ops.prepare_graph(n, adj, attr, torch.zeros(2, 0, dtype=torch.int64, device=dev), groups=True)
    assert g.n_types == n_types
    gen = torch.Generator().manual_seed(1)
    h = (torch.randn(n, 32, generator=gen) * torch.randn(n, 32, generator=gen) * scale).to(dev)
    # [r6] the expectation comes from the PINNED oracle (oracle.nnconv_mean_dedup: edge_conv.py:17-18, 25 -- the edge MLP on the distinct
    # attribute rows, messages, mean, root term), not from a formula of this file: the per-type matrices are GraphConv's edge MLP
    # 4 -> 32 -> 64 -> 1024 with random weights, evaluated in fp64 by the oracle and by tgnn_edge_weight_table on the device
    prefix = "g"
    sd = {f"{prefix}.mlp.mlp.0.linear.weight": torch.randn(32, 4, generator=gen) * 0.5, f"{prefix}.mlp.mlp.0.linear.bias": torch.randn(32, generator=gen) * 0.1,
          f"{prefix}.mlp.mlp.1.linear.weight": torch.randn(64, 32, generator=gen) * 0.2, f"{prefix}.mlp.mlp.1.linear.bias": torch.randn(64, generator=gen) * 0.1,
          f"{prefix}.mlp.mlp.2.linear.weight": torch.randn(1024, 64, generator=gen) * 0.2, f"{prefix}.mlp.mlp.2.linear.bias": torch.randn(1024, generator=gen) * 0.1,
          f"{prefix}.nnConv.root": torch.randn(32, 32, generator=gen) * 0.3, f"{prefix}.nnConv.bias": torch.randn(32, generator=gen) * scale}
    mlp = [sd[f"{prefix}.mlp.mlp.{i}.linear.{k}"].to(dev) for i in range(3) for k in ("weight", "bias")]
    wtab = ops.edge_weight_table(attr, g, *mlp, 32)
    root, bias = sd[f"{prefix}.nnConv.root"].to(dev), sd[f"{prefix}.nnConv.bias"].to(dev)
    with torch.no_grad():
        want = orc.nnconv_mean_dedup(h.double().cpu(), adj.cpu(), attr.double().cpu(), orc.cast_sd(sd, torch.float64), prefix)
    if leaky:
        want = orc.leaky_relu(want)
    want = want.to(dev)
    # (this file's own restatement, on the device's table: the two expectations agree to the table's fp32 rounding)
    assert orc.rel_max_err(fp64_nnconv(h, adj, g.edge_type[:ea].long(), wtab, root, bias, n, leaky).cpu(), want.cpu()) < 1e-6
    part = ops.new_partials(32, dev)
    out, npart = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU if leaky else ops.ACT_NONE, part, kernel="eg")
    assert out.shape == (n, 32) and bool(torch.isfinite(out).all())
    assert orc.rel_max_err(out.cpu(), want.cpu()) < 2e-6
    # BatchNorm partial rows: [blocks][sum 32 | sum of squares 32] in fp64
    p = part[:npart * 64].view(npart, 64).sum(0)
    assert torch.allclose(p[:32].cpu(), want.sum(0).cpu(), rtol=1e-6, atol=1e-6 * float(want.abs().max()) * n)
    assert torch.allclose(p[32:].cpu(), (want * want).sum(0).cpu(), rtol=1e-5, atol=1e-6 * float(want.abs().max()) ** 2 * n)
    # ... and the default per-op entry point takes the same kernel on a layout that carries groups: the same bits
    out2, _ = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU if leaky else ops.ACT_NONE, ops.new_partials(32, dev))
    assert torch.equal(out, out2)


def test_halo_rows_behind_the_destinations(dev):
    """A shard's layout: sources index rows behind the n destination rows (n_src_nodes > n_nodes)."""
    from tilingnn_amd import ops
    import oracle.tilingnn_oracle as orc
    n, n_src, ea = 3000, 3700, 30000
    gen = torch.Generator().manual_seed(4)
    adj = torch.stack([torch.randint(0, n_src, (ea,), generator=gen), torch.randint(0, n, (ea,), generator=gen)]).to(dev)
    table = torch.rand(7, 3, generator=gen)
    attr = table[torch.randint(0, 7, (ea,), generator=gen)].contiguous().to(dev)
    g = ops.prepare_graph(n, adj, attr, torch.zeros(2, 0, dtype=torch.int64, device=dev), n_src_nodes=n_src)
    assert g.groups is not None                                    # (a shard always runs the general schedule)
    h = torch.randn(n_src, 32, generator=gen).to(dev)
    wtab = torch.rand(g.n_types, 32, 32, generator=gen).to(dev)
    root = (torch.randn(32, 32, generator=gen) * 0.3).to(dev)
    bias = torch.randn(32, generator=gen).to(dev)
    want = fp64_nnconv(h, adj, g.edge_type[:ea].long(), wtab, root, bias, n, False)
    out, _ = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_NONE, ops.new_partials(32, dev), kernel="eg")
    assert orc.rel_max_err(out.cpu(), want.cpu()) < 2e-6


def test_forward_on_groups_against_columns_and_fallback(dev):
    """A layout of the general schedule (above the mid-size limit): the forward over edge groups, over type columns
    (tgnn_set_nnconv_eg(0), prepare_graph(groups=False)) and -- groups only, kernel switched off -- over the CSR kernel.
    20 chaotic layers: the three agree as closely as two roundings of the same network do (measured 7e-6 at 100 000 nodes)."""
    from tilingnn_amd import TilinGNN, ops
    from tilingnn_amd._lib import lib
    from tilingnn_amd.synth import make_super_graph
    from tilingnn_amd.weights import make_state_dict
    n = 40_000
    sg = make_super_graph(n, 10 * n, 12 * n, tile_count=2, n_edge_types=13, seed=3)
    x, adj, attr, col, _ = sg.to_torch(dev)
    net = TilinGNN(15, 20, 32, node_features_dim=3)
    net.load_state_dict(make_state_dict(15, 20, 32, 1, 3))
    net = net.to(dev).train()
    net.cache_graph = False
    assert ops.prepare_graph(n, adj, attr, col).groups is not None
    p_groups = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].clone()
    prev = lib.tgnn_set_nnconv_eg(0)
    try:
        p_csr = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].clone()      # groups only, kernel off
        ops.GROUPS, keep = False, ops.GROUPS
        try:
            assert ops.prepare_graph(n, adj, attr, col).cols is not None
            p_cols = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].clone()
        finally:
            ops.GROUPS = keep
    finally:
        lib.tgnn_set_nnconv_eg(prev)
    assert bool(torch.isfinite(p_groups).all())
    assert float((p_groups - p_cols).abs().max()) < 2e-4
    assert float((p_groups - p_csr).abs().max()) < 2e-4


@pytest.mark.parametrize("n,ea,n_types,run", [(37, 200, 3, 0), (1254, 9000, 13, 0), (1000, 12000, 1, 40), (100_000, 1_000_000, 13, 20)])
@pytest.mark.parametrize("leaky", [False, True])
def test_bf16_width64_op_on_groups(dev, n, ea, n_types, run, leaky):
    """BASELINE config 3's NNConv (width 64, bf16 storage; csrc/bf16_path.hip: nnconv64_bf16_eg_kernel) over edge groups: against
    fp64 on the bf16-rounded operands within the path's stated 2^-7 of the output's max-norm (tests/test_bf16_path.py), and
    against the type-column kernel of the same path (both outputs are bf16: up to an ulp, 2^-7 of the max-norm, apart); the
    BatchNorm sums are those of the stored (rounded) values."""
    from tilingnn_amd import ops, ops_bf16
    import oracle.tilingnn_oracle as orc
    adj, attr = random_layout(n, ea, n_types, seed=n + 7, max_type_run=run)
    adj, attr = adj.to(dev), attr.to(dev)
    g = ops.prepare_graph(n, adj, attr, torch.zeros(2, 0, dtype=torch.int64, device=dev), groups=True)
    gen = torch.Generator().manual_seed(2)
    h = torch.randn(n, 64, generator=gen).to(dev).to(torch.bfloat16)
    wtab = torch.rand(g.n_types, 64, 64, generator=gen).to(dev)
    root = (torch.randn(64, 64, generator=gen) * 0.3).to(dev)
    bias = torch.randn(64, generator=gen).to(dev)
    act = ops.ACT_LEAKY_RELU if leaky else ops.ACT_NONE
    src, dst = adj[0], adj[1]
    et = g.edge_type[:ea].long()
    wb, rb = wtab.to(torch.bfloat16).double(), root.to(torch.bfloat16).double()       # the path rounds its weights once
    msg = torch.einsum("ek,eko->eo", h.double()[src], wb[et])
    agg = torch.zeros(n, 64, dtype=torch.float64, device=dev).index_add_(0, dst, msg)
    deg = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, dst, torch.ones_like(dst, dtype=torch.float64))
    want = agg / deg.clamp(min=1).unsqueeze(1) + h.double() @ rb + bias.double()
    if leaky:
        want = torch.where(want >= 0, want, want * 0.01)
    part = ops.new_partials(64, dev)
    out, npart = ops_bf16.nnconv64(h, g, wtab, root, bias, act, part, kernel="eg")
    assert out.dtype == torch.bfloat16 and out.shape == (n, 64) and bool(torch.isfinite(out.float()).all())
    assert orc.rel_max_err(out.float().cpu(), want.cpu()) < 2.0 ** -7
    out_c, _ = ops_bf16.nnconv64(h, g, wtab, root, bias, act, ops.new_partials(64, dev), kernel="cols")
    assert orc.rel_max_err(out.float().cpu(), out_c.double().cpu()) < 2.0 ** -7      # (two roundings to bf16: up to an ulp apart)
    p = part[:npart * 128].view(npart, 128).sum(0)
    stored = out.double()
    assert float((p[:64] - stored.sum(0)).abs().max()) < 1e-9 * float(stored.abs().sum(0).max())
    assert float((p[64:] - (stored * stored).sum(0)).abs().max()) < 1e-9 * float((stored * stored).sum(0).max())
    out_d, _ = ops_bf16.nnconv64(h, g, wtab, root, bias, act, ops.new_partials(64, dev))    # a layout with groups: the default
    assert torch.equal(out, out_d)


def test_in_degree_beyond_the_kernels_limit_takes_another_kernel(dev):
    """The edge-group kernel folds the root term with S = diag(max(deg, 1)) in fp16: exact up to 2 048.  A layout of the general
    schedule with a hub of 3 000 in-edges is prepared with groups like any other, tgnn_forward sees nn_max_in_degree and keeps
    off the kernel (CSR kernel: the layout carries no columns); the per-op entry point builds the columns and takes those.
    Both against the forward / the op on type columns."""
    from tilingnn_amd import TilinGNN, ops
    from tilingnn_amd._lib import lib
    from tilingnn_amd.synth import make_super_graph
    from tilingnn_amd.weights import make_state_dict
    import oracle.tilingnn_oracle as orc
    n = 40_000
    sg = make_super_graph(n, 10 * n, 12 * n, tile_count=2, n_edge_types=13, seed=4)
    x, adj, attr, col, _ = sg.to_torch(dev)
    adj = adj.clone()
    adj[1, :3000] = 123
    g = ops.prepare_graph(n, adj, attr, col)
    assert g.groups is not None and g.max_in_degree >= 3000
    gen = torch.Generator().manual_seed(5)
    h = torch.randn(n, 32, generator=gen).to(dev)
    wtab = torch.rand(g.n_types, 32, 32, generator=gen).to(dev)
    root = (torch.randn(32, 32, generator=gen) * 0.3).to(dev)
    bias = torch.randn(32, generator=gen).to(dev)
    want = fp64_nnconv(h, adj, g.edge_type[:adj.shape[1]].long(), wtab, root, bias, n, True)
    out, _ = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, ops.new_partials(32, dev))     # (columns, built on first use)
    assert orc.rel_max_err(out.cpu(), want.cpu()) < 2e-6
    net = TilinGNN(15, 6, 32, node_features_dim=3)
    net.load_state_dict(make_state_dict(15, 6, 32, 1, 3))
    net = net.to(dev).train()
    net.cache_graph = False
    p_groups_graph = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].clone()
    keep, ops.GROUPS = ops.GROUPS, False
    try:
        p_cols = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].clone()
    finally:
        ops.GROUPS = keep
    assert bool(torch.isfinite(p_groups_graph).all()) and float((p_groups_graph - p_cols).abs().max()) < 2e-4


def test_one_preparation_call_builds_both_structures(dev):
    """tgnn_graph_prep handed the buffers of BOTH NNConv structures (type columns + mid-size batches, edge groups) builds both on
    the same scratch, one behind the other: bit for bit what the calls that build only one of them leave."""
    from tilingnn_amd import ops
    from tilingnn_amd.synth import make_super_graph
    n = 9000
    sg = make_super_graph(n, 9 * n, 11 * n, tile_count=2, n_edge_types=13, seed=6)
    x, adj, attr, col, _ = sg.to_torch(dev)
    both = ops.prepare_graph(n, adj, attr, col, groups="both")
    only_c = ops.prepare_graph(n, adj, attr, col, groups=False)
    only_g = ops.prepare_graph(n, adj, attr, col, groups=True)
    assert both.cols is not None and both.groups is not None and both.mid is not None
    assert only_c.groups is None and only_g.cols is None
    ntiles = (n + 15) // 16
    ncol = int(only_c.cols.tile_col_ptr[ntiles])
    assert torch.equal(both.cols.tile_col_ptr[:ntiles + 1], only_c.cols.tile_col_ptr[:ntiles + 1])
    assert torch.equal(both.cols.col_meta[:ncol], only_c.cols.col_meta[:ncol])
    assert torch.equal(both.cols.col_src[:ncol * 16], only_c.cols.col_src[:ncol * 16])
    ng = int(only_g.groups.tile_grp_ptr[ntiles])
    assert torch.equal(both.groups.tile_grp_ptr[:ntiles + 1], only_g.groups.tile_grp_ptr[:ntiles + 1])
    assert torch.equal(both.groups.grp[:ng * 16], only_g.groups.grp[:ng * 16])
    assert torch.equal(both.adj_type[:both.n_adj_edges], only_g.adj_type[:only_g.n_adj_edges])
    # the mid-size batches: 24 x 36 words per tile, of which the build writes a tile's first tile_nb batches (the words behind
    # them are never read and are whatever the allocation held)
    assert torch.equal(both.mid.tile_nb, only_c.mid.tile_nb)
    nb = only_c.mid.tile_nb[:ntiles].long()
    written = (torch.arange(24 * 36, device=dev).unsqueeze(0) < (nb * 36).unsqueeze(1))
    a = both.mid.ent[:ntiles * 24 * 36].view(ntiles, 24 * 36)
    b = only_c.mid.ent[:ntiles * 24 * 36].view(ntiles, 24 * 36)
    assert int(nb.max()) > 0 and torch.equal(a[written], b[written])
