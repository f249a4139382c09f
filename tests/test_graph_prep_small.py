"""prepare_graph as one library call -- tgnn_graph_prep_small (one launch, layouts of up to 4 096 nodes) and tgnn_graph_prep (any
size: the launches queued by the library, no host round trip in the middle) -- against the separate calls: every array
bit-identical."""
import numpy as np
import pytest
import torch

from tests.golden_util import graph_tensors, load_labyrinth_graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _both(n, adj, attr, col, monkeypatch):
    from tilingnn_amd import ops
    monkeypatch.setattr(ops, "SMALL_PREP", False)
    ref = ops.prepare_graph(n, adj, attr, col)
    monkeypatch.setattr(ops, "SMALL_PREP", True)
    got = ops.prepare_graph(n, adj, attr, col)
    return ref, got


def _same(ref, got):
    assert (got.n_nodes, got.n_adj_edges, got.n_col_edges, got.n_types, got.max_in_degree) == \
        (ref.n_nodes, ref.n_adj_edges, ref.n_col_edges, ref.n_types, ref.max_in_degree)
    n, t = ref.n_nodes, ref.n_types
    na, nc = int(ref.adj_rowptr[n]), int(ref.col_rowptr[n])
    for name, cut in (("adj_rowptr", n + 1), ("adj_src", na), ("adj_eid", na), ("adj_type", na), ("edge_type", ref.n_adj_edges),
                      ("type_rep_edge", t), ("col_rowptr", n + 1), ("col_src", nc), ("col_eid", nc)):
        a, b = getattr(ref, name)[:cut].cpu(), getattr(got, name)[:cut].cpu()
        assert torch.equal(a, b), name
    assert (ref.cols is None) == (got.cols is None)
    if ref.cols is not None:
        ntiles = (n + 15) // 16
        assert torch.equal(ref.cols.tile_col_ptr[:ntiles + 1].cpu(), got.cols.tile_col_ptr[:ntiles + 1].cpu())
        ncol = int(ref.cols.tile_col_ptr[ntiles])
        assert torch.equal(ref.cols.col_meta[:ncol].cpu(), got.cols.col_meta[:ncol].cpu())
        assert torch.equal(ref.cols.col_src[:ncol * 16].cpu(), got.cols.col_src[:ncol * 16].cpu())


def test_labyrinth_layout(dev, monkeypatch):
    x, adj, attr, col, _ = graph_tensors(load_labyrinth_graph(), torch.float32, dev)
    ref, got = _both(1254, adj, attr, col, monkeypatch)
    assert got.adj_rowptr.data_ptr() != ref.adj_rowptr.data_ptr() and got.n_types == 13
    _same(ref, got)


def test_result_words_by_copy_when_the_kernel_cannot_store_them(dev, monkeypatch):
    """[r6] tgnn_graph_prep_small(..., result_host): the kernel stores the words into the pinned buffer and tgnn_graph_prep_wait polls
    it; where that is not possible (here: polling switched off in the library while the caller still hands the buffer over) the
    library copies them and the wait goes through its event -- the same graph either way, and with no buffer at all."""
    from tilingnn_amd import _lib, ops
    x, adj, attr, col, _ = graph_tensors(load_labyrinth_graph(), torch.float32, dev)
    lib = _lib.lib
    want = ops.prepare_graph(1254, adj, attr, col)                              # (kernel store + poll)
    prev = lib.tgnn_set_prep_words_poll(0)
    try:
        plain = ops.prepare_graph(1254, adj, attr, col)                         # (no host buffer: copy + stream synchronise)

        class Proxy:
            def __getattr__(self, k):
                if k == "tgnn_set_prep_words_poll":
                    return lambda v: 1 if v < 0 else lib.tgnn_set_prep_words_poll(v)
                return getattr(lib, k)
        monkeypatch.setattr(ops, "lib", Proxy())
        copied = ops.prepare_graph(1254, adj, attr, col)                        # (host buffer handed over, library copies)
    finally:
        lib.tgnn_set_prep_words_poll(prev)
    for got in (plain, copied):
        assert (got.n_types, got.n_col_edges, got.max_in_degree) == (want.n_types, want.n_col_edges, want.max_in_degree)
        assert torch.equal(got.adj_rowptr.cpu(), want.adj_rowptr.cpu()) and torch.equal(got.col_src.cpu(), want.col_src.cpu())


@pytest.mark.parametrize("n,ea,ec,t,seed", [(17, 68, 50, 13, 1), (300, 3000, 3750, 13, 2), (1000, 6800, 8350, 5, 3),
                                            (2500, 25000, 31250, 13, 4), (4096, 40960, 51200, 30, 5), (640, 6400, 8000, 60, 6)])
def test_synthetic_layouts(dev, monkeypatch, n, ea, ec, t, seed):
    from tilingnn_amd.synth import make_super_graph
    sg = make_super_graph(n, ea, ec, tile_count=2, n_edge_types=t, seed=seed)
    x, adj, attr, col, _ = sg.to_torch(dev)
    _same(*_both(n, adj, attr, col, monkeypatch))


@pytest.mark.parametrize("n,ea,ec,t,seed", [(5000, 50000, 62500, 13, 7), (20000, 200000, 250000, 13, 8), (20000, 160000, 200000, 50, 9)])
def test_larger_layouts_one_call(dev, monkeypatch, n, ea, ec, t, seed):
    from tilingnn_amd.synth import make_super_graph
    sg = make_super_graph(n, ea, ec, tile_count=2, n_edge_types=t, seed=seed)
    x, adj, attr, col, _ = sg.to_torch(dev)
    ref, got = _both(n, adj, attr, col, monkeypatch)
    assert got.adj_src.data_ptr() - got.adj_rowptr.data_ptr() == ((n + 1 + 63) // 64) * 256     # (pieces of one allocation)
    _same(ref, got)


def test_buckets_that_do_not_fit_lds_take_the_slow_path(dev, monkeypatch, debug_hooks):
    """tgnn_graph_prep sorts the edges of 512 destination rows at a time in LDS; with the threshold lowered every bucket of this
    layout goes the in-place way -- same arrays."""
    if debug_hooks:
        return                                                  # (ran against libtgnn_debug.so in a subprocess)
    from tilingnn_amd import _lib
    from tilingnn_amd.synth import make_super_graph
    n = 6000
    sg = make_super_graph(n, 60000, 75000, tile_count=2, n_edge_types=13, seed=11)
    x, adj, attr, col, _ = sg.to_torch(dev)
    before = _lib.lib.tgnn_debug_set_csr_bucket_cap(100)
    try:
        _same(*_both(n, adj, attr, col, monkeypatch))
    finally:
        _lib.lib.tgnn_debug_set_csr_bucket_cap(before)


def test_skewed_in_degrees(dev, monkeypatch):
    """A few rows that collect thousands of edges each (buckets far above the average), rows without any, self loops in the
    collision set."""
    n = 9000
    rng = np.random.default_rng(5)
    hubs = rng.integers(0, n, size=6)
    dst = np.concatenate([rng.integers(0, n, size=60000), np.repeat(hubs, 5000)])
    src = rng.integers(0, n, size=dst.size)
    adj = torch.tensor(np.stack([src, dst]), device=dev)
    col = torch.tensor(np.stack([rng.integers(0, n, size=50000), np.concatenate([rng.integers(0, 64, size=30000),
                                                                                 rng.integers(0, n, size=20000)])]), device=dev)
    attr = torch.tensor(rng.integers(0, 2, size=(dst.size, 3)).astype(np.float32), device=dev)
    _same(*_both(n, adj, attr, col, monkeypatch))


def test_self_loops_isolated_rows_and_signed_zeros(dev, monkeypatch):
    n = 40
    rng = np.random.default_rng(0)
    adj = torch.tensor(rng.integers(0, 20, size=(2, 150)), device=dev)          # rows 20 .. 39 have no in-edges
    col = torch.tensor(rng.integers(0, n, size=(2, 200)), device=dev)
    col[1, :30] = col[0, :30]                                                   # self loops: dropped from the collision set
    attr = torch.tensor(rng.integers(0, 3, size=(150, 4)).astype(np.float32), device=dev)
    attr[::7, 0] = -0.0                                                         # -0.0 == +0.0 for the de-duplication
    attr[attr == 0] = torch.where(torch.rand_like(attr[attr == 0]) < 0.5, 0.0, -0.0)
    _same(*_both(n, adj, attr, col, monkeypatch))


def test_index_errors_are_raised(dev):
    from tilingnn_amd import ops
    adj = torch.tensor([[0, 1, 2], [1, 2, 99]], device=dev)
    col = torch.tensor([[0, 1], [1, 0]], device=dev)
    attr = torch.zeros(3, 2, device=dev)
    with pytest.raises(IndexError):
        ops.prepare_graph(10, adj, attr, col)
    with pytest.raises(IndexError):
        ops.prepare_graph(10, col, attr[:2], adj)


def test_many_distinct_rows_fall_back(dev, monkeypatch):
    n, e = 500, 4000
    rng = np.random.default_rng(1)
    adj = torch.tensor(rng.integers(0, n, size=(2, e)), device=dev)
    col = torch.tensor(rng.integers(0, n, size=(2, 100)), device=dev)
    attr = torch.tensor(rng.normal(size=(e, 3)).astype(np.float32), device=dev)   # every row distinct: 4 000 "types"
    ref, got = _both(n, adj, attr, col, monkeypatch)
    assert got.n_types == e and got.cols is None
    _same(ref, got)


def test_many_distinct_rows_fall_back_above_the_small_limit(dev, monkeypatch):
    """tgnn_graph_prep numbers the edge types from a list of at most 4 096 distinct rows; beyond that it reports the fallback
    and prepare_graph goes through the separate calls."""
    n, e = 6000, 30000
    rng = np.random.default_rng(2)
    adj = torch.tensor(rng.integers(0, n, size=(2, e)), device=dev)
    col = torch.tensor(rng.integers(0, n, size=(2, 5000)), device=dev)
    attr = torch.tensor(rng.normal(size=(e, 3)).astype(np.float32), device=dev)     # every row distinct
    ref, got = _both(n, adj, attr, col, monkeypatch)
    assert got.n_types == e and got.cols is None
    _same(ref, got)
    attr2 = torch.tensor(rng.integers(0, 16, size=(e, 3)).astype(np.float32), device=dev)   # 4 096 distinct rows at most: taken
    ref, got = _both(n, adj, attr2, col, monkeypatch)
    assert got.n_types <= 4096 and got.n_types > 1000
    _same(ref, got)
