"""Small layouts: the persistent whole-layer-loop kernel (csrc/forward_small.hip) against the general launch schedule
(same formulas, different association of the BatchNorm / tile sums) and against the fp64 oracle."""
import contextlib
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import tilingnn_oracle as orc
from tests.golden_util import graph_tensors, load_labyrinth_graph, load_npz
from tests.test_hip_parity import make_net

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@contextlib.contextmanager
def small_limit(n):
    from tilingnn_amd import _lib
    before = _lib.lib.tgnn_get_small_layout_limit()
    _lib.lib.tgnn_set_small_layout_limit(n)
    try:
        yield
    finally:
        _lib.lib.tgnn_set_small_layout_limit(before)


def _forward_with_slots(net, inputs, n, dev, update_running=0):
    """tgnn_forward on a zeroed workspace; returns (probs, skip-buffer slots [D + 1, n, 32]) -- the slots are the first
    carve of the workspace (csrc/forward.hip: carve)."""
    from tilingnn_amd import _lib, ops
    x, adj, adj_attr, col = inputs
    graph = ops.prepare_graph(n, adj, adj_attr, col)
    dims = net._dims()
    table, _ = net._param_table()
    ws_bytes = _lib.lib.tgnn_forward_workspace_bytes(C.byref(dims), n, graph.n_types)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
    probs = torch.empty(n, 1, device=dev)
    g = graph.c_struct()
    _lib.check(_lib.lib.tgnn_forward(C.byref(dims), table, ops.ptr(x), ops.ptr(adj_attr), C.byref(g), update_running, 0,
                                    ops.ptr(probs), ops.ptr(ws), ws_bytes, _lib.current_stream(dev), _lib.side_stream(dev)))
    torch.cuda.synchronize()
    d = net.network_depth
    slots = ws[: (d + 1) * n * 32 * 4].view(torch.float32).view(d + 1, n, 32).clone()
    return probs.cpu(), slots.cpu()


def _synthetic(n, dev, seed=5):
    from tilingnn_amd.synth import make_super_graph
    ea, ec = (10 * n, 12 * n + n // 2) if n >= 100 else (4 * n, 3 * n)
    sg = make_super_graph(n, ea, ec, tile_count=2, n_edge_types=13, seed=seed)
    x, adj, adj_attr, col, _ = sg.to_torch(dev)
    return (x, adj, adj_attr, col)


@pytest.mark.parametrize("n", [17, 300, 1254, 2000, 3333, 4096])
def test_small_kernel_matches_the_general_schedule_layer_by_layer(dev, n):
    """Depth 3 (residual, two-deep collision buffers, BatchNorm folding all exercised): every slot of the skip buffer."""
    from tilingnn_amd import _lib
    assert _lib.lib.tgnn_get_small_layout_limit() >= 4096
    inputs = _synthetic(n, dev)
    net, _ = make_net(dev, depth=3)
    with small_limit(0):
        p_gen, s_gen = _forward_with_slots(net, inputs, n, dev)
    p_small, s_small = _forward_with_slots(net, inputs, n, dev)
    e0 = orc.rel_max_err(s_small[0], s_gen[0].double())         # the init MLP: two BatchNorms, fx = 3 input columns
    print(f"n {n} slot 0: {e0:.2e}")
    assert e0 < 2e-5
    for k in range(1, 4):
        err = orc.rel_max_err(s_small[k], s_gen[k].double())
        print(f"n {n} slot {k}: {err:.2e}")
        # BatchNorm of near-constant collision columns amplifies the last-bit differences of the sums; 17 rows: statistics over
        # 17 values -- against the fp64 oracle the persistent kernel sits at 1.8e-4 in slot 2, the general schedule at 9.0e-5
        # (on edge groups; 1.1e-4 on type columns: scratch/slot_err_17.py), two realisations up to 2x further apart than larger layouts'
        assert err < 2e-5 * (4 ** (k - 1)) * (2 if n < 64 else 1), (k, err)
    assert not torch.equal(s_gen[1], s_small[1])                # (it IS a different path)
    assert float((p_small - p_gen).abs().max()) < 1e-3


@pytest.mark.parametrize("depth,t,tile_count,out_dim,n", [(1, 13, 2, 1, 900), (2, 17, 2, 1, 1500), (6, 3, 4, 3, 700), (30, 13, 2, 1, 400),
                                                         (3, 1, 2, 1, 333)])
def test_small_kernel_other_shapes(dev, depth, t, tile_count, out_dim, n):
    """Depths 1 .. 30 (at 34 and 13 types the final MLP's input planes plus the type-sum tiles no longer fit LDS: general schedule), 1 .. 17 edge types, tile_count 4 (five node features),
    several probability maps: the probabilities of both schedules."""
    import ctypes as C
    from tilingnn_amd import TilinGNN, _lib, ops
    from tilingnn_amd.synth import make_super_graph
    from tilingnn_amd.weights import make_state_dict
    sg = make_super_graph(n, 8 * n, 10 * n, tile_count=tile_count, n_edge_types=t, seed=depth)
    x, adj, attr, col, _ = sg.to_torch(dev)
    fe, fx = int(attr.shape[1]), int(x.shape[1])
    outs = {}
    for name, limit in (("general", 0), ("small", 4096)):
        net = TilinGNN(adj_edge_features_dim=fe, network_depth=depth, network_width=32, output_dim=out_dim, node_features_dim=fx)
        net.load_state_dict(make_state_dict(fe, depth, 32, out_dim, fx, seed=3), strict=True)
        net = net.to(dev).train()
        with small_limit(limit):
            outs[name] = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].cpu()
        graph = ops.prepare_graph(n, adj, attr, col)
        assert graph.n_types == t
        dims = net._dims()
        assert bool(_lib.lib.tgnn_get_small_layout_limit() >= 0) and C.sizeof(dims) > 0
    assert outs["small"].shape == (n, out_dim) and bool(torch.isfinite(outs["small"]).all())
    err = float((outs["small"] - outs["general"]).abs().max())
    print(f"depth {depth} types {t} maps {out_dim}: max |p_small - p_general| = {err:.2e}")
    # shallow: rounding only; deep: twenty-plus train-mode BatchNorms amplify it (the reference's own float32 run is 1e-1 off)
    assert err < (2e-5 if depth <= 2 else 2e-3 if depth <= 6 else 1e-1)
    assert not torch.equal(outs["small"], outs["general"])


def test_layouts_above_the_limit_take_the_general_schedule(dev, general_schedule):
    """4 097 nodes are not the small-layout kernel's (they are the mid-size kernel's, tests/test_mid_layout.py): with that one
    switched off too (the fixture) the result is the general schedule's whatever the small-layout limit says."""
    from tilingnn_amd import _lib
    inputs = _synthetic(4097, dev)
    net, _ = make_net(dev, depth=3)
    p_gen, s_gen = _forward_with_slots(net, inputs, 4097, dev)
    _lib.lib.tgnn_set_small_layout_limit(4096)
    p, s = _forward_with_slots(net, inputs, 4097, dev)
    assert torch.equal(p, p_gen) and torch.equal(s, s_gen)


def test_small_kernel_on_the_real_graph_against_the_reference(dev):
    g = load_labyrinth_graph()
    ref = load_npz("ref_forward_labyrinth.npz")
    inputs = graph_tensors(g, torch.float32, dev)[:4]
    gaps = {}
    for name, limit in (("general", 0), ("small", 4096)):
        net, _ = make_net(dev)
        with small_limit(limit):
            probs, _ = _forward_with_slots(net, inputs, 1254, dev, update_running=1)
        gaps[name] = float(np.abs(probs.numpy() - ref["probs_fp64"]).max())
        sd = net.state_dict()
        for k, tol in (("init_node_feature_trans.mlp.0.batch_norm", 1e-5), ("brch_1_graph_conv_layers.0.batch_norm", 1e-3)):
            np.testing.assert_allclose(sd[k + ".running_mean"].cpu().numpy(), ref[k + ".running_mean"], rtol=tol, atol=tol)
            np.testing.assert_allclose(sd[k + ".running_var"].cpu().numpy(), ref[k + ".running_var"], rtol=tol, atol=tol)
        assert all(int(v) == 1 for k, v in sd.items() if k.endswith("num_batches_tracked"))
    print("max |p - p_fp64| on the labyrinth graph:", gaps)
    # the reference's own float32 run: 1.1e-1 (chaotic end to end); measured: general 2.1e-3, persistent kernel 8.3e-3 -- two
    # rounding realisations of the same formulas: per layer both sit on the float32 floor of their slot (the absolute gates
    # above, the per-layer table in DESIGN.md section 12), and which realisation ends closer after 20 chaotic layers is chance
    # (other weight seeds order them the other way).  ONE gate for both schedules:
    assert gaps["general"] < 2e-2 and gaps["small"] < 2e-2


# Absolute gates of a skip-buffer slot against the float64 oracle, free running (nothing teacher forced), BOTH schedules:
# slot 0 = the init MLP (two BatchNorms over node features with only tile_count distinct rows: the ill-conditioned class,
# 2e-4), slot k >= 1 = k message-passing layers behind it; every layer's two train-mode BatchNorms multiply what
# came in (the collision branch divides sigmoid columns that vary by ~0.3 % of their value), measured growth <= 4x per layer.
SLOT0_TOL = 2e-5
def slot_tol(k):
    return SLOT0_TOL if k == 0 else 2e-5 * 4 ** (k - 1)


def _slot_errors(dev, inputs64, inputs, n, depth, limit, seed=0):
    net, sd = make_net(dev, depth=depth, seed=seed)
    with small_limit(limit):
        probs, slots = _forward_with_slots(net, inputs, n, dev)
    cap = {}
    with torch.no_grad():
        want = orc.tilingnn_forward(orc.cast_sd(sd, torch.float64), *inputs64, capture=cap)[0]
    errs = [orc.rel_max_err(slots[0], cap["init"])] + [orc.rel_max_err(slots[k], cap[f"mid.{k}"]) for k in range(1, depth + 1)]
    return errs, float((probs.double() - want.cpu()).abs().max())


@pytest.mark.parametrize("depth", [1, 4])
@pytest.mark.parametrize("which", ["labyrinth", "synthetic-4096"])
def test_persistent_kernel_slots_against_the_fp64_oracle_absolute(dev, depth, which):
    """The persistent kernel is the default path of every layout the solver scores: its own absolute gates against the oracle
    (depth 1 = one layer behind the init MLP, depth 4 = residual and two-deep collision buffers in play), not a comparison
    with the other schedule.  The general schedule is held to the same numbers beside it."""
    if which == "labyrinth":
        g = load_labyrinth_graph()
        n = 1254
        inputs, inputs64 = graph_tensors(g, torch.float32, dev)[:4], graph_tensors(g, torch.float64)
    else:
        from tilingnn_amd.synth import make_super_graph
        n = 4096
        sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=5)
        inputs = sg.to_torch(dev)[:4]
        inputs64 = tuple(t.double() if t.is_floating_point() else t for t in sg.to_torch("cpu"))
    for name, limit in (("persistent", 4096), ("general", 0)):
        errs, pgap = _slot_errors(dev, inputs64, inputs, n, depth, limit)
        print(f"{which} depth {depth} {name}: slots " + " ".join(f"{e:.1e}" for e in errs) + f"  max |p - p64| {pgap:.1e}")
        for k, e in enumerate(errs):
            assert e < slot_tol(k), (name, k, e, slot_tol(k))
        assert pgap < 1e-4 * 4 ** (depth - 1)


def test_both_schedules_are_equally_close_to_the_fp64_oracle(dev):
    """Depth 4 on the real graph: every slot of the skip buffer of BOTH schedules against the float64 oracle (free running, not
    teacher forced).  Neither may be systematically further away: the persistent kernel within 3x of the general schedule
    wherever that is above the float32 floor."""
    g = load_labyrinth_graph()
    inputs = graph_tensors(g, torch.float32, dev)[:4]
    errs = {}
    for name, limit in (("general", 0), ("small", 4096)):
        net, sd = make_net(dev, depth=4)
        with small_limit(limit):
            _, slots = _forward_with_slots(net, inputs, 1254, dev)
        cap = {}
        with torch.no_grad():
            orc.tilingnn_forward(orc.cast_sd(sd, torch.float64), *graph_tensors(g, torch.float64), capture=cap)
        errs[name] = [orc.rel_max_err(slots[0], cap["init"])] + [orc.rel_max_err(slots[k], cap[f"mid.{k}"]) for k in range(1, 5)]
    print({k: [f"{e:.1e}" for e in v] for k, v in errs.items()})
    for eg, es in zip(errs["general"], errs["small"]):
        assert es < 3 * max(eg, 2e-6), (eg, es)


@pytest.mark.parametrize("n", [1254, 2500, 4096])
def test_small_kernel_is_bit_reproducible(dev, n):
    """Cross-block data moves through sc1 loads / stores and a counter barrier without cache maintenance: a stale read
    would show up as run-to-run differences."""
    inputs = _synthetic(n, dev, seed=11)
    net, _ = make_net(dev)
    first = None
    for _ in range(12):
        probs, slots = _forward_with_slots(net, inputs, n, dev)
        if first is None:
            first = (probs, slots)
        else:
            assert torch.equal(first[0], probs) and torch.equal(first[1], slots)


def test_small_kernel_running_statistics_match_the_general_schedule(dev):
    inputs = _synthetic(3000, dev, seed=3)
    sds = []
    for limit in (0, 4096):
        net, _ = make_net(dev, depth=4)
        with small_limit(limit):
            _forward_with_slots(net, inputs, 3000, dev, update_running=1)
        sds.append({k: v.detach().cpu().double() for k, v in net.state_dict().items()})
    for k in sds[0]:
        if "running" in k:
            assert orc.rel_max_err(sds[1][k], sds[0][k]) < 1e-4, k
        elif k.endswith("num_batches_tracked"):
            assert int(sds[0][k]) == int(sds[1][k]) == 1, k


def test_two_streams_two_threads_do_not_deadlock(dev):
    """Two persistent kernels launched from two threads on two streams: the library orders them (one at a time per device)."""
    import threading
    inputs = _synthetic(3000, dev, seed=2)                      # 188 tiles each: two of them do not fit the chip together
    nets = [make_net(dev, depth=6, seed=k)[0] for k in range(2)]
    want = [nets[k](*inputs)[0].clone() for k in range(2)]
    torch.cuda.synchronize()
    got, errors = [None, None], []

    def work(k):
        try:
            torch.cuda.set_device(dev)
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for _ in range(20):
                    got[k] = nets[k](*inputs)[0]
            st.synchronize()
        except BaseException as exc:                             # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors and not any(t.is_alive() for t in threads)
    torch.cuda.synchronize()
    for k in range(2):
        assert torch.equal(got[k], want[k])


def test_uncached_preparation_beside_a_persistent_forward_does_not_deadlock(dev):
    """The one-launch graph preparation also synchronises its (up to 16) blocks with spin barriers: beside a persistent forward
    that fills the chip (n close to 4096: 256 tiles, one per CU) neither could get all its blocks resident -- both go through
    the per-device chain of spin-barrier kernels.  Two threads, two streams, every forward prepares its layout again."""
    import threading
    from tilingnn_amd.graph_networks import _graph_cache
    inputs = _synthetic(4090, dev, seed=4)
    nets = [make_net(dev, depth=4, seed=k)[0] for k in range(2)]
    want = [nets[k](*inputs)[0].clone() for k in range(2)]
    torch.cuda.synchronize()
    got, errors = [None, None], []

    def work(k):
        try:
            torch.cuda.set_device(dev)
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for _ in range(15):
                    got[k] = nets[k](*inputs)[0]
            st.synchronize()
        except BaseException as exc:                             # noqa: BLE001
            errors.append(exc)

    before = _graph_cache.enabled
    _graph_cache.enabled = False
    try:
        threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=180)
    finally:
        _graph_cache.enabled = before
    assert not errors and not any(t.is_alive() for t in threads)
    torch.cuda.synchronize()
    for k in range(2):
        assert torch.equal(got[k], want[k])


def test_three_layouts_side_by_side_are_their_solo_runs(dev):
    """TilinGNN.forward_many: K independent layouts (own BatchNorm populations), every one on a stream of its own, the
    persistent kernels running beside each other when they fit the device together (the reference's crop loop,
    Tiling-Shape.py:52-64, hands over such layouts one after the other).  Every result is bit-identical to the layout's solo
    forward; repeated, with the layouts in another order and with more layouts than streams."""
    from tilingnn_amd.synth import make_super_graph
    net, _ = make_net(dev, depth=20)
    g = load_labyrinth_graph()
    layouts = [tuple(graph_tensors(g, torch.float32, dev)[:4])]
    for n, seed in ((1100, 3), (900, 4), (1300, 5), (640, 6)):
        sg = make_super_graph(n, 7 * n, 8 * n, tile_count=2, n_edge_types=13, seed=seed)
        layouts.append(tuple(sg.to_torch(dev)[:4]))
    solo = [net(x=l[0], adj_e_index=l[1], adj_e_features=l[2], col_e_idx=l[3])[0].clone() for l in layouts]
    torch.cuda.synchronize()
    for order in ([0, 1, 2], [2, 0, 1], [0, 1, 2, 3, 4], [4, 3, 2, 1, 0, 0]):
        for _ in range(3):
            outs = net.forward_many([layouts[i] for i in order])
            torch.cuda.synchronize()
            for i, o in zip(order, outs):
                assert torch.equal(o, solo[i]), (order, i)
