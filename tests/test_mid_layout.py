"""Mid-size layouts (4 097 .. 65 536 nodes: BASELINE config 2, greedy rounds of large solves, a rank's share of a strong-scaled
100 000-node layout): the 20 layers of /root/reference/graph_networks/networks/TilinGNN.py:59-71 as ONE persistent kernel
(csrc/forward_mid.hip) against the fp64 oracle -- absolute, slot by slot -- and against the general launch schedule."""
import contextlib
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import tilingnn_oracle as orc
from tests.test_hip_parity import make_net
from tests.test_small_layout import _forward_with_slots, slot_tol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@contextlib.contextmanager
def mid_limit(n):
    from tilingnn_amd import _lib
    before = _lib.lib.tgnn_get_mid_layout_limit()
    _lib.lib.tgnn_set_mid_layout_limit(n)
    try:
        yield
    finally:
        _lib.lib.tgnn_set_mid_layout_limit(before)


def _spin_ok(dev):
    from tilingnn_amd import _lib
    code = C.c_uint32(0)
    _lib.check(_lib.lib.tgnn_spin_error_poll(_lib.current_stream(dev), C.byref(code)))
    return code.value


def _layout(n, dev, seed=5, tile_count=2, types=13, ea_per=10, ec_per=12.5):
    from tilingnn_amd.synth import make_super_graph
    sg = make_super_graph(n, int(ea_per * n), int(ec_per * n), tile_count=tile_count, n_edge_types=types, seed=seed)
    inputs = sg.to_torch(dev)[:4]
    inputs64 = tuple(t.double() if t.is_floating_point() else t for t in sg.to_torch("cpu"))
    return inputs, inputs64


def test_the_graph_carries_the_batches_and_they_cover_every_edge(dev):
    """tgnn_mid_entries_build against the column structure it is cut from: per tile and type the multiset of (destination row,
    source) pairs, the order of a row's edges, the store / add flags, no row twice in one gather instruction."""
    from tilingnn_amd import ops
    n = 5000
    (x, adj, attr, col), _ = _layout(n, dev, seed=2)
    g = ops.prepare_graph(n, adj, attr, col)
    assert g.mid is not None and g.cols is not None
    nb = g.mid.tile_nb.cpu().numpy()
    ent = g.mid.ent.cpu().numpy().view(np.uint32).reshape(-1, 24, 36)
    ptr_ = g.cols.tile_col_ptr.cpu().numpy()
    meta = g.cols.col_meta.cpu().numpy()
    src = g.cols.col_src.cpu().numpy().reshape(-1, 16)
    assert nb.max() <= 24 and nb.min() >= 1
    for tile in list(range(0, 40)) + [len(nb) - 1, len(nb) // 2]:
        want = {}                                              # type -> per row the sources in column (= CSR) order
        for c in range(ptr_[tile], ptr_[tile + 1] - 1):
            t = int(meta[c] & 0xff)
            for r in range(16):
                if src[c, r] >= 0:
                    want.setdefault(t, [[] for _ in range(16)])[r].append(int(src[c, r]))
        got, types_seen, masks = {}, [], {}
        for b in range(nb[tile]):
            hdr = ent[tile, b, :4]
            t = int(hdr[0] & 0xff)
            rows = got.setdefault(t, [[] for _ in range(16)])
            for gi in range(4):
                seen = set()
                for o in range(8):
                    w = int(ent[tile, b, 4 + 4 * o + gi])
                    if w == 0x21000000:                        # an empty slot: no source, the kernel's spare row
                        continue
                    r = (w >> 25) & 31
                    assert r < 16 and r not in seen            # never two entries of a row in one instruction
                    seen.add(r)
                    assert bool((w >> 30) & 1) == (len(rows[r]) > 0)   # a row's first edge of a type stores, further ones add
                    rows[r].append(w & 0xffffff)
            if hdr[0] & 0x100:
                types_seen.append(t)
                masks[t] = int(hdr[1])
        assert types_seen == sorted(want)                       # every type once, in order, closed by a `last` batch
        for t in want:
            assert got[t] == want[t]
            assert masks[t] == sum(1 << r for r in range(16) if want[t][r])


# [r6] (50000, 4) -> (40000, 2), (8000, 4) -> (8000, 3), (20000, 4) -> (20000, 3) and (65536, 2) -> (65536, 1): the fp64 oracle on the host was 107 + 69 s of a 626 s suite (profiles/r06_gpu_test_durations.txt);
# depth 3 (the slot tolerance 2e-5 * 4^(k-1) up to k = 3) stays covered at 8 000 and 20 000 nodes, the largest sizes keep their first slots
@pytest.mark.parametrize("n,depth", [(8000, 1), (8000, 3), (20000, 3), (40000, 2), (4097, 2), (65536, 1)])
def test_mid_kernel_slots_against_the_fp64_oracle_absolute(dev, n, depth):
    """Every slot of the skip buffer, free running, against the float64 oracle with the per-class tolerances of
    tests/test_small_layout.py (slot k: 2e-5 * 4^(k-1)); the general schedule is held to the same numbers beside it."""
    inputs, inputs64 = _layout(n, dev)
    from tilingnn_amd import _lib, ops
    with mid_limit(65536):
        assert ops.prepare_graph(n, *inputs[1:]).mid is not None
    for name, limit in (("persistent", 65536), ("general", 0)):
        net, sd = make_net(dev, depth=depth)
        with mid_limit(limit):
            c0 = _lib.forward_path_counts()
            probs, slots = _forward_with_slots(net, inputs, n, dev)
            assert _lib.forward_path_counts()[2 if limit else 0] == c0[2 if limit else 0] + 1
        assert _spin_ok(dev) == 0
        cap = {}
        with torch.no_grad():
            want = orc.tilingnn_forward(orc.cast_sd(sd, torch.float64), *inputs64, capture=cap)[0]
        errs = [orc.rel_max_err(slots[0], cap["init"])] + [orc.rel_max_err(slots[k], cap[f"mid.{k}"]) for k in range(1, depth + 1)]
        pgap = float((probs.double() - want.cpu()).abs().max())
        print(f"n {n} depth {depth} {name}: slots " + " ".join(f"{e:.1e}" for e in errs) + f"  max |p - p64| {pgap:.1e}")
        for k, e in enumerate(errs):
            assert e < slot_tol(k), (name, k, e, slot_tol(k))
        assert pgap < 1e-4 * 4 ** (depth - 1)
    assert True


@pytest.mark.parametrize("n", [4100, 10000, 30000])
def test_mid_kernel_against_the_general_schedule_layer_by_layer(dev, n):
    """Depth 3: same formulas, other association of the BatchNorm and same-type sums -- rounding only; and it IS another path."""
    inputs, _ = _layout(n, dev, seed=7)
    net, _ = make_net(dev, depth=3)
    from tilingnn_amd import _lib
    with mid_limit(0):
        c0 = _lib.forward_path_counts()
        p_gen, s_gen = _forward_with_slots(net, inputs, n, dev)
        c1 = _lib.forward_path_counts()
    with mid_limit(65536):
        p_mid, s_mid = _forward_with_slots(net, inputs, n, dev)
    c2 = _lib.forward_path_counts()
    assert (c1[0] - c0[0], c1[2] - c0[2]) == (1, 0) and (c2[0] - c1[0], c2[2] - c1[2]) == (0, 1)   # it IS another path
    assert _spin_ok(dev) == 0
    e0 = orc.rel_max_err(s_mid[0], s_gen[0].double())
    print(f"n {n} slot 0: {e0:.2e}")
    assert e0 < slot_tol(0)                                    # (the init MLP may run in the kernel's prologue: rounding only)
    for k in range(1, 4):
        err = orc.rel_max_err(s_mid[k], s_gen[k].double())
        print(f"n {n} slot {k}: {err:.2e}")
        assert err < 2e-5 * (4 ** (k - 1)), (k, err)
    assert float((p_mid - p_gen).abs().max()) < 1e-3


@pytest.mark.parametrize("depth,t,tile_count,out_dim,n,ea_per,ec_per", [(1, 13, 2, 1, 6000, 10, 12.5), (2, 15, 2, 1, 9000, 8, 10),
                                                                       (6, 3, 4, 3, 7000, 8, 10), (20, 13, 2, 1, 10000, 8, 10),
                                                                       (3, 1, 2, 1, 5000, 6, 4), (3, 13, 1, 1, 12000, 10, 12.5)])
def test_mid_kernel_other_shapes(dev, depth, t, tile_count, out_dim, n, ea_per, ec_per):
    """1 .. 15 edge types (the NNConv image of 17 type blocks no longer fits LDS beside the rest: general schedule), depth 1 .. 20 (BASELINE config 2 itself: 10 000 nodes / 80 000 + 100 000 edges), tile_count 1 / 2 / 4,
    several probability maps: the probabilities of both schedules."""
    from tilingnn_amd import TilinGNN
    from tilingnn_amd.synth import make_super_graph
    from tilingnn_amd.weights import make_state_dict
    sg = make_super_graph(n, int(ea_per * n), int(ec_per * n), tile_count=tile_count, n_edge_types=t, seed=depth)
    x, adj, attr, col, _ = sg.to_torch(dev)
    fe, fx = int(attr.shape[1]), int(x.shape[1])
    from tilingnn_amd import _lib
    outs, took = {}, {}
    for name, limit in (("general", 0), ("mid", 65536)):
        net = TilinGNN(adj_edge_features_dim=fe, network_depth=depth, network_width=32, output_dim=out_dim, node_features_dim=fx)
        net.load_state_dict(make_state_dict(fe, depth, 32, out_dim, fx, seed=3), strict=True)
        net = net.to(dev).train()
        with mid_limit(limit):
            c0 = _lib.forward_path_counts()
            outs[name] = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].cpu()
            c1 = _lib.forward_path_counts()
        took[name] = tuple(b - a for a, b in zip(c0, c1))
    assert took == {"general": (1, 0, 0), "mid": (0, 0, 1)}, took
    assert _spin_ok(dev) == 0
    assert outs["mid"].shape == (n, out_dim) and bool(torch.isfinite(outs["mid"]).all())
    err = float((outs["mid"] - outs["general"]).abs().max())
    print(f"depth {depth} types {t} maps {out_dim}: max |p_mid - p_general| = {err:.2e}")
    # the two schedules sum in different orders (~1e-7 per layer) and every BatchNorm of the collision branch multiplies what is
    # there (the per-slot gate above: 2e-5 . 4^(k-1)); measured 1.2e-5 / 3.0e-5 at depth 1 / 2, 3e-5 .. 7e-4 at 3 .. 6, 2.9e-4 at 20
    assert err < (2e-5 * 4 ** (depth - 1) if depth <= 2 else 2e-3 if depth <= 6 else 1e-1)


def test_layouts_above_the_mid_limit_take_the_general_schedule(dev):
    inputs, _ = _layout(6000, dev)
    net, _ = make_net(dev, depth=3)
    with mid_limit(0):
        p_gen, s_gen = _forward_with_slots(net, inputs, 6000, dev)
    with mid_limit(5999):
        p, s = _forward_with_slots(net, inputs, 6000, dev)
    assert torch.equal(p, p_gen) and torch.equal(s, s_gen)


@pytest.mark.parametrize("n", [8000, 20000, 50000])
def test_mid_kernel_is_bit_reproducible(dev, n):
    """Cross-block data moves through sc1 loads / stores, tagged partial rows and a counter barrier without cache maintenance: a
    stale read would show up as run-to-run differences.  Twelve runs, depth 20."""
    from tilingnn_amd import _lib
    inputs, _ = _layout(n, dev, seed=11)
    net, _ = make_net(dev)
    first = None
    c0 = _lib.forward_path_counts()
    with mid_limit(65536):                                      # (above the default limit the kernel takes its tiles in rounds)
        for _ in range(12):
            probs, slots = _forward_with_slots(net, inputs, n, dev)
            if first is None:
                first = (probs, slots)
            else:
                assert torch.equal(first[0], probs) and torch.equal(first[1], slots)
    assert _lib.forward_path_counts()[2] - c0[2] == 12
    assert _spin_ok(dev) == 0


def test_mid_kernel_running_statistics_match_the_general_schedule(dev):
    inputs, _ = _layout(9000, dev, seed=3)
    sds = []
    for limit in (0, 65536):
        net, _ = make_net(dev, depth=4)
        with mid_limit(limit):
            _forward_with_slots(net, inputs, 9000, dev, update_running=1)
        sds.append({k: v.detach().cpu().double() for k, v in net.state_dict().items()})
    for k in sds[0]:
        if "running" in k:
            assert orc.rel_max_err(sds[1][k], sds[0][k]) < 1e-4, k
        elif k.endswith("num_batches_tracked"):
            assert int(sds[0][k]) == int(sds[1][k]) == 1, k


@contextlib.contextmanager
def mid_tail(on):
    from tilingnn_amd import _lib
    before = _lib.lib.tgnn_set_mid_tail(int(on))
    try:
        yield
    finally:
        _lib.lib.tgnn_set_mid_tail(before)


def _forward_slots_maps(net, inputs, n, dev, out_dim, update_running=0):
    """_forward_with_slots for any number of probability maps"""
    from tilingnn_amd import _lib, ops
    x, adj, adj_attr, col = inputs
    graph = ops.prepare_graph(n, adj, adj_attr, col)
    dims = net._dims()
    table, _ = net._param_table()
    ws_bytes = _lib.lib.tgnn_forward_workspace_bytes(C.byref(dims), n, graph.n_types)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
    probs = torch.empty(n, out_dim, device=dev)
    g = graph.c_struct()
    _lib.check(_lib.lib.tgnn_forward(C.byref(dims), table, ops.ptr(x), ops.ptr(adj_attr), C.byref(g), update_running, 0,
                                    ops.ptr(probs), ops.ptr(ws), ws_bytes, _lib.current_stream(dev), _lib.side_stream(dev)))
    torch.cuda.synchronize()
    d = net.network_depth
    return probs.cpu(), ws[: (d + 1) * n * 32 * 4].view(torch.float32).view(d + 1, n, 32).clone().cpu()


@pytest.mark.parametrize("n,depth,out_dim", [(4100, 20, 1), (10000, 20, 1), (16384, 20, 1), (7000, 6, 3), (9000, 8, 1), (12000, 23, 2)])
def test_final_mlp_as_one_persistent_kernel_against_the_launch_per_layer_one_and_the_oracle(dev, n, depth, out_dim):
    """csrc/forward_tail.hip (TilinGNN.py:74-76 behind the persistent layer loop): same skip buffer in, probabilities and the four
    BatchNorms' running statistics against the general final MLP (fp16 pairs / bf16 x 3 there too: rounding only) and against the
    float64 oracle's final MLP on the kernel's own skip buffer; K = 32 (depth + 1) below, at and above a multiple of the 8-step
    chunk; 2 / 3 / 4 tiles per block; several probability maps; twelve runs bit-identical."""
    from tilingnn_amd import TilinGNN, _lib
    from tilingnn_amd.synth import make_super_graph
    from tilingnn_amd.weights import make_state_dict
    sg = make_super_graph(n, 10 * n, 12 * n, tile_count=2, n_edge_types=13, seed=depth)
    inputs = sg.to_torch(dev)[:4]
    sd = make_state_dict(15, depth, 32, out_dim, 3, seed=2)
    res = {}
    for tail in (0, 1):
        net = TilinGNN(adj_edge_features_dim=15, network_depth=depth, network_width=32, output_dim=out_dim, node_features_dim=3)
        net.load_state_dict(sd, strict=True)
        net = net.to(dev).train()
        with mid_limit(65536), mid_tail(tail):
            c0 = _lib.forward_path_counts()
            probs, slots = _forward_slots_maps(net, inputs, n, dev, out_dim, update_running=1)
            assert _lib.forward_path_counts()[2] == c0[2] + 1
            stats = {k: v.detach().cpu().double() for k, v in net.state_dict().items() if k.startswith("final_mlp")}
            if tail:
                for _ in range(11):
                    again, _ = _forward_slots_maps(net, inputs, n, dev, out_dim)
                    assert torch.equal(again, probs)
        assert _spin_ok(dev) == 0
        res[tail] = (probs.double(), slots, stats)
    assert torch.equal(res[0][1], res[1][1])                     # the same skip buffer went into both
    gap = float((res[0][0] - res[1][0]).abs().max())
    with torch.no_grad():                                        # the oracle's final MLP on the kernel's own skip buffer (float64)
        cat = torch.cat([res[1][1][k].double() for k in range(depth + 1)], dim=1)
        want = orc.final_mlp(cat, orc.cast_sd(sd, torch.float64))
    ogap = float((res[1][0] - want).abs().max())
    print(f"n {n} depth {depth} maps {out_dim}: max |p_tail - p_general| {gap:.1e}, max |p_tail - p64(final MLP)| {ogap:.1e}")
    assert ogap < 5e-6 and gap < 1e-5, (ogap, gap)
    for k, v in res[0][2].items():
        if "running" in k:
            assert orc.rel_max_err(res[1][2][k], v) < 1e-5, k
        elif k.endswith("num_batches_tracked"):
            assert int(v) == int(res[1][2][k]) == 1, k


@pytest.mark.parametrize("n,tile_count", [(4100, 1), (10000, 2), (20000, 4), (50000, 2)])
def test_init_mlp_in_the_layer_loops_prologue_against_the_launches_and_the_oracle(dev, n, tile_count):
    """TilinGNN.py:54 inside forward_layers_mid_kernel (2, 4, 8 and 16 tiles per block: every way the waves share them; node features
    of 2, 3 and 5 columns): slot 0 against the five launches it replaces and against the float64 oracle, the two BatchNorms' running
    statistics, and what the rest of the network makes of it."""
    from tilingnn_amd import TilinGNN, _lib
    from tilingnn_amd.synth import make_super_graph
    from tilingnn_amd.weights import make_state_dict
    depth = 3
    sg = make_super_graph(n, 10 * n, 12 * n, tile_count=tile_count, n_edge_types=13, seed=4)
    inputs = sg.to_torch(dev)[:4]
    fx = int(inputs[0].shape[1])
    sd = make_state_dict(15, depth, 32, 1, fx, seed=6)
    res = {}
    for mode in (1, 3):
        net = TilinGNN(adj_edge_features_dim=15, network_depth=depth, network_width=32, output_dim=1, node_features_dim=fx)
        net.load_state_dict(sd, strict=True)
        net = net.to(dev).train()
        with mid_limit(65536), mid_tail(mode):
            c0 = _lib.forward_path_counts()
            probs, slots = _forward_slots_maps(net, inputs, n, dev, 1, update_running=1)
            assert _lib.forward_path_counts()[2] == c0[2] + 1
        assert _spin_ok(dev) == 0
        res[mode] = (probs.double(), slots, {k: v.detach().cpu().double() for k, v in net.state_dict().items() if k.startswith("init_node")})
    with torch.no_grad():
        want0 = orc.init_node_feature_trans(inputs[0].detach().cpu().double(), orc.cast_sd(sd, torch.float64))
    e_launch = orc.rel_max_err(res[3][1][0], res[1][1][0].double())
    e_oracle = orc.rel_max_err(res[3][1][0], want0)
    print(f"n {n} fx {fx}: slot 0 against the launches {e_launch:.1e}, against the oracle {e_oracle:.1e}; "
          f"max |p - p_launches| {float((res[3][0] - res[1][0]).abs().max()):.1e}")
    assert e_launch < slot_tol(0) and e_oracle < slot_tol(0), (e_launch, e_oracle)
    assert float((res[3][0] - res[1][0]).abs().max()) < 1e-4
    for k, v in res[1][2].items():
        if "running" in k:
            assert orc.rel_max_err(res[3][2][k], v) < 1e-5, k
        elif k.endswith("num_batches_tracked"):
            assert int(v) == int(res[3][2][k]) == 1, k


def test_final_mlp_kernel_is_left_out_above_four_tiles_per_block(dev):
    """From 16 385 nodes on (5 tiles per CU) the launch-per-layer final MLP runs behind the layer loop: bit-identical with the switch off."""
    n = 20000
    inputs, _ = _layout(n, dev, seed=2)
    net, _ = make_net(dev, depth=3)
    with mid_limit(65536):
        with mid_tail(1):
            p1, _ = _forward_with_slots(net, inputs, n, dev)
        with mid_tail(0):
            p0, _ = _forward_with_slots(net, inputs, n, dev)
    assert torch.equal(p0, p1)


@pytest.mark.parametrize("n", [2000, 8000])
def test_a_starved_persistent_kernel_gives_up_and_the_module_falls_back(dev, n, debug_hooks):
    """Bounded spins (csrc/forward_persist.h).  A persistent kernel one of whose blocks never shows up -- what a kernel looks like
    to its other blocks when another process holds compute units; simulated by the debug library's test hook -- must TERMINATE
    after the spin budget, the device's error word must say why, and
      * TilinGNN.forward_checked (the call ML_Solver.predict makes) must come back with the general schedule's result, the
        persistent schedules off for a WINDOW of forwards (tgnn_persist_fallback) and back afterwards [r5];
      * a failure nobody polled for must be loud at the NEXT forward of any kind (TGNN_ERR_STALE_RESULT) instead of poisoning
        every later persistent kernel silently [r5: ADVICE r4].
    Both persistent kernels: 2 000 nodes (forward_small.hip), 8 000 (forward_mid.hip)."""
    if debug_hooks:
        return                                                  # (ran against libtgnn_debug.so in a subprocess)
    import time
    import warnings
    from tilingnn_amd import _lib
    inputs, _ = _layout(n, dev, seed=9)
    net, _ = make_net(dev, depth=6)
    path = 1 if n <= 4096 else 2                               # index into tgnn_forward_path_counts: small / mid
    _lib.lib.tgnn_persist_fallback(1 << 40)                    # the general schedule's result
    want = net(*inputs)[0].clone()
    _lib.lib.tgnn_persist_fallback(0)
    torch.cuda.synchronize()
    before = _lib.lib.tgnn_set_spin_budget_us(20000)
    try:
        _lib.lib.tgnn_debug_spin_fault(1)
        t0 = time.perf_counter()
        net(*inputs)
        torch.cuda.synchronize()                                # (terminates: every wait is bounded)
        took = time.perf_counter() - t0
        assert took < 5.0, took
        code = _spin_ok(dev)
        assert code != 0
        assert b"gave up" in _lib.lib.tgnn_last_error()
        assert _spin_ok(dev) == 0                               # (the poll cleared the word)
        # ... and opened the fallback window: the next forwards take the general schedule, then the persistent one is back
        c0 = _lib.forward_path_counts()
        net(*inputs)
        c1 = _lib.forward_path_counts()
        assert c1[0] - c0[0] == 1 and c1[path] == c0[path]
        _lib.lib.tgnn_persist_fallback(0)
        net(*inputs)
        c2 = _lib.forward_path_counts()
        assert c2[path] - c1[path] == 1
        assert _spin_ok(dev) == 0
        _lib.lib.tgnn_persist_fallback(0)
        # forward_checked: warning, the general schedule's result, window open
        _lib.lib.tgnn_debug_spin_fault(1)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            got = net.forward_checked(*inputs)[0]
        assert any("gave up" in str(w.message) for w in caught)
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        c3 = _lib.forward_path_counts()
        net(*inputs)
        assert _lib.forward_path_counts()[path] == c3[path]     # still inside the window
        _lib.lib.tgnn_persist_fallback(0)
        # a failure NOBODY polls for: plain forward(); the next forward of any kind raises, the one after it is sound again
        _lib.lib.tgnn_debug_spin_fault(1)
        net(*inputs)
        torch.cuda.synchronize()
        with pytest.raises(_lib.TgnnError, match="EARLIER call gave up"):
            net(*inputs)
        again = net(*inputs)[0]
        torch.cuda.synchronize()
        assert torch.equal(again, want) and _spin_ok(dev) == 0   # (inside the window the stale failure opened: the general schedule)
    finally:
        _lib.lib.tgnn_debug_spin_fault(0)
        _lib.lib.tgnn_set_spin_budget_us(before)
        _lib.lib.tgnn_persist_fallback(0)
    assert _spin_ok(dev) == 0
    assert torch.equal(net.forward_checked(*inputs)[0].cpu(), net(*inputs)[0].cpu())   # healthy: the persistent schedule's own result


def test_batches_that_do_not_fit_are_seen_behind_the_launches(dev):
    """[r5] prepare_graph returns before the NNConv structure of a mid-size layout is built (tgnn_graph_prep copies the words the
    host reads out early); the batches' verdict -- result[9]: a tile with more batches than the persistent layer loop takes --
    comes with the LAST launch and is looked at behind the forward's launches (PreparedGraph.late_words_failed): a layout with
    such a tile (one node with 600 in-edges) must still come back with the general schedule's probabilities."""
    from tilingnn_amd import _lib, ops
    from tilingnn_amd.synth import make_super_graph
    n = 6000
    sg = make_super_graph(n, 8 * n, 10 * n, tile_count=2, n_edge_types=13, seed=11)
    x, adj, attr, col, _ = sg.to_torch(dev)
    adj = adj.clone()
    adj[1, :600] = 77                                             # a hub: its tile needs more than 24 batches
    net, _ = make_net(dev, depth=3)
    net.cache_graph = False
    g = ops.prepare_graph(n, adj, attr, col)
    assert "_late_words" in g.__dict__ and g.mid is not None      # (optimistic until the words are looked at)
    assert g.late_words_failed() and g.mid is None and not g.late_words_failed()
    # [r6] the forward is queued behind the preparation without waiting for the verdict, with its DEVICE address in the graph
    # struct (tgnn_graph.nn_mid_verdict): the persistent kernels read it and leave without output or running-statistics update,
    # the host looks at the word behind its launches and repeats on the general schedule -- ONE update, from valid statistics
    before = _lib.forward_path_counts()
    probs_upd = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].clone()
    after = _lib.forward_path_counts()
    assert after[2] - before[2] == 1 and after[0] - before[0] == 1
    fresh, _ = make_net(dev, depth=3)
    mid_limit = _lib.lib.tgnn_get_mid_layout_limit()
    _lib.lib.tgnn_set_mid_layout_limit(0)
    try:
        fresh(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
    finally:
        _lib.lib.tgnn_set_mid_layout_limit(mid_limit)
    for k, v in fresh.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(net.state_dict()[k]) == 1 == int(v), k
        elif k.endswith(("running_mean", "running_var")):
            assert torch.allclose(net.state_dict()[k], v, rtol=1e-5, atol=1e-7), k
    # without the update (forward_many's mode): the same
    before = _lib.forward_path_counts()
    probs = net._forward_one(x, adj, attr, col, update_running=False)[0].clone()
    after = _lib.forward_path_counts()
    assert float((probs - probs_upd).abs().max()) < 1e-6
    with_limits = (_lib.lib.tgnn_get_small_layout_limit(), _lib.lib.tgnn_get_mid_layout_limit())
    _lib.lib.tgnn_set_mid_layout_limit(0)
    try:
        want = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0]
    finally:
        _lib.lib.tgnn_set_mid_layout_limit(with_limits[1])
    # the optimistic launch took the mid-size kernel once, the repeat the general schedule; what comes back is the repeat's result
    assert after[2] - before[2] == 1 and after[0] - before[0] == 1
    assert float((probs - want).abs().max()) < 1e-4
