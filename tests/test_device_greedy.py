"""The greedy assembly loop with the acceptance batched on the device (csrc/greedy.hip, tilingnn_amd.util.algorithms.
solve_by_device_greedy): the documented substitute of /root/reference/util/algorithms.py:41-54 for large layouts.  It does not
reproduce numpy's RNG stream (the host loop does, tests/test_hip_parity.py); what it must keep are the loop's invariants: a
collision-free selection, maximal at the end, every node labelled; and it must be seeded and reproducible."""
import numpy as np
import pytest
import torch

from tests.test_hip_parity import make_net

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _solver(dev, net):
    from tilingnn_amd.solver.ml_solver.ml_solver import ML_Solver
    return ML_Solver(None, dev, None, net, num_prob_maps=1)


def _layout(n, dev, seed):
    from tilingnn_amd.synth import make_super_graph
    from tilingnn_amd.util.algorithms import DeviceLayout
    sg = make_super_graph(n, 8 * n, 10 * n, tile_count=2, n_edge_types=13, seed=seed)
    x, adj, attr, col, _ = sg.to_torch(dev)
    return DeviceLayout(x, adj, attr, col), col.cpu().numpy()


def _check_selection(sel, col, n):
    sel = np.asarray(sel) > 0
    u, v = col
    assert not np.any(sel[u] & sel[v] & (u != v))               # no two selected tiles collide
    covered = sel.copy()
    covered[v[sel[u]]] = True
    assert covered.all()                                        # maximal: every node is selected or collides with a selected one
    return int(sel.sum())


@pytest.mark.parametrize("n", [3000, 20000, 100000])
def test_device_greedy_selection_is_collision_free_maximal_and_seeded(dev, n):
    from tilingnn_amd.util import algorithms as alg
    layout, col = _layout(n, dev, seed=3)
    net, _ = make_net(dev)
    ms = _solver(dev, net)
    sel_a, score, order_a = alg.solve_by_device_greedy(ms, layout, seed=7)
    rounds = alg.solve_by_device_greedy.last_rounds
    k = _check_selection(sel_a, col, n)
    assert score is None and len(order_a) == k and sorted(order_a) == list(np.flatnonzero(sel_a))
    sel_b, _, order_b = alg.solve_by_device_greedy(ms, layout, seed=7)
    assert np.array_equal(sel_a, sel_b) and order_a == order_b  # seeded: the same tiles in the same order
    sel_c, _, _ = alg.solve_by_device_greedy(ms, layout, seed=8)
    _check_selection(sel_c, col, n)
    assert not np.array_equal(sel_a, sel_c)                     # (another seed, another draw)
    print(f"n {n}: {k} tiles selected in {rounds} rounds")
    assert rounds <= 40 + 4 * int(np.log2(n))                   # O(log N) rounds, not O(sqrt N)


@pytest.mark.parametrize("n,seed", [(300, 1), (3000, 7), (20000, 5)])
def test_finishing_in_one_launch_is_the_round_by_round_solve(dev, n, seed):
    """[r6] tgnn_greedy_finish: once the sub-layout has no adjacency (or no collision) edge left, the remaining rounds as one launch
    on that sub-layout -- the same selection, order and round count as building every sub-layout and running tgnn_greedy_round."""
    from tilingnn_amd.util import algorithms as alg
    layout, col = _layout(n, dev, seed=3)
    net, _ = make_net(dev)
    ms = _solver(dev, net)
    sel_a, _, order_a = alg.solve_by_device_greedy(ms, layout, seed=seed, finish=False)
    rounds_a = alg.solve_by_device_greedy.last_rounds
    net2, _ = make_net(dev)                                     # (the same running statistics at the start: train-mode forwards)
    seen = []
    orig = alg.lib.tgnn_greedy_finish

    class Proxy:
        def __getattr__(self, k):
            if k == "tgnn_greedy_finish":
                return lambda *a: (seen.append(int(a[1])), orig(*a))[1]
            return getattr(_real, k)
    _real = alg.lib
    alg.lib = Proxy()
    try:
        sel_b, _, order_b = alg.solve_by_device_greedy(_solver(dev, net2), layout, seed=seed, finish=True)
    finally:
        alg.lib = _real
    assert seen and seen[0] >= 1, "the solve never reached a sub-layout without adjacency edges"
    assert np.array_equal(sel_a, sel_b) and order_a == order_b and alg.solve_by_device_greedy.last_rounds == rounds_a
    _check_selection(sel_b, col, n)
    print(f"n {n}: finished in one launch from {seen[0]} nodes, {rounds_a} rounds in all")


@pytest.mark.parametrize("n,drop", [(3000, "adj"), (4096, "adj"), (2000, "col")])
def test_a_layout_without_adjacency_or_collision_edges_is_one_launch(dev, n, drop):
    """The whole solve inside tgnn_greedy_finish (thousands of nodes, many rounds, real collisions) against the round-by-round path."""
    from tilingnn_amd.util import algorithms as alg
    layout, col = _layout(n, dev, seed=4)
    if drop == "adj":
        layout = alg.DeviceLayout(layout.node_feature, layout.align_edge_index[:, :0], layout.align_edge_features[:0], layout.collide_edge_index)
    else:
        layout = alg.DeviceLayout(layout.node_feature, layout.align_edge_index, layout.align_edge_features, layout.collide_edge_index[:, :0])
        col = col[:, :0]
    net, _ = make_net(dev)
    ms = _solver(dev, net)
    sel_a, _, order_a = alg.solve_by_device_greedy(ms, layout, seed=11, finish=False)
    rounds_a = alg.solve_by_device_greedy.last_rounds
    sel_b, _, order_b = alg.solve_by_device_greedy(ms, layout, seed=11, finish=True)
    assert np.array_equal(sel_a, sel_b) and order_a == order_b and alg.solve_by_device_greedy.last_rounds == rounds_a
    assert rounds_a >= (2 if drop == "adj" else 1)      # (no collision edge: p = 1 in round 1, everything is accepted at once)
    _check_selection(sel_b, col, n)
    print(f"n {n} without {drop} edges: {int(np.sum(sel_b))} tiles in {rounds_a} rounds")


def test_device_greedy_round_against_a_numpy_restatement(dev):
    """One round of tgnn_greedy_round on its own against the same rule in numpy (same running mean, same precedence, the same
    counter-based uniforms): exact agreement of the accepted set, the alive mask and the saved means."""
    import ctypes as C
    from tilingnn_amd import _lib
    from tilingnn_amd._lib import check, lib, ptr
    n = 5000
    layout, col = _layout(n, dev, seed=5)
    rng = np.random.default_rng(0)
    prob = rng.uniform(0.05, 1.0, size=n).astype(np.float32)
    prob[rng.integers(0, n, 200)] = 0.5                         # ties
    saved = rng.uniform(0.2, 1.0, size=n)
    rnd, seed = 3, 12345

    def uniform(seed, rnd, node):
        m = (1 << 64) - 1
        z = (seed + 0x9E3779B97F4A7C15 * (node + 1) + 0xD1B54A32D192ED03 * (rnd + 1)) & m
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
        z ^= z >> 31
        return (z >> 11) / 9007199254740992.0
    p = np.power(np.power(saved, rnd - 1) * prob.astype(np.float64), 1.0 / rnd)
    u, v = col
    beaten = np.zeros(n, dtype=bool)
    worse = (p[v] > p[u]) | ((p[v] == p[u]) & (v < u))
    beaten[u[worse & (u != v)]] = True
    acc = np.array([(not beaten[i]) and np.exp(p[i] - 1.0) > uniform(seed, rnd, i) for i in range(n)])
    alive_want = np.ones(n, dtype=np.int32)
    alive_want[acc] = 0
    alive_want[v[acc[u]]] = 0
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    prob_d, saved_d = t(prob, torch.float32), t(saved, torch.float64)
    alive_d = torch.ones(n, dtype=torch.int32, device=dev)
    sel_d = torch.zeros(n, dtype=torch.int32, device=dev)
    tail = torch.zeros(2, dtype=torch.int64, device=dev)
    wsb = int(lib.tgnn_greedy_round_workspace_bytes(n))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    check(lib.tgnn_greedy_round(ptr(prob_d), 1, None, n, ptr(layout.collide_edge_index), int(col.shape[1]), rnd, seed, ptr(saved_d),
                                ptr(alive_d), ptr(sel_d), ptr(tail[:1]), ptr(tail[1:].view(torch.int32)[:1]), ptr(ws), wsb,
                                _lib.current_stream(dev)))
    torch.cuda.synchronize()
    assert int(tail[0]) == int(acc.sum()) and int(tail[1]) == 0
    assert np.array_equal(sel_d.cpu().numpy() == rnd, acc)
    assert np.array_equal(alive_d.cpu().numpy(), alive_want)
    np.testing.assert_allclose(saved_d.cpu().numpy(), p, rtol=1e-14)
    assert C.sizeof(C.c_int64) == 8


def test_ml_solver_switches_to_the_device_loop_above_its_threshold(dev):
    from tilingnn_amd.util import algorithms as alg
    layout, col = _layout(6000, dev, seed=9)
    net, _ = make_net(dev, depth=4)
    ms = _solver(dev, net)
    ms.device_greedy_min_nodes = 5000
    alg.solve_by_device_greedy.last_rounds = -1
    out, score = ms.solve(layout)
    assert alg.solve_by_device_greedy.last_rounds > 0
    _check_selection(out.predict, col, 6000)
    assert out.predict_probs.shape == (6000,)


def test_device_greedy_tilings_score_like_the_host_sweeps_on_the_real_layout(dev):
    """The substitute must not buy its O(log N) rounds with worse tilings.  The real labyrinth layout (1 254 placements), the
    reference's own score formula (oracle.greedy_oracle.solution_score = losses.py:120-148; the fixture carries no polygons, so
    with unit perimeters and the layout's total tile area as the contour: a quantity both loops are measured with alike), five
    seeds each: the device loop's mean score within 3 % of the host sweep's (util/algorithms.py:18-62, the reference's random
    stream), its tile counts within 5 %, every selection collision free and maximal."""
    from oracle import greedy_oracle as go
    from tests.golden_util import load_labyrinth_graph
    from tilingnn_amd.solver.ml_solver.ml_solver import LayoutArrays, ML_Solver
    from tilingnn_amd.util import algorithms as alg
    g = load_labyrinth_graph()
    n = int(g["x"].shape[0])
    col = np.asarray(g["col"])
    contour = float(np.asarray(g["x"], dtype=np.float64)[:, -1].sum())

    def score(sel):
        return go.solution_score((np.asarray(sel) > 0).astype(np.float32), g["x"], g["adj"], g["adj_attr"], np.ones(n), 1.0, 1.0, contour)

    host, device, tiles_h, tiles_d = [], [], [], []
    for seed in range(5):
        net, _ = make_net(dev)                                  # fresh running statistics for every solve
        solver = ML_Solver(None, dev, None, net, num_prob_maps=1)
        layout = LayoutArrays(g["x"], g["adj"], g["adj_attr"], g["col"], g["col_attr"])
        np.random.seed(seed)
        out, _ = solver.solve(layout)
        tiles_h.append(_check_selection(out.predict, col, n))
        host.append(score(out.predict))
        net, _ = make_net(dev)
        solver = ML_Solver(None, dev, None, net, num_prob_maps=1)
        sel, _, _ = alg.solve_by_device_greedy(solver, LayoutArrays(g["x"], g["adj"], g["adj_attr"], g["col"], g["col_attr"]), seed=seed)
        tiles_d.append(_check_selection(sel, col, n))
        device.append(score(sel))
    mh, md = float(np.mean(host)), float(np.mean(device))
    print(f"labyrinth, 5 seeds: host sweep score {mh:.5f} ({np.mean(tiles_h):.1f} tiles), device loop {md:.5f} ({np.mean(tiles_d):.1f} tiles)")
    assert md >= 0.97 * mh, (host, device)
    assert abs(np.mean(tiles_d) - np.mean(tiles_h)) <= 0.05 * np.mean(tiles_h), (tiles_h, tiles_d)
