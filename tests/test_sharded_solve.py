"""The greedy assembly loop (/root/reference/util/algorithms.py:18-62) on a layout that STAYS SHARDED, as a product call
(tilingnn_amd.dist.solve_sharded + finish_on_one_device): shard -> score -> gather probabilities -> accept -> compact, four
thread-simulated ranks on one GPU.  Against the single-GPU loop with the same numpy seed: the same tiles in the same order.  And
the device-side shard compaction (compact_shard_device) against the numpy one it replaces, bit for bit."""
import threading

import numpy as np
import pytest
import torch

from tests.test_hip_parity import make_net

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.mark.parametrize("world", [2, 4])
def test_device_shard_compaction_equals_the_numpy_one(dev, world):
    """Three rounds of shrinking alive sets (one of them empties a rank): x, edges, attributes, halo list, receive counts and
    range bounds of every rank's compacted shard."""
    from tilingnn_amd import dist as tdist
    from tilingnn_amd.synth import make_super_graph
    n = 9000
    sg = make_super_graph(n, 8 * n, 10 * n, tile_count=2, n_edge_types=13, seed=4)
    shards = [tdist.make_shard(sg.node_feature, sg.align_edge_index, sg.align_edge_features, sg.collide_edge_index, r, world)
              for r in range(world)]
    be = tdist.HipBackend(dev)
    dshards = [(s, be.upload(s)) for s in shards]
    rng = np.random.default_rng(1)
    alive = np.ones(n, dtype=bool)
    for keep in (0.8, 0.5, 0.3):
        cur = int(alive.sum())
        alive_rel = rng.uniform(size=cur) < keep
        if keep == 0.5:
            alive_rel[: cur // world] = False                     # a rank left without a single node
        shards = [tdist.compact_shard(s, alive_rel) for s in shards]
        dshards = [tdist.compact_shard_device(s, inp, alive_rel) for s, inp in dshards]
        alive = np.ones(int(alive_rel.sum()), dtype=bool)
        for want, (got, inp) in zip(shards, dshards):
            assert (got.n_total, got.lo, got.n_own, got.recv_counts) == (want.n_total, want.lo, want.n_own, want.recv_counts)
            np.testing.assert_array_equal(got.halo_ids, want.halo_ids)
            np.testing.assert_array_equal(got.bounds, want.bounds)
            np.testing.assert_array_equal(inp["x"].cpu().numpy(), want.x.astype(np.float32))
            np.testing.assert_array_equal(inp["adj"].cpu().numpy(), want.adj)
            np.testing.assert_array_equal(inp["attr"].cpu().numpy(), want.adj_attr.astype(np.float32))
            np.testing.assert_array_equal(inp["col"].cpu().numpy(), want.col)


def test_sharded_solve_selects_the_tiles_of_the_single_gpu_loop(dev, general_schedule):
    from tilingnn_amd import dist as tdist
    from tilingnn_amd.solver.ml_solver.ml_solver import LayoutArrays, ML_Solver
    from tilingnn_amd.synth import make_super_graph
    from tilingnn_amd.util.algorithms import solve_by_probablistic_greedy
    world, n, seed = 4, 6000, 5
    sg = make_super_graph(n, 60000, 75000, tile_count=2, n_edge_types=13, seed=21)
    layout = LayoutArrays(sg.node_feature, sg.align_edge_index, sg.align_edge_features, sg.collide_edge_index,
                          np.zeros((sg.collide_edge_index.shape[1], 1), dtype=np.float32))
    net, _ = make_net(dev, depth=3)
    np.random.seed(seed)
    want_sel, _, want_order = solve_by_probablistic_greedy(ML_Solver(None, dev, None, net, num_prob_maps=1), layout,
                                                           score_fn=lambda *a, **k: 0.0)
    nets = [make_net(dev, depth=3)[0] for _ in range(world)]
    hub = tdist.ThreadSimCollectives.Hub(world)
    results, errors = [None] * world, []

    def work(r):
        try:
            torch.cuda.set_device(dev)
            shard = tdist.make_shard(sg.node_feature, sg.align_edge_index, sg.align_edge_features, sg.collide_edge_index, r, world)
            coll = tdist.ThreadSimCollectives(hub, r)
            coll.setup(shard)
            uniform = np.random.RandomState(seed).uniform         # this rank's copy of the stream np.random.seed(seed) starts
            sweep, rounds = tdist.solve_sharded(nets[r], shard, dev, coll, coll.setup, sg.collide_edge_index, uniform=uniform)
            sharded_rounds = rounds
            left = int(sweep.unlabelled.sum())
            tdist.finish_on_one_device(ML_Solver(None, dev, None, nets[r], num_prob_maps=1), layout, sweep)
            results[r] = (sweep.selection.copy(), list(sweep.order), sharded_rounds, left)
        except BaseException as exc:                               # noqa: BLE001
            errors.append(exc)
            hub.barrier.abort()
    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    print(f"sharded rounds {results[0][2]}, nodes left for the single-device tail {results[0][3]}, tiles {int(want_sel.sum())}")
    assert results[0][2] >= 5
    for sel, order, _, _ in results:                               # every rank took the same decisions ...
        assert np.array_equal(sel, results[0][0]) and order == results[0][1]
    assert np.array_equal(results[0][0], want_sel) and results[0][1] == want_order   # ... those of the single-GPU loop
