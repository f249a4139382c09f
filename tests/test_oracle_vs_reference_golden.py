"""Pins oracle/tilingnn_oracle.py against outputs of the REFERENCE ITSELF (tests/golden/*.npz,
produced by tests/golden/generate_golden.py from /root/reference's unchanged graph_networks/*).
CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import tilingnn_oracle as orc
from tests.golden_util import GOLDEN, graph_tensors, load_labyrinth_graph, load_npz
from tilingnn_amd.weights import make_state_dict, state_dict_spec


def test_state_dict_layout_matches_reference():
    ref = json.load(open(os.path.join(GOLDEN, "ref_state_dict_keys.json")))
    spec = state_dict_spec(15, 20, 32, 1, 3)
    assert list(ref) == list(spec)
    assert {k: list(v) for k, v in spec.items()} == ref
    assert len(spec) == 664


@pytest.fixture(scope="module")
def laby_fp64():
    torch.set_num_threads(1)
    g = load_labyrinth_graph()
    sd = orc.cast_sd(make_state_dict(15, 20, 32, 1, 3, seed=0), torch.float64)
    cap = {}
    with torch.no_grad():
        probs, passthrough = orc.tilingnn_forward(sd, *graph_tensors(g, torch.float64), update_running=True, capture=cap)
    return g, sd, cap, probs, passthrough


def test_forward_fp64_matches_reference_end_to_end(laby_fp64):
    g, sd, cap, probs, passthrough = laby_fp64
    ref = load_npz("ref_forward_labyrinth.npz")
    assert probs.shape == (1254, 1)
    # same op order in fp64 -> agreement far below the network's fp32 chaos (1e-1)
    assert np.abs(probs.numpy() - ref["probs_fp64"]).max() < 1e-9
    assert passthrough.shape == (8502, 15)


def test_every_intermediate_matches_reference(laby_fp64):
    g, sd, cap, probs, _ = laby_fp64
    ref = load_npz("ref_forward_labyrinth.npz")
    rows = ref["sample_rows"]
    np.testing.assert_allclose(cap["init"].numpy()[rows], ref["init.rows"], rtol=0, atol=1e-9)
    for i in range(20):
        for name in ("nnconv", "gconv", "gin", "cconv"):
            t = cap[f"{name}.{i}"].numpy()
            scale = max(1.0, float(np.abs(ref[f"{name}.{i}.rows"]).max()))
            assert np.abs(t[rows] - ref[f"{name}.{i}.rows"]).max() < 1e-9 * scale, (name, i)
            assert np.abs(t.mean(0) - ref[f"{name}.{i}.colmean"]).max() < 1e-9 * scale, (name, i)
            assert np.abs(np.sqrt((t ** 2).mean(0)) - ref[f"{name}.{i}.colrms"]).max() < 1e-9 * scale, (name, i)


def test_running_stats_after_one_forward(laby_fp64):
    g, sd, cap, probs, _ = laby_fp64
    ref = load_npz("ref_forward_labyrinth.npz")
    for k in ("init_node_feature_trans.mlp.0.batch_norm", "brch_1_graph_conv_layers.0.batch_norm",
              "brch_2_coll_conv_layers.19.batch_norm", "final_mlp.0.mlp.3.batch_norm"):
        np.testing.assert_allclose(sd[k + ".running_mean"].numpy(), ref[k + ".running_mean"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(sd[k + ".running_var"].numpy(), ref[k + ".running_var"], rtol=1e-9, atol=1e-12)
        assert int(sd[k + ".num_batches_tracked"]) == int(ref[k + ".num_batches_tracked"]) == 1


def test_per_op_teacher_forced_small_graph():
    """Per-op pairs from the reference on the 256-node induced sub-graph: the oracle in fp64 on the
    same fp32-rounded inputs must land on the stored (fp32-rounded) reference outputs."""
    torch.set_num_threads(1)
    z = load_npz("ref_ops_small.npz")
    sd = orc.cast_sd(make_state_dict(15, 20, 32, 1, 3, seed=0), torch.float64)
    adj = torch.from_numpy(z["adj"].astype(np.int64)); col = torch.from_numpy(z["col"].astype(np.int64))
    adj_attr = torch.from_numpy(z["adj_attr"]).double()
    tol = 2e-7                                             # fp32 rounding of the stored expectation
    with torch.no_grad():
        out = orc.init_node_feature_trans(torch.from_numpy(z["x"]).double(), sd)
        assert orc.rel_max_err(out, torch.from_numpy(z["init.out"])) < tol
        for i in (0, 2, 19):
            h1 = torch.from_numpy(z[f"h1_in.{i}"]).double(); h2 = torch.from_numpy(z[f"h2_in.{i}"]).double()
            p1, p2 = f"brch_1_graph_conv_layers.{i}", f"brch_2_coll_conv_layers.{i}"
            assert orc.rel_max_err(orc.nnconv_mean(h1, adj, adj_attr, sd, p1), torch.from_numpy(z[f"nnconv.{i}.out"])) < tol
            assert orc.rel_max_err(orc.graph_conv(h1, adj, adj_attr, sd, p1), torch.from_numpy(z[f"gconv.{i}.out"])) < tol
            assert orc.rel_max_err(orc.gin_conv(h2, col, sd, p2), torch.from_numpy(z[f"gin.{i}.out"])) < tol
            assert orc.rel_max_err(orc.coll_conv(h2, col, sd, p2), torch.from_numpy(z[f"cconv.{i}.out"])) < tol
        fin = orc.final_mlp(torch.from_numpy(z["final.in"]).double(), sd)
        assert orc.rel_max_err(fin, torch.from_numpy(z["final.out"])) < tol


def test_tiny_graph_zero_indegree_and_self_loops():
    z = load_npz("tiny_graph.npz")
    sd = orc.cast_sd(make_state_dict(6, 3, 32, 1, 3, seed=3), torch.float64)
    x = torch.from_numpy(z["x"]); adj = torch.from_numpy(z["adj"]); col = torch.from_numpy(z["col"])
    adj_attr = torch.from_numpy(z["adj_attr"])
    cap = {}
    with torch.no_grad():
        probs, _ = orc.tilingnn_forward(sd, x, adj, adj_attr, col, capture=cap)
    assert np.abs(probs.numpy() - z["probs_fp64"]).max() < 1e-10
    assert np.abs(cap["nnconv.0"].numpy() - z["nnconv0"]).max() < 1e-10
    assert np.abs(cap["gin.0"].numpy() - z["gin0"]).max() < 1e-10
    assert np.abs(cap["gconv.2"].numpy() - z["gconv2"]).max() < 1e-9
    assert np.abs(cap["cconv.2"].numpy() - z["cconv2"]).max() < 1e-9
    # node 5 has no in-edges: its NNConv output is root term + bias only
    h = cap["h1_in.0"]
    only_root = h[5] @ sd["brch_1_graph_conv_layers.0.nnConv.root"] + sd["brch_1_graph_conv_layers.0.nnConv.bias"]
    assert torch.allclose(cap["nnconv.0"][5], only_root, atol=1e-12)


def test_unsupervised_loss_is_at_least_one():
    g = load_labyrinth_graph()
    x, adj, adj_attr, col, _ = graph_tensors(g, torch.float64)
    probs = torch.rand(x.shape[0], 3, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    losses = orc.unsupervised_losses(probs, x, col, adj, adj_attr)
    assert losses.shape == (3,) and bool((losses >= 1.0).all())          # losses.py:108 assert


# ------------------------------------------------------------------------------------------ loss on the predict path
def _loss_case_inputs(name):
    """Inputs of a case of tests/golden/ref_losses.npz (generate_loss_golden.py), float64 numpy."""
    ref = load_npz("ref_losses.npz")
    probs = ref[f"{name}.probs"]
    if name.startswith("laby"):
        g = load_labyrinth_graph()
        x, col, adj, adj_attr = g["x"], g["col"], g["adj"], g["adj_attr"]
    else:
        t = load_npz("tiny_graph.npz")
        x, col, adj, adj_attr = t["x"], t["col"], t["adj"], t["adj_attr"]
        if name == "tiny_no_col":
            col = np.zeros((2, 0), dtype=np.int64)
        if name == "tiny_no_adj":
            adj, adj_attr = np.zeros((2, 0), dtype=np.int64), np.zeros((0, adj_attr.shape[1]))
    return ref, probs, x, col, adj, adj_attr


LOSS_CASES = ["laby_ref_probs", "laby_3maps", "laby_extreme", "tiny_2maps", "tiny_no_col", "tiny_no_adj"]


@pytest.mark.parametrize("name", LOSS_CASES)
def test_oracle_loss_matches_reference(name):
    """oracle.unsupervised_losses vs Losses.calculate_unsupervised_loss of the reference (losses.py:48-116)."""
    ref, probs, x, col, adj, adj_attr = _loss_case_inputs(name)
    got = orc.unsupervised_losses(torch.from_numpy(probs), torch.from_numpy(x), torch.from_numpy(col),
                                  torch.from_numpy(adj), torch.from_numpy(adj_attr)).numpy()
    want = ref[f"{name}.losses_fp64"]
    assert np.abs(got - want).max() < 1e-12 * np.abs(want).max()
    assert int(np.argsort(got)[0]) == int(ref[f"{name}.min_index_fp64"])
    assert abs(got.min() - float(ref[f"{name}.loss_fp64"])) < 1e-12 * abs(got.min())


# ------------------------------------------------------------------------------------------ greedy assembly loop
def _greedy_fake_probs(x, adj, adj_attr, col, col_attr):
    """The fake predictor of tests/golden/generate_greedy_golden.py (FakeSolver.predict + fake_probs)."""
    n = x.shape[0]
    if col.size == 0 or adj.size == 0:                       # ml_solver.py:31-32
        return np.ones(n)
    deg_a = np.bincount(adj[1], minlength=n)
    deg_c = np.bincount(col[1], minlength=n)
    t = np.sin(12.9898 * x[:, -1] + 78.233 * deg_a + 37.719 * deg_c + 0.37 * np.arange(n)) * 43758.5453
    return 0.05 + 0.9 * (t - np.floor(t))


@pytest.mark.parametrize("k", [0, 1, 2])
def test_oracle_sub_layout_matches_reference(k):
    """greedy_oracle.compute_sub_layout vs BrickLayout.compute_sub_layout of the reference (brick_layout.py:248-286)."""
    from oracle import greedy_oracle as go
    ref = load_npz("ref_greedy.npz")
    g = load_labyrinth_graph()
    n = g["x"].shape[0]
    unl = np.setdiff1d(np.arange(n), ref[f"sub{k}.labelled"])
    x2, adj2, aa2, col2, ca2, inv = go.compute_sub_layout(g["x"], g["adj"], g["adj_attr"], g["col"], g["col_attr"], unl)
    np.testing.assert_array_equal(x2, ref[f"sub{k}.x"])
    np.testing.assert_array_equal(adj2, ref[f"sub{k}.adj"])
    np.testing.assert_array_equal(aa2, ref[f"sub{k}.adj_attr"])
    np.testing.assert_array_equal(col2, ref[f"sub{k}.col"])
    np.testing.assert_array_equal(ca2, ref[f"sub{k}.col_attr"])
    np.testing.assert_array_equal(inv, ref[f"sub{k}.inverse"])


@pytest.mark.parametrize("seed", [0, 7])
def test_oracle_greedy_loop_matches_reference(seed):
    """greedy_oracle.greedy_solve vs solve_by_probablistic_greedy of the reference (algorithms.py:18-62) on numpy's
    global RNG stream: same selection, same order, same sub-layout sizes in every round."""
    from oracle import greedy_oracle as go
    ref = load_npz("ref_greedy.npz")
    g = load_labyrinth_graph()
    np.random.seed(seed)
    sel, order, sizes = go.greedy_solve(_greedy_fake_probs, g["x"], g["adj"], g["adj_attr"], g["col"], g["col_attr"])
    np.testing.assert_array_equal(sizes, ref[f"greedy{seed}.sizes"])
    np.testing.assert_array_equal(order, ref[f"greedy{seed}.order"])
    np.testing.assert_array_equal(sel, ref[f"greedy{seed}.selection"])


def test_dedup_nnconv_equals_the_port():
    """The chunked, type-deduplicated NNConv the full-size GPU tests compare against (oracle.nnconv_mean_dedup) is the
    port's `nnconv_mean` (itself pinned against the reference above) on the real labyrinth graph: fp64, every layer
    sampled, difference at rounding level."""
    g = load_labyrinth_graph()
    sd64 = orc.cast_sd(make_state_dict(15, 20, 32, 1, 3, seed=0), torch.float64)
    x, adj, attr, col, _ = graph_tensors(g, torch.float64)
    gen = torch.Generator().manual_seed(1)
    h = torch.randn(x.shape[0], 32, generator=gen, dtype=torch.float64)
    with torch.no_grad():
        for i in (0, 7, 19):
            p = f"brch_1_graph_conv_layers.{i}"
            a = orc.nnconv_mean(h, adj, attr, sd64, p)
            b = orc.nnconv_mean_dedup(h, adj, attr, sd64, p, chunk=1000)
            assert float((a - b).abs().max()) < 1e-12 * float(a.abs().max())
