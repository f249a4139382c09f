"""`Losses.solution_score` (/root/reference/solver/ml_solver/losses.py:120-148).

Expected scores come from the REFERENCE's own function (tests/golden/generate_score_golden.py, ref_scores.npz) on
layouts its own producer cut out of complete_graph_small.pkl; the one shapely number (the super contour's area) is an
input of the fixture.  CPU: the numpy restatement and the perimeter mirror; GPU: the product path through the C ABI."""
import os
from collections import defaultdict

import numpy as np
import pytest
import torch

from tests.golden_util import GOLDEN, load_npz

SMALL = os.path.join(GOLDEN, "complete_graph_small.pkl")
CASES = ["all", "first80", "random60", "shuffled40", "single", "no_adj"]


def _graph():
    from tilingnn_amd.tiling.tile_graph import TileGraph
    g = TileGraph(2)
    g.load_graph_state(SMALL, sidecar=False)
    return g


def _layout(g, ref, case):
    from tilingnn_amd.tiling.brick_layout import BrickLayout
    from tilingnn_amd.util import data_util as du
    re_index = defaultdict(int)
    for i, t in enumerate(ref[f"{case}.super_tiles"]):
        re_index[int(t)] = i
    x, ci, cf, ai, af = du.recover_features_from_reindex(re_index, g)
    if int(ref[f"{case}.drop_adj"]):
        ai, af = np.zeros((2, 0), dtype=np.int64), np.zeros((0, g.total_feature_dim))
    layout = BrickLayout(g, x, ci, cf, ai, af, re_index)
    layout.super_contour_area = float(ref[f"{case}.contour_area"])
    return layout


def test_perimeters_match_the_reference_tiles():
    ref = load_npz("ref_scores.npz")
    g = _graph()
    got = np.array([t.get_perimeter() for t in g.tiles])
    assert np.array_equal(got, ref["perimeters"])               # same vertices, same operations, same order


@pytest.mark.parametrize("case", CASES)
def test_oracle_score_equals_reference(case):
    from oracle import greedy_oracle as go
    ref = load_npz("ref_scores.npz")
    g = _graph()
    lay = _layout(g, ref, case)
    per = np.array([g.tiles[lay.inverse_index[i]].get_perimeter() for i in range(lay.node_feature.shape[0])])
    got = go.solution_score(ref[f"{case}.predict"], lay.node_feature, lay.align_edge_index, lay.align_edge_features, per,
                            g.max_area, g.max_align_length, lay.super_contour_area)
    want = float(ref[f"{case}.score"])
    assert abs(got - want) <= 2e-7 * max(1.0, abs(want)), (got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_score_equals_reference(case):
    from tilingnn_amd.solver.ml_solver.losses import Losses
    ref = load_npz("ref_scores.npz")
    g = _graph()
    lay = _layout(g, ref, case)
    got = Losses.solution_score(ref[f"{case}.predict"], lay, device=torch.device("cuda:0"))
    want = float(ref[f"{case}.score"])
    assert isinstance(got, float) and abs(got - want) <= 5e-7 * max(1.0, abs(want)), (got, want)
    # the contour area can also be passed in; without any the error says what is missing
    again = Losses.solution_score(ref[f"{case}.predict"], lay, super_contour_area=lay.super_contour_area,
                                  device=torch.device("cuda:0"))
    assert again == got
    lay.super_contour_area = None
    with pytest.raises(ValueError, match="super contour"):
        Losses.solution_score(ref[f"{case}.predict"], lay, device=torch.device("cuda:0"))


@pytest.mark.gpu
def test_score_and_loss_report_out_of_range_edges():
    """torch.gather raises on an edge end outside [0, N) (losses.py:70-73, :133-136); the kernels skip such edges,
    never touch memory out of range, and the wrappers raise IndexError."""
    from tilingnn_amd.solver.ml_solver.losses import Losses
    ref = load_npz("ref_scores.npz")
    g = _graph()
    lay = _layout(g, ref, "first80")
    lay.align_edge_index = lay.align_edge_index.copy()
    lay.align_edge_index[1, 3] = 80
    with pytest.raises(IndexError):
        Losses.solution_score(ref["first80.predict"], lay, device=torch.device("cuda:0"))
    dev = torch.device("cuda:0")
    x, adj, attr, col, _ = lay.get_data_as_torch_tensor(dev)
    p = torch.rand(80, 2, device=dev)
    with pytest.raises(IndexError):
        Losses.calculate_unsupervised_loss(p, x, col, adj, attr)
    bad_col = col.clone()
    bad_col[0, 0] = -1
    lay2 = _layout(g, ref, "first80")
    x, adj, attr, _, _ = lay2.get_data_as_torch_tensor(dev)
    with pytest.raises(IndexError):
        Losses.calculate_unsupervised_loss(p, x, bad_col, adj, attr)


@pytest.mark.gpu
def test_greedy_solve_returns_the_score():
    """create_solution (util/algorithms.py:210-220): the greedy loop's second return value is solution_score of its
    selection -- a float, as in the reference -- when the layout carries its complete graph and contour area."""
    from tilingnn_amd.solver.ml_solver.losses import Losses
    from tilingnn_amd.util.algorithms import solve_by_probablistic_greedy
    ref = load_npz("ref_scores.npz")
    g = _graph()
    lay = _layout(g, ref, "all")

    class FakeSolver:
        device = torch.device("cuda:0")

        def predict(self, layout):
            n = int(layout.node_feature.shape[0])
            return np.linspace(0.2, 0.9, n)

    np.random.seed(3)
    selection, score, order = solve_by_probablistic_greedy(FakeSolver(), lay)
    assert isinstance(score, float) and 0.0 < score < 1.1
    assert score == Losses.solution_score(selection, lay, device=torch.device("cuda:0"))
