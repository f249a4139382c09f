"""CPU restatement (numpy) of the greedy assembly loop of TilinGNN -- TEST INFRASTRUCTURE ONLY.

PINNED: tests/test_oracle_vs_reference_golden.py checks every function here against tests/golden/ref_greedy.npz, which
tests/golden/generate_greedy_golden.py produced by running the REFERENCE's own code (tiling/brick_layout.py and
util/algorithms.py imported unchanged).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (tilingnn_amd/) never does.

Restated:
  compute_sub_layout      BrickLayout.compute_sub_layout      /root/reference/tiling/brick_layout.py:248-286
  greedy_solve            solve_by_probablistic_greedy        /root/reference/util/algorithms.py:18-62
                          label_collision_neighbor            /root/reference/util/algorithms.py:196-207
                          SelectionSolution                   /root/reference/util/algorithms.py:282-294
"""
import numpy as np


def compute_sub_layout(x, adj, adj_attr, col, col_attr, unlabelled):
    """Sub-layout of the `unlabelled` nodes (ascending original ids), brick_layout.py:248-286: node i of the result is
    unlabelled[i]; an edge survives iff both ends are unlabelled; survivors keep their order with re-indexed ends.
    Returns (x', adj', adj_attr', col', col_attr', inverse) with inverse[i] = original id."""
    unlabelled = np.asarray(unlabelled, dtype=np.int64)
    assert np.all(np.diff(unlabelled) > 0), "the reference sorts the unlabelled nodes by id first (:250-252)"
    n = x.shape[0]
    re_index = np.full(n, -1, dtype=np.int64)
    re_index[unlabelled] = np.arange(unlabelled.shape[0])

    def cut(ei, attr):
        if ei.size == 0:
            return np.zeros((2, 0), dtype=np.int64), attr[:0]
        keep = (re_index[ei[0]] >= 0) & (re_index[ei[1]] >= 0)
        return re_index[ei[:, keep]], attr[keep]

    adj2, adj_attr2 = cut(adj, adj_attr)
    col2, col_attr2 = cut(col, col_attr)
    return x[unlabelled], adj2, adj_attr2, col2, col_attr2, unlabelled.copy()


def greedy_solve(predict, x, adj, adj_attr, col, col_attr, uniform=np.random.uniform):
    """solve_by_probablistic_greedy (algorithms.py:18-62).  `predict(x', adj', adj_attr', col', col_attr')` -> float
    array [N'] (ML_Solver.predict of the sub-layout).  Consumes `uniform()` exactly as the reference consumes
    np.random.uniform(): one draw per node that reaches the acceptance test.
    Returns (selection [N] in {0,1}, order of selection, per-round (N', Ea', Ec'))."""
    n = x.shape[0]
    prob_saved = np.ones(n)                     # SelectionSolution.unlabelled_nodes values (:285)
    unlabelled = np.ones(n, dtype=bool)
    selection = np.zeros(n, dtype=np.int8)
    order, sizes = [], []
    # collision neighbours of a node in edge order: collision_edges[1][collision_edges[0] == idx] (:199)
    if col.size:
        by_src = np.argsort(col[0], kind="stable")
        src_sorted = col[0][by_src]
        starts = np.searchsorted(src_sorted, np.arange(n + 1))
        nbr = col[1][by_src]
    round_cnt = 1
    while unlabelled.any():
        ids = np.flatnonzero(unlabelled)                                         # sorted by key (:250-252)
        sub = compute_sub_layout(x, adj, adj_attr, col, col_attr, ids)
        sizes.append((sub[0].shape[0], sub[1].shape[1], sub[3].shape[1]))
        prob = np.asarray(predict(*sub[:5]), dtype=np.float64)
        previous = prob_saved[ids]
        prob_per_node = np.power(np.power(previous, round_cnt - 1) * prob, 1 / round_cnt)   # (:33-34)
        prob_saved[ids] = prob_per_node                                          # (:37-38)
        for idx in np.argsort(-prob_per_node):                                   # (:41)
            origin = ids[idx]
            if not unlabelled[origin]:                                           # collision handling: stop the sweep (:47-48)
                break
            if np.exp((prob_per_node[idx] - 1) * 1.0) > uniform():               # (:51)
                unlabelled[origin] = False
                selection[origin] = 1
                order.append(int(origin))
                if col.size:                                                     # label_collision_neighbor (:196-207)
                    for v in nbr[starts[origin]:starts[origin + 1]]:
                        if unlabelled[v]:
                            unlabelled[v] = False
        round_cnt += 1
    return selection, np.asarray(order, dtype=np.int64), np.asarray(sizes, dtype=np.int64)


def solution_score(predict, x, adj, adj_attr, perimeters, max_area, max_align_length, contour_area,
                   weights=(1.0, 0.02)):
    """Losses.solution_score, /root/reference/solver/ml_solver/losses.py:120-148, in the reference's float32:
    AVG_AREA_WEIGHT * predict . (x[:, -1] max_area) / contour_area
      + ALIGN_LENGTH_WEIGHT * ((p_i p_j) . (len max_align_length)) / sum of the selected tiles' perimeters.
    `perimeters` [N] float64 (Tile.get_perimeter of every layout node); weights = (AVG_AREA_WEIGHT, ALIGN_LENGTH_WEIGHT).
    PINNED by tests/golden/ref_scores.npz (tests/golden/generate_score_golden.py runs the reference's own function)."""
    p = np.asarray(predict, dtype=np.float32)
    x = np.asarray(x, dtype=np.float32)
    filled_area = np.dot(p, x[:, -1] * np.float32(max_area)) / contour_area                  # :126
    adj = np.asarray(adj)
    if adj.size:
        lengths = np.asarray(adj_attr, dtype=np.float32)[:, 1] * np.float32(max_align_length)   # :131
        align = np.dot(p[adj[0]] * p[adj[1]], lengths)                                      # :132-141
    else:
        align = 0.0
    all_edge_length = sum(float(perimeters[i]) for i in range(p.shape[0]) if p[i] == 1)      # :143-144
    return float(weights[0] * filled_area + weights[1] * (align / all_edge_length))         # :148
