"""`solve_by_probablistic_greedy` -- the greedy assembly loop around `ML_Solver.predict`
(/root/reference/util/algorithms.py:18-62), with the layout resident on the GPU.

Per round the reference rebuilds the sub-layout of the still unlabelled nodes with four Python comprehensions and
dict look-ups over ALL edges (`BrickLayout.compute_sub_layout`, tiling/brick_layout.py:248-286) and ships it to the
device again.  Here the five arrays of the original layout are uploaded once; a round is
    alive mask -> `tgnn_sublayout_compact` (flags, scans, scatters: csrc/graph_prep.hip) -> `ml_solver.predict` on the
    device-resident sub-layout -> probabilities to the host -> the acceptance sweep.
The sweep itself stays on the host ON PURPOSE: it is sequential by definition (descending probability, stop at the
first node a previous acceptance has killed), touches a handful of nodes per round, and consumes numpy's global RNG
stream one `np.random.uniform()` per visited node (algorithms.py:51) -- the stream the reference consumes, so that a
seeded run selects the same tiles.

Same return values as the reference: (selection_predict, score, predict_order).  The score is
`Losses.solution_score` (losses.py:120-148 -> tilingnn_amd/solver/ml_solver/losses.py, sums on the GPU) whenever the
layout carries what it needs -- its complete graph (tile rings, max_area, max_align_length) and the area of its super
contour (`layout.super_contour_area`, or the reference class's `get_super_contour_poly()`); a bare `DeviceLayout` has
neither and scores `None` unless `score_fn(selection, origin_layout)` is given.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib, ops
from .._lib import check, lib, ptr


class DeviceLayout:
    """The data side of a BrickLayout (brick_layout.py:242-246) living on the GPU: what `ML_Solver.predict` reads."""

    def __init__(self, node_feature, align_edge_index, align_edge_features, collide_edge_index, inverse_index=None):
        self.node_feature = node_feature
        self.align_edge_index = align_edge_index
        self.align_edge_features = align_edge_features
        self.collide_edge_index = collide_edge_index
        self.collide_edge_features = None                     # never read by the network (TilinGNN.py:51)
        self.inverse_index = inverse_index                    # sub-layout node -> original node

    def get_data_as_torch_tensor(self, device):
        return (self.node_feature, self.align_edge_index, self.align_edge_features, self.collide_edge_index,
                self.collide_edge_features)

    @staticmethod
    def upload(layout, device) -> "DeviceLayout":
        """From the numpy arrays of a BrickLayout / LayoutArrays, with the conversion of util/data_util.py:110-117."""
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(device)
        adj = np.asarray(layout.align_edge_index).reshape(2, -1)
        col = np.asarray(layout.collide_edge_index).reshape(2, -1)
        fe = np.asarray(layout.align_edge_features).shape[-1] if np.asarray(layout.align_edge_features).size else 1
        return DeviceLayout(t(layout.node_feature, torch.float32), t(adj, torch.int64),
                            t(np.asarray(layout.align_edge_features).reshape(-1, fe), torch.float32), t(col, torch.int64))


class SubLayoutBuilder:
    """compute_sub_layout on the device: buffers sized once for the original layout, reused every round."""

    def __init__(self, origin: DeviceLayout):
        self.o = origin
        x, adj, attr, col = origin.node_feature, origin.align_edge_index, origin.align_edge_features, origin.collide_edge_index
        dev = x.device
        self.n, self.fx = int(x.shape[0]), int(x.shape[1])
        self.ea, self.ec, self.fe = int(adj.shape[1]), int(col.shape[1]), int(attr.shape[1]) if attr.numel() else 1
        self.x_out = torch.empty(self.n, self.fx, dtype=torch.float32, device=dev)
        self.inverse = torch.empty(self.n, dtype=torch.int64, device=dev)
        self.adj_out = torch.empty(2 * max(self.ea, 1), dtype=torch.int64, device=dev)
        self.attr_out = torch.empty(max(self.ea, 1) * self.fe, dtype=torch.float32, device=dev)
        self.col_out = torch.empty(2 * max(self.ec, 1), dtype=torch.int64, device=dev)
        self._tail = torch.zeros(4, dtype=torch.int64, device=dev)       # counts [3] | error flag: ONE read-back per round
        self.counts = self._tail[:3]
        self.err = self._tail[3:].view(torch.int32)[:1]
        self.ws_bytes = lib.tgnn_sublayout_workspace_bytes(self.n, self.ea, self.ec)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self._host = torch.empty(4, dtype=torch.int64, pin_memory=True)    # (`.cpu()` stages through a pageable tensor: ~10 us more per round)

    def build(self, alive: torch.Tensor) -> DeviceLayout:
        """alive: int32 [N] on the device (!= 0 = unlabelled).  One host sync (the three counts)."""
        o = self.o
        check(lib.tgnn_sublayout_compact(ptr(alive), self.n, ptr(o.node_feature), self.fx,
                                         ptr(o.align_edge_index) if self.ea else None, self.ea,
                                         ptr(o.align_edge_features) if self.ea else None, self.fe,
                                         ptr(o.collide_edge_index) if self.ec else None, self.ec,
                                         ptr(self.x_out), ptr(self.inverse), ptr(self.adj_out), ptr(self.attr_out),
                                         ptr(self.col_out), ptr(self.counts), ptr(self.err), ptr(self.ws), self.ws_bytes,
                                         _lib.current_stream(alive.device)))
        self._host.copy_(self._tail, non_blocking=True)
        torch.cuda.current_stream(alive.device).synchronize()
        n2, ea2, ec2, err = self._host.tolist()
        if err:
            raise IndexError("edge index out of range in the layout")
        return DeviceLayout(self.x_out[:n2], self.adj_out[:2 * ea2].view(2, ea2), self.attr_out[:ea2 * self.fe].view(ea2, self.fe),
                            self.col_out[:2 * ec2].view(2, ec2), self.inverse[:n2])


class HostSweep:
    """The state and the acceptance sweep of the reference's loop (algorithms.py:23-54), over the ORIGINAL node numbering: the
    running geometric mean of the probabilities (:33-34), the descending walk that stops at the first node labelled in this
    round (:41-48), the test against numpy's global RNG stream (:51), label_collision_neighbor (:196-207).  Shared by the
    single-GPU loop below and by tilingnn_amd.dist.solve_sharded (every rank runs the same sweep on the gathered probabilities
    with the same seed, so every rank takes the same decisions)."""

    def __init__(self, n: int, collide_edge_index: np.ndarray, uniform=None):
        # uniform: the draw of :51; default numpy's GLOBAL stream, as the reference (np.random.RandomState(seed).uniform gives the
        # stream np.random.seed(seed) would: what thread-simulated ranks, which share one interpreter, use instead)
        self.uniform = uniform if uniform is not None else np.random.uniform
        col = np.asarray(collide_edge_index).reshape(2, -1)
        self.n, self.has_col = n, bool(col.size)
        if self.has_col:                                        # collision neighbours in edge order (algorithms.py:199)
            by_src = np.argsort(col[0], kind="stable")
            self.starts = np.searchsorted(col[0][by_src], np.arange(n + 1))
            self.nbr = col[1][by_src]
        self.prob_saved = np.ones(n)                            # SelectionSolution.unlabelled_nodes (:285)
        self.unlabelled = np.ones(n, dtype=bool)
        self.selection = np.zeros(n)
        self.order = []
        self.round_cnt = 1

    def ids(self) -> np.ndarray:
        return np.flatnonzero(self.unlabelled)

    def round(self, ids: np.ndarray, prob) -> list:
        """One round: `prob` = the network's probabilities of the nodes `ids` (ascending original numbers).  Returns the original
        numbers of the nodes labelled in this round (selected ones and their collision neighbours)."""
        prob = np.asarray(prob, dtype=np.float64).reshape(-1)
        prob_per_node = np.power(np.power(self.prob_saved[ids], self.round_cnt - 1) * prob, 1 / self.round_cnt)     # (:33-34)
        self.prob_saved[ids] = prob_per_node
        unlabelled, killed = self.unlabelled, []
        for idx in np.argsort(-prob_per_node):                  # (:41)
            origin_idx = ids[idx]
            if not unlabelled[origin_idx]:                      # (:47-48)
                break
            if np.exp((prob_per_node[idx] - 1) * 1.0) > self.uniform():     # (:51)
                unlabelled[origin_idx] = False
                self.selection[origin_idx] = 1
                self.order.append(int(origin_idx))
                killed.append(origin_idx)
                if self.has_col:                                # label_collision_neighbor (:196-207)
                    for v in self.nbr[self.starts[origin_idx]:self.starts[origin_idx + 1]]:
                        if unlabelled[v]:
                            unlabelled[v] = False
                            killed.append(v)
        self.round_cnt += 1
        return killed


def solve_by_probablistic_greedy(ml_solver, origin_layout, score_fn=None, on_round=None):
    """algorithms.py:18-62.  `origin_layout`: BrickLayout-like numpy arrays (uploaded once) or a DeviceLayout."""
    device = ml_solver.device
    origin = origin_layout if isinstance(origin_layout, DeviceLayout) else DeviceLayout.upload(origin_layout, device)
    n = int(origin.node_feature.shape[0])
    sweep = HostSweep(n, origin.collide_edge_index.cpu().numpy())
    builder = SubLayoutBuilder(origin)
    alive_dev = torch.ones(n, dtype=torch.int32, device=origin.node_feature.device)
    while sweep.unlabelled.any():
        temp_layout = builder.build(alive_dev)
        ids = sweep.ids()                                       # == temp_layout.inverse_index (kept on the device)
        if on_round is not None:
            on_round(temp_layout)
        killed = sweep.round(ids, ml_solver.predict(temp_layout))
        if killed:
            alive_dev[torch.from_numpy(np.asarray(killed, dtype=np.int64)).to(alive_dev.device)] = 0
    score = create_score(sweep.selection, origin_layout, score_fn, device)
    return sweep.selection, score, sweep.order


def solve_by_device_greedy(ml_solver, origin_layout, seed=0, score_fn=None, on_round=None, max_rounds=100000, finish=True):
    """The assembly loop with the acceptance BATCHED on the device (csrc/greedy.hip: tgnn_greedy_round) -- the documented
    substitute of the reference's sequential sweep (algorithms.py:41-54) for large layouts (BASELINE config 5: "batched greedy
    selection"): per round every node that precedes all its unlabelled collision neighbours in the reference's visiting order
    and passes the reference's test exp(p - 1) > u is accepted at once, u from a counter-based generator keyed by (seed, round,
    node).  NOT the reference's RNG stream -- `solve_by_probablistic_greedy` stays the default where seeded parity matters --
    but the same invariants: a collision-free selection, maximal when the loop ends, the same running geometric mean of the
    probabilities (:33-34).  O(log N) rounds of {compaction, forward, four small launches}; nothing but one count per round
    travels to the host.  Same return values: (selection, score, predict_order); the order = by round, then by node number.
    finish (no `on_round` given): once a sub-layout has no adjacency edge or no collision edge left -- from there on ML_Solver.predict
    answers 1 for every node without the network (ml_solver.py:31-32) -- the remaining rounds run as ONE launch on that sub-layout
    (tgnn_greedy_finish: the same means, order, draws and round numbers, hence the same selection and round count)."""
    device = ml_solver.device
    origin = origin_layout if isinstance(origin_layout, DeviceLayout) else DeviceLayout.upload(origin_layout, device)
    dev = origin.node_feature.device
    n = int(origin.node_feature.shape[0])
    builder = SubLayoutBuilder(origin)
    alive = torch.ones(n, dtype=torch.int32, device=dev)
    selected = torch.zeros(n, dtype=torch.int32, device=dev)
    saved = torch.ones(n, dtype=torch.float64, device=dev)
    tail = torch.zeros(2, dtype=torch.int64, device=dev)        # accepted so far | error flag
    ws_bytes = int(lib.tgnn_greedy_round_workspace_bytes(n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    rounds = 0
    while True:
        sub = builder.build(alive)                              # (one sync: the sub-layout's sizes)
        n2 = int(sub.node_feature.shape[0])
        if n2 == 0:
            break
        rounds += 1
        if rounds > max_rounds:
            raise RuntimeError(f"solve_by_device_greedy: {n2} nodes still unlabelled after {max_rounds} rounds")
        if on_round is not None:
            on_round(sub)
        ea2, ec2 = int(sub.align_edge_index.shape[1]), int(sub.collide_edge_index.shape[1])
        if finish and on_round is None and (ea2 == 0 or ec2 == 0) and n2 <= int(lib.tgnn_greedy_finish_max_nodes()):
            out = torch.zeros(2, dtype=torch.int32, device=dev)
            check(lib.tgnn_greedy_finish(ptr(sub.inverse_index), n2, ptr(sub.collide_edge_index) if ec2 else None, ec2, rounds,
                                         max_rounds - rounds + 1, int(seed) & (2 ** 64 - 1), ptr(saved), ptr(alive), ptr(selected),
                                         ptr(tail[:1]), ptr(tail[1:].view(torch.int32)[:1]), ptr(out), _lib.current_stream(dev)))
            ran, left = out.cpu().tolist()
            rounds += ran - 1
            if left:
                raise RuntimeError(f"solve_by_device_greedy: {left} nodes still unlabelled after {max_rounds} rounds")
            break
        probs = ml_solver.predict_on_device(sub)                # [n2] float32 on the device
        ec2 = int(sub.collide_edge_index.shape[1])
        check(lib.tgnn_greedy_round(ptr(probs), 1, ptr(sub.inverse_index), n2, ptr(sub.collide_edge_index) if ec2 else None, ec2,
                                    rounds, int(seed) & (2 ** 64 - 1), ptr(saved), ptr(alive), ptr(selected), ptr(tail[:1]),
                                    ptr(tail[1:].view(torch.int32)[:1]), ptr(ws), ws_bytes, _lib.current_stream(dev)))
    sel_round = selected.cpu().numpy()
    if int(tail[1].item()):
        raise IndexError("collision edge index out of range in a sub-layout")
    selection = (sel_round > 0).astype(np.float64)
    picked = np.flatnonzero(sel_round > 0)
    order = [int(v) for v in picked[np.lexsort((picked, sel_round[picked]))]]
    score = create_score(selection, origin_layout, score_fn, device)
    solve_by_device_greedy.last_rounds = rounds
    return selection, score, order


def create_score(selection, origin_layout, score_fn=None, device=None):
    """The score half of `create_solution` (algorithms.py:210-220)."""
    if score_fn is not None:
        return score_fn(selection, origin_layout)
    has_area = getattr(origin_layout, "super_contour_area", None) is not None or hasattr(origin_layout, "get_super_contour_poly")
    if getattr(origin_layout, "complete_graph", None) is None or not has_area:
        return None
    from ..solver.ml_solver.losses import Losses
    return Losses.solution_score(selection, origin_layout, device=device)
