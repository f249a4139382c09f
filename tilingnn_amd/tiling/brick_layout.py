"""`BrickLayout` -- the data side (/root/reference/tiling/brick_layout.py:22-56, :242-286).

A layout is five numpy arrays + the index maps to the complete graph.  The reference's class also draws, unions and
measures polygons through shapely (`show_*`, `get_super_contour_poly`, `detect_holes`, ...): geometry is outside this
package and those methods are not mirrored -- hand the arrays of this object to the reference's class when they are
needed (same constructor arguments).  What `ML_Solver.predict` and the greedy loop read is here:
`get_data_as_torch_tensor` (:242-246) and `compute_sub_layout` (:248-286, vectorised: one membership mask instead of
four comprehensions over every edge).
"""
import copy
from collections import defaultdict

import numpy as np


class BrickLayout:
    def __init__(self, complete_graph, node_feature, collide_edge_index, collide_edge_features, align_edge_index,
                 align_edge_features, re_index, target_polygon=None):
        self.complete_graph = complete_graph
        self.node_feature = node_feature
        self.collide_edge_index = collide_edge_index
        self.collide_edge_features = collide_edge_features
        self.align_edge_index = align_edge_index
        self.align_edge_features = align_edge_features
        self.re_index = re_index                                   # complete-graph tile -> layout node (:37-38)
        self.inverse_index = defaultdict(int)                      # layout node -> complete-graph tile (:40-43)
        for k, v in self.re_index.items():
            self.inverse_index[v] = k
        self.predict = np.zeros(len(self.node_feature))
        self.predict_probs = []
        self.predict_order = []
        self.target_polygon = target_polygon
        self.super_contour_poly = None

    def __deepcopy__(self, memo):                                  # :57-73: arrays shared, predictions copied
        new = type(self).__new__(self.__class__)
        new.__dict__.update(self.__dict__)
        new.predict = copy.deepcopy(self.predict)
        new.predict_probs = copy.deepcopy(self.predict_probs)
        return new

    def is_solved(self):                                           # :75-76
        return len(self.predict) != 0

    def get_data_as_torch_tensor(self, device):
        """:242-246."""
        from ..util.data_util import to_torch_tensor
        return to_torch_tensor(device, self.node_feature, self.align_edge_index, self.align_edge_features,
                               self.collide_edge_index, self.collide_edge_features)

    def compute_sub_layout(self, predict):
        """:248-286.  `predict`: a SelectionSolution-like object with dicts `labelled_nodes` / `unlabelled_nodes`."""
        assert len(self.node_feature) == len(predict.labelled_nodes) + len(predict.unlabelled_nodes)
        sorted_items = sorted(predict.unlabelled_nodes.items(), key=lambda kv: kv[0])         # :250-252
        predict.unlabelled_nodes.clear()
        predict.unlabelled_nodes.update(sorted_items)
        keep = np.fromiter(predict.unlabelled_nodes.keys(), dtype=np.int64, count=len(sorted_items))
        n = len(self.node_feature)
        new_id = np.full(n, -1, dtype=np.int64)
        new_id[keep] = np.arange(keep.shape[0])

        def cut(index, feats):
            index, feats = np.asarray(index), np.asarray(feats)
            if index.shape[0] == 0:                                # the reference's `else np.array([])` branches
                return np.array([]), np.array([])
            a, b = new_id[index[0]], new_id[index[1]]
            alive = (a >= 0) & (b >= 0)
            if not alive.any():
                return np.array([]), np.array([])                 # np.array([]).T / np.array([]) of empty lists
            return np.stack([a[alive], b[alive]]), feats[alive]

        col, colf = cut(self.collide_edge_index, self.collide_edge_features)
        adj, adjf = cut(self.align_edge_index, self.align_edge_features)
        node_inverse_index = {i: int(k) for i, k in enumerate(keep)}                          # :276-278
        fixed_re_index = {self.inverse_index[int(k)]: i for i, k in enumerate(keep)}         # :280-282
        return BrickLayout(self.complete_graph, self.node_feature[keep], col, colf, adj, adjf, fixed_re_index,
                           target_polygon=self.target_polygon), node_inverse_index

    @staticmethod
    def assert_equal_layout(a, b):                                 # :288-300
        for name in ("node_feature", "collide_edge_index", "collide_edge_features", "align_edge_index",
                     "align_edge_features"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), name
        assert dict(a.re_index) == dict(b.re_index)
