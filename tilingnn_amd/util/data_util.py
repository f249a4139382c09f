"""Layout producer and brick-layout files (/root/reference/util/data_util.py).

Same functions, same argument meaning, same return values (down to dtype and to the shape-(0,) arrays the reference
hands back for an empty edge list) as the reference's module; what changed is how the work is done:

  generate_brick_layout_data      :164-205  per-edge dict look-ups, a Python loop per tile
      -> one searchsorted over the graph's (u, v) keys + three fancy-index gathers (GraphArrays, tile_graph.py).
  recover_features_from_reindex   :144-162  `edge[0] in tiles_super_set and ...` over ALL edges with a LIST on the right
      (O(E * N); 132 ms for the full 1 254-tile labyrinth graph here) -> a boolean membership mask, O(E + N), 2.3 ms.
  layout_on_device                (new)     the same super-set cut carried out on the GPU: the complete graph is uploaded
      once (`CompleteGraphOnDevice`), a layout is its `alive` mask run through `tgnn_sublayout_compact`
      (csrc/graph_prep.hip) -- the arrays `to_torch_tensor` (:110-117) would have produced, without a host copy.
  write_brick_layout_data / load_brick_layout_data / write_bricklayout / load_bricklayout   :18-78
      the same pickle schema; read through the schema-restricted unpickler of tile_graph.py.
"""
from __future__ import annotations

import os
import pickle
from collections import defaultdict

import numpy as np

from ..tiling.tile_graph import TileGraph, load_schema_pickle, _reference_names

optional_variables_names = ['node_features', 'collide_edge_index', 'collide_edge_features', 'align_edge_index',
                            'align_edge_features', 'predict', 'predict_order', 'target_shape', 'predict_probs']


# ------------------------------------------------------------------------------------------ files
def write_brick_layout_data(save_path, re_index, node_features=None, collide_edge_index=None,
                            collide_edge_features=None, align_edge_index=None, align_edge_features=None, prefix=None,
                            predict=None, predict_order=None, target_shape=None, predict_probs=None):
    """data_util.py:18-33."""
    if not os.path.exists(prefix):
        os.mkdir(prefix)
    given = locals()
    dic = {"re_index": re_index}
    for name in optional_variables_names:
        if given[name] is not None:
            dic[name] = given[name]
    with _reference_names():
        data = pickle.dumps(dic)
    with open(os.path.join(prefix, save_path), "wb") as f:
        f.write(data)


def load_brick_layout_data(save_path):
    """data_util.py:38-56."""
    f = load_schema_pickle(save_path)
    assert 're_index' in f.keys()
    return (f['re_index'],) + tuple(f.get(name) for name in optional_variables_names)


def load_bricklayout(file_path, complete_graph):
    """data_util.py:58-78."""
    from ..tiling.brick_layout import BrickLayout
    re_index, node_features, collide_edge_index, collide_edge_features, align_edge_index, align_edge_features, \
        predict, predict_order, target_polygon, predict_probs = load_brick_layout_data(file_path)
    if node_features is None or collide_edge_index is None or collide_edge_features is None or \
            align_edge_index is None or align_edge_features is None:
        node_features, collide_edge_index, collide_edge_features, align_edge_index, align_edge_features = \
            recover_features_from_reindex(re_index, complete_graph)
    layout = BrickLayout(complete_graph, node_features, collide_edge_index, collide_edge_features, align_edge_index,
                         align_edge_features, re_index)
    if predict is not None:
        layout.predict = predict
    if predict_order is not None:
        layout.predict_order = predict_order
    if target_polygon is not None:
        layout.target_polygon = target_polygon
    if predict_probs is not None:
        layout.predict_probs = predict_probs
    return layout


def write_bricklayout(folder_path, file_name, brick_layout, with_features=True):
    """data_util.py:80-108."""
    feats = dict(node_features=brick_layout.node_feature, collide_edge_index=brick_layout.collide_edge_index,
                 collide_edge_features=brick_layout.collide_edge_features,
                 align_edge_index=brick_layout.align_edge_index,
                 align_edge_features=brick_layout.align_edge_features) if with_features else {}
    write_brick_layout_data(save_path=file_name, re_index=brick_layout.re_index, prefix=folder_path,
                            predict=brick_layout.predict, predict_order=brick_layout.predict_order,
                            target_shape=brick_layout.target_polygon, predict_probs=brick_layout.predict_probs, **feats)


def to_torch_tensor(device, node_feature, align_edge_index, align_edge_features, collide_edge_index,
                    collide_edge_features):
    """data_util.py:110-117."""
    import torch
    return (torch.from_numpy(node_feature).float().to(device), torch.from_numpy(align_edge_index).long().to(device),
            torch.from_numpy(align_edge_features).float().to(device),
            torch.from_numpy(collide_edge_index).long().to(device),
            torch.from_numpy(collide_edge_features).float().to(device))


# ------------------------------------------------------------------------------------------ producer, host
def generate_brick_layout_data(graph: TileGraph, super_tiles: list, collide_edges, adj_edges):
    """data_util.py:164-205.  `collide_edges` / `adj_edges`: lists of (u, v) complete-graph pairs as in the reference.
    Returns node_feature [n, types + 1] f64, collide index [2, Ec] (or shape (0,)), collide features [Ec, F] f64,
    adjacency index, adjacency features (column 1 / max_align_length), re_index."""
    a = graph.arrays
    col = np.asarray(collide_edges, dtype=np.int64).reshape(-1, 2).T
    adj = np.asarray(adj_edges, dtype=np.int64).reshape(-1, 2).T
    super_arr = np.asarray(super_tiles, dtype=np.int64).reshape(-1)
    return _produce(a, super_arr, col, a.edge_rows(0, col), adj, a.edge_rows(1, adj), super_tiles)


def _produce(a, super_arr, col, col_rows, adj, adj_rows, super_tiles):
    n = super_arr.shape[0]
    re_index = defaultdict(int)                                   # :178-180
    for i, t in enumerate(super_tiles):
        re_index[t] = i
    lookup = np.zeros(a.n_tiles, dtype=np.int64)                   # defaultdict(int): unknown tile -> 0
    lookup[super_arr] = np.arange(n)
    node_feature = np.zeros((n, a.tile_type_count + 1))            # :183-187
    node_feature[np.arange(n), a.tile_ids[super_arr]] = 1
    node_feature[:, -1] = a.tile_areas[super_arr] / a.max_area
    colf = a.colli_features[col_rows]
    adjf = a.adj_features[adj_rows]
    if adjf.shape[0] > 0:
        adjf[:, 1] = adjf[:, 1] / a.max_align_length               # :168-169
    empty = np.array([])                                          # np.array([]).T of an empty list: shape (0,), float64
    return (node_feature,
            lookup[col] if col.shape[1] else empty.copy(), colf if col.shape[1] else empty.copy(),
            lookup[adj] if adj.shape[1] else empty.copy(), adjf if adj.shape[1] else empty.copy(), re_index)


def filter_edges(graph: TileGraph, tiles_super_set):
    """The two comprehensions of tile_factory.py:42-45 / data_util.py:148-151 as a membership mask: rows (into the
    graph's edge arrays) of the edges with both ends in the super set, in file order."""
    a = graph.arrays
    member = np.zeros(a.n_tiles, dtype=bool)
    member[np.asarray(tiles_super_set, dtype=np.int64).reshape(-1)] = True
    col_rows = np.flatnonzero(member[a.colli_edges[0]] & member[a.colli_edges[1]])
    adj_rows = np.flatnonzero(member[a.adj_edges[0]] & member[a.adj_edges[1]])
    return col_rows, adj_rows


def recover_features_from_reindex(re_index, complete_graph: TileGraph):
    """data_util.py:144-162."""
    tiles_super_set = list(re_index.keys())
    a = complete_graph.arrays
    col_rows, adj_rows = filter_edges(complete_graph, tiles_super_set)
    out = _produce(a, np.asarray(tiles_super_set, dtype=np.int64).reshape(-1), a.colli_edges[:, col_rows], col_rows,
                   a.adj_edges[:, adj_rows], adj_rows, tiles_super_set)
    for key, item in out[5].items():
        assert re_index[key] == item                               # :158-159
    return out[:5]


def create_brick_layout_from_super_set(graph: TileGraph, tiles_super_set):
    """tile_factory.py:49-58 after the polygon test: the six producer outputs for a given list of tiles.  (Which tiles
    lie inside a target polygon is a shapely question, tile_factory.py:39 -- not answered here.)"""
    a = graph.arrays
    col_rows, adj_rows = filter_edges(graph, tiles_super_set)
    return _produce(a, np.asarray(tiles_super_set, dtype=np.int64).reshape(-1), a.colli_edges[:, col_rows], col_rows,
                    a.adj_edges[:, adj_rows], adj_rows, list(tiles_super_set))


# ------------------------------------------------------------------------------------------ producer, device
class CompleteGraphOnDevice:
    """The complete graph resident in HBM: node features of ALL tiles, both edge lists, the normalised adjacency
    features, each converted exactly as `to_torch_tensor` converts a layout (float64 -> .float(), int -> .long()).
    Edge subsetting commutes with that conversion, so a layout cut out of these arrays on the device is bit-identical
    to the reference's host-side producer followed by the upload."""

    def __init__(self, graph: TileGraph, device):
        import torch
        from .algorithms import DeviceLayout, SubLayoutBuilder
        a = graph.arrays
        self.device = torch.device(device)
        self.n_tiles = a.n_tiles
        t = lambda arr, dt: torch.from_numpy(np.ascontiguousarray(arr)).to(dt).to(self.device)
        self.full = DeviceLayout(t(a.node_features(), torch.float32), t(a.adj_edges, torch.int64),
                                 t(a.adj_features_normalised(), torch.float32), t(a.colli_edges, torch.int64))
        self.adj_type = t(a.adj_type, torch.int32)
        self._builder = SubLayoutBuilder(self.full)
        self._alive = torch.zeros(self.n_tiles, dtype=torch.int32, device=self.device)

    def layout(self, tiles_super_set):
        """DeviceLayout of the given tiles (ASCENDING complete-graph ids: the order get_all_placement_in_polygon,
        tile_factory.py:39, and compute_sub_layout, brick_layout.py:250-252, both produce).  `inverse_index` maps layout
        node -> complete-graph tile (BrickLayout.inverse_index, brick_layout.py:42-44).  The result aliases this object's
        buffers: valid until the next call."""
        import torch
        s = np.asarray(tiles_super_set, dtype=np.int64).reshape(-1)
        if s.size and (np.any(np.diff(s) <= 0) or s[0] < 0 or s[-1] >= self.n_tiles):
            raise ValueError("the device producer takes strictly ascending tile ids; use "
                             "create_brick_layout_from_super_set for an arbitrary order")
        self._alive.zero_()
        if s.size:
            self._alive[torch.from_numpy(s).to(self.device)] = 1
        return self._builder.build(self._alive)


def layout_on_device(graph_on_device: CompleteGraphOnDevice, tiles_super_set):
    return graph_on_device.layout(tiles_super_set)
