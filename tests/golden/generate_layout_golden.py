"""Golden vectors for the input producer and the on-disk formats, produced by the
REFERENCE itself.

Run in the build container only (needs /root/reference):   python tests/golden/generate_layout_golden.py

Imported UNCHANGED: `tiling/tile_graph.py` (TileGraph.load_graph_state :311-332, save_current_state :296-309) and
`util/data_util.py` (generate_brick_layout_data :164-205, recover_features_from_reindex :144-162,
write_brick_layout_data :18-33, load_brick_layout_data :38-56).  What the image lacks is stood in for at import time
only: shapely / PyQt5 / torch_geometric are the attribute-absorbing dummies of generate_greedy_golden.py, EXCEPT
`shapely.geometry.polygon.Polygon`, which has to exist as a class for the pickle to load: the stand-in keeps the WKB
bytes the pickle carries and answers `.area` with the ring-area formula of GEOS (x translated by the first vertex:
sum (x_i - x_0) (y_{i-1} - y_{i+1}) / 2).  That formula is checked below against the one shapely number the pickle
holds -- `max_area`, which the reference computed with the real shapely -- and reproduces it to the bit.

Stored
  ref_layouts.npz          for super sets of the labyrinth graph (first 200 tiles, 400 random, all 1254, one isolated
                           tile, a shuffled set): the six outputs of recover_features_from_reindex, float64 as returned;
                           the same five kinds of super set on complete_graph_small.pkl (`small.*`: what the tests that
                           must run without /root/reference use), and one direct generate_brick_layout_data call
                           with edge lists (`small.direct.*`).
  complete_graph_small.pkl the first 150 tiles of data/labyrinth/complete_graph_ring9.pkl and the edges among them,
                           written by the reference's own save_current_state (same schema, same classes by name).
  layout_with_features.pkl / layout_reindex_only.pkl
                           written by the reference's write_brick_layout_data for the 200-tile super set.
"""
import os
import pickle
import struct
import sys
import types
from collections import defaultdict

import numpy as np
import torch  # noqa: F401  (before the reference tree goes on sys.path)

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import generate_greedy_golden as ggg                           # noqa: E402  (the import stubs)


class Polygon:
    """Stand-in for shapely.geometry.polygon.Polygon: WKB in, GEOS ring area out."""

    def __init__(self, *a, **k):
        self.wkb = b""

    def __setstate__(self, st):
        self.wkb = bytes(st)

    def __reduce__(self):
        return (Polygon, (), self.wkb)

    @property
    def area(self):
        w = self.wkb
        bo = "<" if w[0] == 1 else ">"
        gtype, nrings = struct.unpack(bo + "II", w[1:9])
        assert gtype == 3 and nrings == 1
        (npts,) = struct.unpack(bo + "I", w[9:13])
        p = np.frombuffer(w[13:13 + 16 * npts], dtype=bo + "f8").reshape(npts, 2)
        s, x0 = 0.0, p[0, 0]
        for i in range(1, npts - 1):
            s += (p[i, 0] - x0) * (p[i - 1, 1] - p[i + 1, 1])
        return abs(s / 2.0)


Polygon.__module__ = "shapely.geometry.polygon"


def main():
    ggg.import_reference()
    sys.modules["shapely.geometry.polygon"].Polygon = Polygon
    sys.modules["shapely.geometry"].Polygon = Polygon
    import tiling.tile as rt
    rt.Polygon = Polygon
    from tiling.tile_graph import TileGraph
    import util.data_util as du

    cg = TileGraph(2)
    cg.load_graph_state(os.path.join(REF, "data/labyrinth/complete_graph_ring9.pkl"))
    n = len(cg.tiles)
    assert max(t.area() for t in cg.tiles) == cg.max_area, "the stand-in area formula is not shapely's"

    rng = np.random.default_rng(11)
    shuffled = rng.permutation(n)[:60]
    isolated = [5]
    cases = {
        "first200": list(range(200)),
        "random400": sorted(int(v) for v in rng.choice(n, size=400, replace=False)),
        "all": list(range(n)),
        "isolated": isolated,
        "shuffled60": [int(v) for v in shuffled],
    }
    out = {"max_area": np.float64(cg.max_area), "max_align_length": np.float64(cg.max_align_length),
           "tile_type_count": np.int64(cg.tile_type_count), "n_tiles": np.int64(n),
           "n_adj_edges": np.int64(len(cg.adj_edges)), "n_colli_edges": np.int64(len(cg.colli_edges)),
           "tile_areas": np.array([t.area() for t in cg.tiles]), "tile_ids": np.array([t.id for t in cg.tiles])}
    for name, tiles in cases.items():
        re_index = defaultdict(int)
        for i, t in enumerate(tiles):
            re_index[t] = i
        x, ci, cf, ai, af = du.recover_features_from_reindex(re_index, cg)
        out[f"{name}.super_tiles"] = np.asarray(tiles, dtype=np.int64)
        for key, arr in (("x", x), ("col", ci), ("col_attr", cf), ("adj", ai), ("adj_attr", af)):
            out[f"{name}.{key}"] = np.asarray(arr)
        print(name, x.shape, np.asarray(ci).shape, np.asarray(cf).shape, np.asarray(ai).shape, np.asarray(af).shape,
              np.asarray(ci).dtype, np.asarray(af).dtype)
    # --- the small complete graph, written by the reference's own writer
    k = 150
    small = TileGraph(2)
    small.tiles = cg.tiles[:k]
    small.graph = defaultdict(list, {u: [v for v in vs if v < k] for u, vs in cg.graph.items() if u < k})
    small.edges_features = defaultdict(list)
    for u, row in cg.edges_features.items():
        if u < k:
            small.edges_features[u] = defaultdict(list, {v: f for v, f in row.items() if v < k})
    small.colli_edges = [e for e in cg.colli_edges if e[0] < k and e[1] < k]
    small.adj_edges = [e for e in cg.adj_edges if e[0] < k and e[1] < k]
    small.unique_adj_features = cg.unique_adj_features
    small.max_area, small.max_align_length, small.align_start_index = cg.max_area, cg.max_align_length, cg.align_start_index
    small.save_current_state(os.path.join(HERE, "complete_graph_small.pkl"))
    # ... read back by the reference's own loader; the producer run on it (these cases need no other file)
    sg = TileGraph(2)
    sg.load_graph_state(os.path.join(HERE, "complete_graph_small.pkl"))
    small_cases = {
        "small.first80": list(range(80)),
        "small.random60": sorted(int(v) for v in rng.choice(k, size=60, replace=False)),
        "small.all": list(range(k)),
        "small.isolated": [5],
        "small.shuffled40": [int(v) for v in rng.permutation(k)[:40]],
    }
    for name, tiles in small_cases.items():
        re_index = defaultdict(int)
        for i, t in enumerate(tiles):
            re_index[t] = i
        x, ci, cf, ai, af = du.recover_features_from_reindex(re_index, sg)
        out[f"{name}.super_tiles"] = np.asarray(tiles, dtype=np.int64)
        for key, arr in (("x", x), ("col", ci), ("col_attr", cf), ("adj", ai), ("adj_attr", af)):
            out[f"{name}.{key}"] = np.asarray(arr)
    # generate_brick_layout_data called directly with edge LISTS (the tile_factory.py:49-58 call shape)
    tiles = small_cases["small.first80"]
    ce = [e for e in sg.colli_edges if e[0] < 80 and e[1] < 80][::3]
    ae = [e for e in sg.adj_edges if e[0] < 80 and e[1] < 80][::2]
    x, ci, cf, ai, af, _ = du.generate_brick_layout_data(sg, tiles, ce, ae)
    out["small.direct.col_edges"], out["small.direct.adj_edges"] = np.asarray(ce), np.asarray(ae)
    for key, arr in (("x", x), ("col", ci), ("col_attr", cf), ("adj", ai), ("adj_attr", af)):
        out[f"small.direct.{key}"] = np.asarray(arr)
    np.savez_compressed(os.path.join(HERE, "ref_layouts.npz"), **out)

    # --- brick-layout files, written by the reference's own writer
    tiles = cases["first200"]
    re_index = defaultdict(int)
    for i, t in enumerate(tiles):
        re_index[t] = i
    x, ci, cf, ai, af = du.recover_features_from_reindex(re_index, cg)
    du.write_brick_layout_data("layout_with_features.pkl", re_index, node_features=x, collide_edge_index=ci,
                               collide_edge_features=cf, align_edge_index=ai, align_edge_features=af, prefix=HERE,
                               predict=np.arange(200) % 2, predict_order=[3, 1, 2], predict_probs=[0.5, 0.25])
    du.write_brick_layout_data("layout_reindex_only.pkl", re_index, prefix=HERE)
    for f in ("ref_layouts.npz", "complete_graph_small.pkl", "layout_with_features.pkl", "layout_reindex_only.pkl"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
