"""Golden vectors for `Losses.solution_score`, produced by the REFERENCE itself.

Run in the build container only (needs /root/reference):   python tests/golden/generate_score_golden.py

Imported UNCHANGED: `solver/ml_solver/losses.py` (solution_score :120-148), `tiling/tile.py` (Tile.get_perimeter :41-42
over util/algo_util.py:72-74), `tiling/tile_graph.py` (the loader), `tiling/brick_layout.py`, `util/data_util.py` (the
producer).  shapely / PyQt5 / torch_geometric are the attribute-absorbing dummies of generate_greedy_golden.py; the one
shapely class that has to carry data, `Polygon`, is a WKB holder answering `.area` (GEOS ring formula, checked against
the file's `max_area` by generate_layout_golden.py) and `.exterior.coords` (the ring's vertices).  The one shapely NUMBER of
the function -- `get_super_contour_poly().area`, a polygon union -- is supplied: the layout's cached `super_contour_poly`
is an object whose `.area` is the sum of the super set's tile areas times 0.9 (tiles overlap; any value that keeps the
reference's assert `filled_area <= 1` quiet does), and that same number is stored as the fixture's input.

Stored (ref_scores.npz), for super sets of complete_graph_small.pkl (the fixture that travels): the super set, a
collision-free 0/1 selection (deterministic sweep over the layout's collision edges), the contour area fed in, the
reference's score; plus one selection with a single tile and one layout without adjacency edges.
"""
import os
import struct
import sys
import types
from collections import defaultdict

import numpy as np
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import generate_greedy_golden as ggg                           # noqa: E402  (the import stubs)


class Polygon:
    def __init__(self, *a, **k):
        self.wkb = b""

    def __setstate__(self, st):
        self.wkb = bytes(st)

    def _ring(self):
        w = self.wkb
        bo = "<" if w[0] == 1 else ">"
        gtype, nrings = struct.unpack(bo + "II", w[1:9])
        assert gtype == 3 and nrings == 1
        (npts,) = struct.unpack(bo + "I", w[9:13])
        return np.frombuffer(w[13:13 + 16 * npts], dtype=bo + "f8").reshape(npts, 2)

    @property
    def exterior(self):
        return types.SimpleNamespace(coords=[tuple(map(float, p)) for p in self._ring()])

    @property
    def area(self):
        p = self._ring()
        s, x0 = 0.0, p[0, 0]
        for i in range(1, p.shape[0] - 1):
            s += (p[i, 0] - x0) * (p[i - 1, 1] - p[i + 1, 1])
        return abs(s / 2.0)


Polygon.__module__ = "shapely.geometry.polygon"


def collision_free_selection(n, col, order):
    """Deterministic maximal independent set of the collision graph, nodes visited in `order`."""
    nbrs = defaultdict(list)
    for u, v in zip(*col) if np.asarray(col).size else ():
        nbrs[int(u)].append(int(v))
    sel, dead = np.zeros(n), np.zeros(n, dtype=bool)
    for v in order:
        if not dead[v]:
            sel[v] = 1
            dead[v] = True
            for w in nbrs[int(v)]:
                dead[w] = True
    return sel


def main():
    bl, alg = ggg.import_reference()
    sys.modules["shapely.geometry.polygon"].Polygon = Polygon
    sys.modules["shapely.geometry"].Polygon = Polygon
    import tiling.tile as rt
    rt.Polygon = Polygon
    from tiling.tile_graph import TileGraph
    import util.data_util as du
    from solver.ml_solver.losses import Losses                 # the reference's own code
    import solver.ml_solver.losses as ref_losses
    assert str(ref_losses.device) == "cpu"

    cg = TileGraph(2)
    cg.load_graph_state(os.path.join(HERE, "complete_graph_small.pkl"))
    k = len(cg.tiles)
    rng = np.random.default_rng(17)
    out = {"perimeters": np.array([t.get_perimeter() for t in cg.tiles], dtype=np.float64)}

    def case(name, tiles, order_seed, drop_adj=False, single=False):
        re_index = defaultdict(int)
        for i, t in enumerate(tiles):
            re_index[t] = i
        x, ci, cf, ai, af = du.recover_features_from_reindex(re_index, cg)
        if drop_adj:
            ai, af = np.array([]), np.array([])
        layout = bl.BrickLayout(cg, x, ci, cf, ai, af, re_index)
        n = x.shape[0]
        area = 0.9 * float(sum(cg.tiles[t].area() for t in tiles))
        layout.super_contour_poly = types.SimpleNamespace(area=area)
        order = np.random.default_rng(order_seed).permutation(n)
        sel = collision_free_selection(n, np.asarray(ci).reshape(2, -1) if np.asarray(ci).size else ci, order)
        if single:
            sel = np.zeros(n)
            sel[int(order[0])] = 1
        score = Losses.solution_score(sel, layout)
        out[f"{name}.super_tiles"] = np.asarray(tiles, dtype=np.int64)
        out[f"{name}.predict"] = sel
        out[f"{name}.contour_area"] = np.float64(area)
        out[f"{name}.score"] = np.float64(score)
        out[f"{name}.drop_adj"] = np.int64(drop_adj)
        print(f"{name}: n={n} selected={int(sel.sum())} contour_area={area:.6f} score={score:.9f}")

    case("all", list(range(k)), 1)
    case("first80", list(range(80)), 2)
    case("random60", sorted(int(v) for v in rng.choice(k, size=60, replace=False)), 3)
    case("shuffled40", [int(v) for v in rng.permutation(k)[:40]], 4)
    case("single", list(range(80)), 5, single=True)
    case("no_adj", list(range(80)), 6, drop_adj=True)
    np.savez_compressed(os.path.join(HERE, "ref_scores.npz"), **out)
    print(f"ref_scores.npz: {os.path.getsize(os.path.join(HERE, 'ref_scores.npz')) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
