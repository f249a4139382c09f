"""`TileGraph` -- the complete graph of candidate tile placements, read side only
(/root/reference/tiling/tile_graph.py:60-332).

The reference BUILDS this graph with shapely polygon intersections (`_form__graph`, :139-190) and stores it with
`save_current_state` (:296-309) as one pickle: tiles (shapely polygons), adjacency lists, a dict-of-dicts of edge
feature rows, the two edge lists, the adjacency feature codebook and three scalars.  Building is outside this
package (polygon boolean operations) -- LOADING is what feeds the scoring path, and here it
works without shapely and without executing anything the file names:

  * `load_graph_state(path)` (:311-332) reads the reference's pickle through an unpickler that resolves exactly the
    seven globals that schema uses (`tiling.tile.Tile` and `shapely.geometry.polygon.Polygon` to the two plain classes
    below, defaultdict/list/int and numpy's scalar/array reconstructors to themselves) and refuses everything else;
  * tile polygons stay WKB -> vertex arrays; `Tile.area()` (tile.py:37-39) is the ring area in GEOS's order of
    operations (x translated by the first vertex), which reproduces the `max_area` the reference stored (computed by the
    real shapely) to the last bit -- tests/test_layout_formats.py;
  * besides the reference's list/dict attributes (same names, same contents) the graph carries ARRAY forms:
    edge lists [2, E] int64, feature rows [E, F] float64 and the edge-type id of every adjacency edge taken from the
    one-hot columns, which is what the producer (util/data_util.py) and the device path work on;
  * `save_sidecar` / automatic `<pickle>.tgnn.npz`: the parsed arrays next to the pickle (validated by size + mtime of
    the pickle), so that a second load does not touch the pickle at all.
"""
from __future__ import annotations

import io
import os
import pickle
import struct
from collections import defaultdict

import numpy as np

SIDECAR_SUFFIX = ".tgnn.npz"
SIDECAR_VERSION = 1


class Polygon:
    """What this package keeps of a shapely polygon: the rings, as float64 [k, 2] arrays (closed: last == first)."""
    __slots__ = ("wkb", "_rings")

    def __init__(self, *a, **k):
        if a or k:
            raise NotImplementedError("polygons are only ever unpickled here, never constructed")
        self.wkb, self._rings = b"", None

    def __setstate__(self, state):                       # shapely 1.x: __reduce__ -> (Polygon, (), wkb bytes)
        if not isinstance(state, (bytes, bytearray)):
            raise pickle.UnpicklingError("polygon state is not WKB")
        self.wkb, self._rings = bytes(state), None

    def __reduce__(self):
        if not self.wkb and self._rings is not None:     # came from the side-car: rebuild the WKB (little endian)
            parts = [struct.pack("<BII", 1, 3, len(self._rings))]
            for r in self._rings:
                parts.append(struct.pack("<I", r.shape[0]) + np.ascontiguousarray(r, dtype="<f8").tobytes())
            self.wkb = b"".join(parts)
        return (Polygon, (), self.wkb)

    @property
    def rings(self):
        if self._rings is None:
            self._rings = parse_wkb_polygon(self.wkb)
        return self._rings

    @property
    def exterior(self):
        return self.rings[0]

    @property
    def area(self):
        a = ring_area(self.rings[0])
        for hole in self.rings[1:]:
            a -= ring_area(hole)
        return a


def parse_wkb_polygon(wkb: bytes):
    bo = "<" if wkb[0] == 1 else ">"
    gtype, nrings = struct.unpack(bo + "II", wkb[1:9])
    if gtype != 3:
        raise ValueError(f"WKB geometry type {gtype}: only 2-D polygons are stored in tile graphs")
    off, rings = 9, []
    for _ in range(nrings):
        (npts,) = struct.unpack(bo + "I", wkb[off:off + 4])
        off += 4
        rings.append(np.frombuffer(wkb, dtype=bo + "f8", count=2 * npts, offset=off).reshape(npts, 2).astype(np.float64))
        off += 16 * npts
    return rings


def ring_area(p: np.ndarray) -> float:
    """|signed area| of a closed ring, in the operation order of GEOS: sum (x_i - x_0)(y_{i-1} - y_{i+1}) / 2, left to right."""
    s, x0 = 0.0, float(p[0, 0])
    for i in range(1, p.shape[0] - 1):
        s += (float(p[i, 0]) - x0) * (float(p[i - 1, 1]) - float(p[i + 1, 1]))
    return abs(s / 2.0)


class Tile:
    """tiling/tile.py:7-10 (data) and :37-39 (`area`)."""

    def __init__(self, tile_poly=None, id: int = 0):
        self.tile_poly = tile_poly
        self.id = id

    def area(self):
        return self.tile_poly.area

    def get_edge_num(self):                              # tile.py:33-35
        return self.tile_poly.exterior.shape[0] - 1

    def get_edge(self, edge_idx):                        # tile.py:24-27
        ring = self.tile_poly.exterior
        return np.array(ring[edge_idx]), np.array(ring[edge_idx + 1])

    def get_edge_length(self, edge_idx):                 # tile.py:29-31 over util/algo_util.py:72-74 (_distance)
        p0, p1 = self.get_edge(edge_idx)
        return np.sqrt(np.square(p0[0] - p1[0]) + np.square(p0[1] - p1[1]))

    def get_perimeter(self):                             # tile.py:41-42
        return np.sum([self.get_edge_length(i) for i in range(self.get_edge_num())])


Polygon.__module__ = "shapely.geometry.polygon"          # pickles written from here name the reference's classes
Tile.__module__ = "tiling.tile"

_ALLOWED = {
    ("tiling.tile", "Tile"): Tile,
    ("shapely.geometry.polygon", "Polygon"): Polygon,
    ("collections", "defaultdict"): defaultdict,
    ("builtins", "list"): list,
    ("builtins", "int"): int,
    ("numpy", "dtype"): np.dtype,
    ("numpy", "ndarray"): np.ndarray,
}
_NUMPY_RECONSTRUCTORS = ("scalar", "_reconstruct")


class SchemaUnpickler(pickle.Unpickler):
    """Resolves the globals of the reference's two pickle schemas and nothing else."""

    def find_class(self, module, name):
        hit = _ALLOWED.get((module, name))
        if hit is not None:
            return hit
        if module in ("numpy.core.multiarray", "numpy._core.multiarray") and name in _NUMPY_RECONSTRUCTORS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"global {module}.{name} is not part of the tile-graph / brick-layout schema")


def load_schema_pickle(path):
    with open(path, "rb") as f:
        return SchemaUnpickler(io.BytesIO(f.read())).load()


class _reference_names:
    """While dumping, `tiling.tile` / `shapely.geometry.polygon` resolve to the two classes above (pickle stores classes
    by module + name and checks that the name resolves back); whatever was registered under those names is restored."""
    _MODS = ("tiling", "tiling.tile", "shapely", "shapely.geometry", "shapely.geometry.polygon")

    def __enter__(self):
        import sys
        import types
        self.saved = {m: sys.modules.get(m) for m in self._MODS}
        for m in self._MODS:
            mod = types.ModuleType(m)
            mod.__path__ = []
            sys.modules[m] = mod
        sys.modules["tiling.tile"].Tile = Tile
        sys.modules["shapely.geometry.polygon"].Polygon = Polygon

    def __exit__(self, *exc):
        import sys
        for m, old in self.saved.items():
            if old is None:
                sys.modules.pop(m, None)
            else:
                sys.modules[m] = old


def dump_schema_pickle(obj, path, protocol=3):
    """pickle.dump under the reference's class names, so that the reference (shapely 1.x) reads the file back."""
    with _reference_names():
        data = pickle.dumps(obj, protocol=protocol)
    with open(path, "wb") as f:
        f.write(data)


_KEYS = ("tiles", "graph", "edges_features", "colli_edges", "adj_edges", "unique_adj_features", "max_area",
         "max_align_length", "align_start_index")


def _view(name):
    """A list/dict attribute of the reference's class; after a side-car load it is rebuilt from the arrays on first use."""
    def get(self):
        if self.__dict__.get("_v_" + name) is None and self.__dict__.get("_views_pending"):
            self._build_views()
        return self.__dict__.get("_v_" + name)

    def put(self, value):
        self.__dict__["_v_" + name] = value
    return property(get, put)


class TileGraph:
    tiles, graph, edges_features = _view("tiles"), _view("graph"), _view("edges_features")
    colli_edges, adj_edges = _view("colli_edges"), _view("adj_edges")

    def __init__(self, tile_type_count: int, tiles=None, one_hot=True, proto_tiles=None):
        if tiles is not None or proto_tiles is not None:
            raise NotImplementedError(
                "forming a complete graph needs polygon intersections (tile_graph.py:139-190): build it with the "
                "reference, save_current_state(), and load_graph_state() it here")
        self.tile_type_count = tile_type_count           # tile_graph.py:66
        self.one_hot = one_hot
        self.tiles = None
        self.graph = defaultdict(list)
        self.edges_features = defaultdict(list)
        self.adj_edges, self.colli_edges = [], []
        self.align_start_index = 2                       # :78
        self.max_align_length = 1e-10                    # :79
        self.unique_adj_features = None
        self.max_area = None
        self._arrays = None
        self._views_pending = False

    # ------------------------------------------------------------------ the reference's file
    def load_graph_state(self, path, sidecar: bool = True):
        """tile_graph.py:311-332.  With `sidecar`, `<path>.tgnn.npz` is used when it matches the pickle (size, mtime)
        and written after a pickle load when the directory is writable."""
        side = path + SIDECAR_SUFFIX
        if sidecar and os.path.exists(side) and self._load_sidecar(side, path):
            return
        temp = load_schema_pickle(path)
        missing = [k for k in _KEYS if k not in temp]
        if missing:
            raise AssertionError(f"{path}: not a complete-graph file, missing {missing}")          # :313-323
        self.tiles = temp["tiles"]
        self.graph = temp["graph"]
        self.edges_features = temp["edges_features"]
        self.colli_edges = temp["colli_edges"]
        self.adj_edges = temp["adj_edges"]
        self.unique_adj_features = temp["unique_adj_features"]
        self.max_area = temp["max_area"]
        self.align_start_index = temp["align_start_index"]
        self.max_align_length = temp["max_align_length"]
        self.total_feature_dim = self.align_start_index + len(self.unique_adj_features)             # :332
        self._arrays = None
        if sidecar:
            try:
                self.save_sidecar(side, path)
            except OSError:
                pass                                     # read-only data directory: the pickle stays the only copy

    def save_current_state(self, path):
        """tile_graph.py:296-309: same nine keys; the two classes pickle under the reference's names."""
        dump_schema_pickle({k: getattr(self, k) for k in _KEYS}, path)

    def _get_graph_statistics(self):                     # :334-338
        return len(self.tiles), len(self.adj_edges) // 2, len(self.colli_edges) // 2

    # ------------------------------------------------------------------ array forms
    @property
    def arrays(self) -> "GraphArrays":
        if self._arrays is None:
            self._arrays = GraphArrays.from_graph(self)
        return self._arrays

    def save_sidecar(self, side_path, pickle_path):
        st = os.stat(pickle_path)
        a = self.arrays
        ring_ptr = np.zeros(len(self.tiles) + 1, dtype=np.int64)
        rings = [t.tile_poly.exterior for t in self.tiles]
        ring_ptr[1:] = np.cumsum([r.shape[0] for r in rings])
        tmp = side_path + f".{os.getpid()}.tmp.npz"
        np.savez(
            tmp, version=np.int64(SIDECAR_VERSION), src_size=np.int64(st.st_size), src_mtime_ns=np.int64(st.st_mtime_ns),
            tile_type_count=np.int64(self.tile_type_count), tile_ids=a.tile_ids, tile_areas=a.tile_areas,
            ring_ptr=ring_ptr, ring_xy=np.concatenate(rings) if rings else np.zeros((0, 2)),
            colli_edges=a.colli_edges, adj_edges=a.adj_edges, colli_features=a.colli_features,
            adj_features=a.adj_features, adj_type=a.adj_type,
            unique_adj_features=np.asarray(self.unique_adj_features, dtype=np.float64),
            max_area=np.float64(self.max_area), max_align_length=np.float64(self.max_align_length),
            align_start_index=np.int64(self.align_start_index))
        os.replace(tmp, side_path)

    def _load_sidecar(self, side_path, pickle_path) -> bool:
        try:
            z = np.load(side_path)
            st = os.stat(pickle_path)
            if int(z["version"]) != SIDECAR_VERSION or int(z["src_size"]) != st.st_size or \
                    int(z["src_mtime_ns"]) != st.st_mtime_ns:
                return False
            arr = GraphArrays(z["tile_ids"], z["tile_areas"], z["colli_edges"], z["adj_edges"], z["colli_features"],
                              z["adj_features"], z["adj_type"], float(z["max_area"]), float(z["max_align_length"]),
                              int(self.tile_type_count))
            rings = (z["ring_ptr"], z["ring_xy"])
            unique = z["unique_adj_features"]
            scalars = (float(z["max_area"]), np.float64(z["max_align_length"]), int(z["align_start_index"]))
        except (OSError, KeyError, ValueError):
            return False
        self.unique_adj_features = [list(r) for r in unique]
        self.max_area, self.max_align_length, self.align_start_index = scalars
        self.total_feature_dim = self.align_start_index + len(self.unique_adj_features)
        self._arrays, self._rings = arr, rings
        for name in ("tiles", "graph", "edges_features", "colli_edges", "adj_edges"):
            self.__dict__["_v_" + name] = None
        self._views_pending = True
        return True

    def _build_views(self):
        """The reference's list / dict attributes from the arrays (same contents, same order)."""
        self._views_pending = False
        arr = self._arrays
        ring_ptr, ring_xy = self._rings
        tiles = []
        for i in range(arr.n_tiles):
            p = Polygon()
            p._rings = [ring_xy[ring_ptr[i]:ring_ptr[i + 1]]]
            tiles.append(Tile(p, int(arr.tile_ids[i])))
        self.tiles = tiles
        self.colli_edges = [(int(u), int(v)) for u, v in arr.colli_edges.T]
        self.adj_edges = [(int(u), int(v)) for u, v in arr.adj_edges.T]
        ef, graph = defaultdict(list), defaultdict(list)
        for edges, feats in ((arr.colli_edges, arr.colli_features), (arr.adj_edges, arr.adj_features)):
            for (u, v), row in zip(edges.T.tolist(), feats.tolist()):
                if u not in ef:
                    ef[u] = defaultdict(list)
                ef[u][v] = row
        both = np.concatenate([arr.colli_edges, arr.adj_edges], axis=1)
        order = np.lexsort((both[1], both[0]))
        for u, v in zip(both[0][order].tolist(), both[1][order].tolist()):
            graph[u].append(v)
        self.edges_features, self.graph = ef, graph


class GraphArrays:
    """The complete graph as arrays.  Edge order = the order of `colli_edges` / `adj_edges` in the file, which is the
    order every layout's edges inherit (tile_factory.py:42-45 filters, never reorders)."""

    def __init__(self, tile_ids, tile_areas, colli_edges, adj_edges, colli_features, adj_features, adj_type,
                 max_area, max_align_length, tile_type_count):
        self.tile_ids, self.tile_areas = tile_ids, tile_areas
        self.colli_edges, self.adj_edges = colli_edges, adj_edges
        self.colli_features, self.adj_features = colli_features, adj_features
        self.adj_type = adj_type
        self.max_area, self.max_align_length, self.tile_type_count = max_area, max_align_length, tile_type_count
        n = tile_ids.shape[0]
        self.n_tiles = n
        # (u, v) -> row of the edge's feature vector, for look-ups by edge (data_util.py:166-167)
        self._keys = [None, None]
        for k, e in enumerate((colli_edges, adj_edges)):
            key = e[0] * n + e[1]
            order = np.argsort(key, kind="stable")
            self._keys[k] = (key[order], order)

    @staticmethod
    def from_graph(g: TileGraph) -> "GraphArrays":
        ids = np.array([t.id for t in g.tiles], dtype=np.int64)
        areas = np.array([t.area() for t in g.tiles], dtype=np.float64)
        ef = g.edges_features
        col = np.array(g.colli_edges, dtype=np.int64).reshape(-1, 2).T
        adj = np.array(g.adj_edges, dtype=np.int64).reshape(-1, 2).T
        f = g.align_start_index + len(g.unique_adj_features)
        colf = np.array([ef[u][v] for u, v in g.colli_edges], dtype=np.float64).reshape(-1, f)
        adjf = np.array([ef[u][v] for u, v in g.adj_edges], dtype=np.float64).reshape(-1, f)
        onehot = adjf[:, g.align_start_index:]
        adj_type = np.argmax(onehot, axis=1).astype(np.int32) if adjf.shape[0] else np.zeros(0, dtype=np.int32)
        if adjf.shape[0] and not (np.all(onehot.sum(axis=1) == 1) and np.all(onehot.max(axis=1) == 1)):
            raise ValueError("adjacency feature rows are not one-hot past align_start_index (tile_graph.py:262-276)")
        return GraphArrays(ids, areas, np.ascontiguousarray(col), np.ascontiguousarray(adj), colf, adjf, adj_type,
                           float(g.max_area), float(g.max_align_length), int(g.tile_type_count))

    @staticmethod
    def perimeters(tiles) -> np.ndarray:
        """`Tile.get_perimeter()` (tile.py:41-42) of every tile, float64 -- the same edge lengths added in the same order
        (np.sum over a short list is a left-to-right pairwise-free sum for < 8 elements; longer rings go through np.sum
        as in the reference)."""
        return np.array([t.get_perimeter() for t in tiles], dtype=np.float64)

    def edge_rows(self, which: int, edges: np.ndarray) -> np.ndarray:
        """Rows (into colli_* if which == 0 else adj_*) of the given [2, E] complete-graph edges."""
        keys, order = self._keys[which]
        q = edges[0] * self.n_tiles + edges[1]
        pos = np.searchsorted(keys, q)
        if q.size and (np.any(pos >= keys.shape[0]) or np.any(keys[np.minimum(pos, keys.shape[0] - 1)] != q)):
            raise KeyError("edge not in the complete graph")
        return order[pos]

    def node_features(self) -> np.ndarray:
        """data_util.py:185-189 for ALL tiles: one-hot tile type, then area / max_area."""
        x = np.zeros((self.n_tiles, self.tile_type_count + 1))
        x[np.arange(self.n_tiles), self.tile_ids] = 1
        x[:, -1] = self.tile_areas / self.max_area
        return x

    def adj_features_normalised(self) -> np.ndarray:
        """data_util.py:168-169: column 1 (align length) divided by max_align_length, in float64."""
        f = self.adj_features.copy()
        if f.shape[0]:
            f[:, 1] = f[:, 1] / self.max_align_length
        return f
