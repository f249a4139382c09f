"""Golden vectors for the greedy assembly loop, produced by the REFERENCE itself.

Run in the build container only (needs /root/reference):   python tests/golden/generate_greedy_golden.py

Imported UNCHANGED: `tiling/brick_layout.py` (BrickLayout.compute_sub_layout, :248-286) and `util/algorithms.py`
(SelectionSolution :282-294, solve_by_probablistic_greedy :18-62, label_collision_neighbor :196-207).  What the image
lacks is stood in for at import time only: shapely / PyQt5 / torch_geometric become attribute-absorbing dummies (none
of their arithmetic is on this path: the loop only `.union()`s and `.buffer()`s polygons it never reads), and
`inputs.config` is the stub of generate_golden.py.  The network is replaced by a deterministic fake predictor
(`fake_probs`, a pure function of the sub-layout arrays) so that the loop's own logic -- re-indexing, the
geometric-mean probability update, the descending sweep with its `exp(p - 1) > U` acceptance on numpy's global
RNG stream, collision labelling, the early break -- is what gets pinned.  `create_solution` scores through
shapely areas; the layout handed in answers those calls with constants, the score is not recorded.

Stored (ref_greedy.npz): for the real labyrinth graph
  * compute_sub_layout for three seeded random label sets: the five arrays + the inverse index,
  * solve_by_probablistic_greedy for two RNG seeds: selection, selection order, rounds, sub-layout sizes per round.
"""
import math
import os
import sys
import types

import numpy as np
import torch  # noqa: F401  (before the reference tree goes on sys.path: its `util` package must not shadow torch's imports)

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import generate_golden as gg                                   # noqa: E402  (load_labyrinth only)


class _Any:
    """Absorbs every attribute access / call; polygons only need union() and buffer() here."""
    def __init__(self, *a, **k): pass
    def __call__(self, *a, **k): return _Any()
    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Any()
    def __iter__(self): return iter(())
    def __mro_entries__(self, bases): return (object,)
    def union(self, o): return self
    def buffer(self, *a, **k): return self


def import_reference():
    def stub(name):
        m = types.ModuleType(name); m.__path__ = []
        m.__getattr__ = lambda n: _Any()
        sys.modules[name] = m
    for n in ["shapely", "shapely.geometry", "shapely.ops", "shapely.affinity", "shapely.geometry.polygon",
              "shapely.geometry.multipolygon", "shapely.wkt", "shapely.validation", "PyQt5", "PyQt5.QtWidgets",
              "PyQt5.QtCore", "PyQt5.QtGui", "torch_geometric", "torch_geometric.data", "torch_geometric.nn"]:
        stub(n)
    inputs = types.ModuleType("inputs"); inputs.__path__ = []
    cfg = types.ModuleType("inputs.config")
    cfg.environment = types.SimpleNamespace(tile_count=2)
    cfg.COLLISION_WEIGHT, cfg.ALIGN_LENGTH_WEIGHT, cfg.AVG_AREA_WEIGHT = 1 / math.log(1.1), 0.02, 1   # config.py:49-51
    cfg.debug_base_folder, cfg.experiment_id = "/tmp/_tgnn_golden_dbg", 0
    cfg.__getattr__ = lambda n: 0
    inputs.config = cfg
    sys.modules["inputs"], sys.modules["inputs.config"] = inputs, cfg
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import tiling.brick_layout as bl
    import util.algorithms as alg
    return bl, alg


def fake_probs(node_feature, align_edge_index, collide_edge_index):
    """Deterministic stand-in for the network: a pure function of the sub-layout (float64 numpy), in (0.05, 0.95)."""
    n = node_feature.shape[0]
    deg_a = np.bincount(align_edge_index[1], minlength=n) if align_edge_index.size else np.zeros(n)
    deg_c = np.bincount(collide_edge_index[1], minlength=n) if collide_edge_index.size else np.zeros(n)
    t = np.sin(12.9898 * node_feature[:, -1] + 78.233 * deg_a + 37.719 * deg_c + 0.37 * np.arange(n)) * 43758.5453
    return 0.05 + 0.9 * (t - np.floor(t))


class FakeSolver:
    def __init__(self):
        self.sizes = []

    def predict(self, layout):
        self.sizes.append((layout.node_feature.shape[0], int(np.asarray(layout.align_edge_index).shape[-1]) if
                           np.asarray(layout.align_edge_index).size else 0,
                           int(np.asarray(layout.collide_edge_index).shape[-1]) if np.asarray(layout.collide_edge_index).size else 0))
        if len(layout.collide_edge_index) == 0 or len(layout.align_edge_index) == 0:       # ml_solver.py:31-32
            return np.ones(layout.node_feature.shape[0])
        return fake_probs(layout.node_feature, layout.align_edge_index, layout.collide_edge_index)


def main():
    bl, alg = import_reference()
    g = gg.load_labyrinth()
    n = g["x"].shape[0]
    tile = types.SimpleNamespace(tile_poly=_Any(), get_perimeter=lambda: 1e9)
    cg = types.SimpleNamespace(tiles=[tile] * n, max_area=1.0, max_align_length=1.0)

    class Layout(bl.BrickLayout):                                   # answers create_solution's geometry calls
        def get_super_contour_poly(self):
            return types.SimpleNamespace(area=1e9)

    layout = Layout(cg, g["x"], g["col"], g["col_attr"], g["adj"], g["adj_attr"], {i: i for i in range(n)})
    out = {}
    rng = np.random.default_rng(5)
    for k, frac in enumerate((0.1, 0.5, 0.93)):
        labelled = np.sort(rng.choice(n, size=int(frac * n), replace=False))
        sol = alg.SelectionSolution(n)
        for v in labelled:
            sol.label_node(int(v), int(v) % 2, layout)
        sub, inv = layout.compute_sub_layout(sol)
        out[f"sub{k}.labelled"] = labelled
        out[f"sub{k}.x"] = sub.node_feature
        out[f"sub{k}.adj"] = np.asarray(sub.align_edge_index).reshape(2, -1).astype(np.int64)
        out[f"sub{k}.adj_attr"] = np.asarray(sub.align_edge_features).reshape(-1, g["adj_attr"].shape[1])
        out[f"sub{k}.col"] = np.asarray(sub.collide_edge_index).reshape(2, -1).astype(np.int64)
        out[f"sub{k}.col_attr"] = np.asarray(sub.collide_edge_features).reshape(-1, g["col_attr"].shape[1])
        out[f"sub{k}.inverse"] = np.array([inv[i] for i in range(len(inv))], dtype=np.int64)
    for seed in (0, 7):
        np.random.seed(seed)
        fake = FakeSolver()
        selection, _score, order = alg.solve_by_probablistic_greedy(fake, layout)
        out[f"greedy{seed}.selection"] = np.asarray(selection, dtype=np.int8)
        out[f"greedy{seed}.order"] = np.asarray(order, dtype=np.int64)
        out[f"greedy{seed}.sizes"] = np.asarray(fake.sizes, dtype=np.int64)
        print(f"seed {seed}: {len(fake.sizes)} rounds, {int(np.sum(selection))} tiles selected, first sizes {fake.sizes[:4]}")
    np.savez_compressed(os.path.join(HERE, "ref_greedy.npz"), **out)
    print(f"ref_greedy.npz: {os.path.getsize(os.path.join(HERE, 'ref_greedy.npz')) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
