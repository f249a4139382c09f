"""tilingnn_amd -- MI355X (gfx950) implementation of TilinGNN's graph-conv scoring forward.

Public surface = the reference's own for this path:
    tilingnn_amd.TilinGNN                        graph_networks/networks/TilinGNN.py
    tilingnn_amd.graph_networks.layers.*         GraphConv / CollConv / MLP / Linear_trans
    tilingnn_amd.get_network_prediction          graph_networks/network_utils.py
    tilingnn_amd.solver.ml_solver.ML_Solver      predict / get_predict_probs / load_saved_network
plus `tilingnn_amd.ops` (torch-tensor front end of the C ABI in include/tgnn.h),
`tilingnn_amd.weights` (state-dict layout + seeded recipe) and `tilingnn_amd.synth`
(seeded synthetic super-graphs).  Importing the GPU-facing parts requires the built
libtgnn.so; `weights` and `synth` are pure numpy/torch and import anywhere.
"""
__version__ = "0.1.0"

_GPU_ATTRS = {"TilinGNN": ("graph_networks.networks.TilinGNN", "TilinGNN"),
              "get_network_prediction": ("graph_networks.network_utils", "get_network_prediction"),
              "ops": ("ops", None)}


def __getattr__(name):          # lazy: `import tilingnn_amd.weights` must not need the shared library
    if name in _GPU_ATTRS:
        import importlib
        mod, attr = _GPU_ATTRS[name]
        m = importlib.import_module(f"{__name__}.{mod}")
        return m if attr is None else getattr(m, attr)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
