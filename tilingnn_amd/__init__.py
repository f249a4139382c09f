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
import os as _os

__version__ = "0.1.0"
# HIP hands its hardware queues to streams round robin, 4 by default; the forward uses a side stream, forward_many three more:
# with 8 queues they do not share one (a shared queue runs its streams' kernels one after the other).  Read when the HIP
# runtime starts, i.e. effective if this import comes before the process's first GPU call; set it yourself otherwise.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
