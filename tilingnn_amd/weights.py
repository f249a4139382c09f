"""State-dict layout of TilinGNN and a seeded, torch-RNG-independent weight recipe.

The key layout is the one the reference's modules produce (664 entries at depth 20):
TilinGNN.__init__ (/root/reference/graph_networks/networks/TilinGNN.py:14-48),
GraphConv (layers/edge_conv.py:8-22, the edge MLP is registered twice: '<p>.mlp.*' and
'<p>.nnConv.nn.*'), CollConv (layers/coll_conv.py:8-22), MLP / Linear_trans
(layers/util.py:4-29); NNConv.root is [in, out] and GINConv.eps a buffer [1] (PyG 1.3.2).

The trained checkpoints (pre-trained_models/*.pth) are not in the reference checkout, so
tests and the benchmark use `make_state_dict`: every tensor is drawn from a numpy
Generator seeded by (seed, crc32(key)), i.e. independent of torch's RNG stream, of the
order in which keys are generated and of the device.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

EDGE_MLP_HIDDEN = (32, 64)       # edge_conv.py:9 / coll_conv.py:9 hidden_dims
FINAL_MLP_HIDDEN = (256, 128, 64)  # TilinGNN.py:46


def _linear_trans(spec, prefix, fin, fout, bn):
    spec[f"{prefix}.linear.weight"] = (fout, fin)
    spec[f"{prefix}.linear.bias"] = (fout,)
    if bn:
        _bn(spec, f"{prefix}.batch_norm", fout)


def _bn(spec, prefix, f):
    spec[f"{prefix}.weight"] = (f,)
    spec[f"{prefix}.bias"] = (f,)
    spec[f"{prefix}.running_mean"] = (f,)
    spec[f"{prefix}.running_var"] = (f,)
    spec[f"{prefix}.num_batches_tracked"] = ()


def _mlp(spec, prefix, dims, bn):
    for i in range(len(dims) - 1):
        _linear_trans(spec, f"{prefix}.mlp.{i}", dims[i], dims[i + 1], bn)


def state_dict_spec(adj_edge_features_dim: int, network_depth: int, network_width: int,
                    output_dim: int = 1, node_features_dim: int = 3) -> "OrderedDict[str, Tuple[int, ...]]":
    """{key: shape} in the registration order of the reference modules."""
    c, fe, fx = network_width, adj_edge_features_dim, node_features_dim
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    _mlp(spec, "init_node_feature_trans", [fx, c, c], bn=True)
    for i in range(network_depth):
        p = f"brch_1_graph_conv_layers.{i}"
        dims = [fe, *EDGE_MLP_HIDDEN, c * c]
        _mlp(spec, f"{p}.mlp", dims, bn=False)
        spec[f"{p}.nnConv.root"] = (c, c)
        spec[f"{p}.nnConv.bias"] = (c,)
        _mlp(spec, f"{p}.nnConv.nn", dims, bn=False)          # alias of <p>.mlp (same tensors)
        _bn(spec, f"{p}.batch_norm", c)
    for i in range(network_depth):
        p = f"brch_2_coll_conv_layers.{i}"
        spec[f"{p}.ginConv.eps"] = (1,)
        _mlp(spec, f"{p}.ginConv.nn", [c, *EDGE_MLP_HIDDEN, c], bn=False)
        _bn(spec, f"{p}.batch_norm", c)
    _mlp(spec, "final_mlp.0", [c * (network_depth + 1), *FINAL_MLP_HIDDEN, c], bn=True)
    _linear_trans(spec, "final_mlp.1", c, output_dim, bn=False)
    return spec


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(key.encode())])


def make_state_dict(adj_edge_features_dim: int, network_depth: int, network_width: int,
                    output_dim: int = 1, node_features_dim: int = 3, seed: int = 0,
                    dtype: torch.dtype = torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """Seeded random-init weights, loadable with `load_state_dict(strict=True)` into both the
    reference TilinGNN and `tilingnn_amd.TilinGNN`.

    Linear weight/bias ~ U(+-1/sqrt(fan_in)) (torch's nn.Linear scale); NNConv root/bias
    ~ U(+-1/sqrt(in)) (PyG's `uniform(size, tensor)`); BatchNorm gamma ~ U(0.8, 1.2) and
    beta ~ U(-0.1, 0.1) (not the 1 / 0 defaults, so that a swapped gamma/beta or a dropped
    affine term is visible); running_mean 0, running_var 1, num_batches_tracked 0; eps 0.
    """
    spec = state_dict_spec(adj_edge_features_dim, network_depth, network_width, output_dim, node_features_dim)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, shape in spec.items():
        if ".nnConv.nn." in key:                              # alias: same values as <p>.mlp.*
            sd[key] = sd[key.replace(".nnConv.nn.", ".mlp.")]
            continue
        g = _rng(seed, key)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            sd[key] = torch.zeros((), dtype=torch.int64)
            continue
        if leaf == "eps":
            arr = np.zeros(shape)
        elif leaf == "running_mean":
            arr = np.zeros(shape)
        elif leaf == "running_var":
            arr = np.ones(shape)
        elif ".batch_norm." in key:
            arr = g.uniform(0.8, 1.2, shape) if leaf == "weight" else g.uniform(-0.1, 0.1, shape)
        elif leaf == "root":
            arr = g.uniform(-1.0, 1.0, shape) / np.sqrt(shape[0])
        elif key.endswith(".nnConv.bias"):
            arr = g.uniform(-1.0, 1.0, shape) / np.sqrt(network_width)
        elif leaf == "weight":
            arr = g.uniform(-1.0, 1.0, shape) / np.sqrt(shape[1])
        else:  # linear bias: fan_in of the matching weight
            fan_in = spec[key[: -len("bias")] + "weight"][1]
            arr = g.uniform(-1.0, 1.0, shape) / np.sqrt(fan_in)
        sd[key] = torch.from_numpy(np.ascontiguousarray(arr)).to(dtype)
    return sd


def infer_dims(sd: Dict[str, torch.Tensor]) -> Dict[str, int]:
    """Recover the ctor arguments from a checkpoint (Fe and Fx are not stored explicitly)."""
    depth = 0
    while f"brch_1_graph_conv_layers.{depth}.nnConv.bias" in sd:
        depth += 1
    width = int(sd["init_node_feature_trans.mlp.0.linear.weight"].shape[0])
    return {
        "adj_edge_features_dim": int(sd["brch_1_graph_conv_layers.0.mlp.mlp.0.linear.weight"].shape[1]),
        "network_depth": depth,
        "network_width": width,
        "output_dim": int(sd["final_mlp.1.linear.weight"].shape[0]),
        "node_features_dim": int(sd["init_node_feature_trans.mlp.0.linear.weight"].shape[1]),
    }
