"""`GraphConv` (adjacency branch layer) and the `NNConv` it wraps.

Mirror of /root/reference/graph_networks/layers/edge_conv.py:6-30 (same ctor arguments, same
state-dict keys: the edge MLP is registered twice, as `mlp` and as `nnConv.nn`; `nnConv.root` is
[in, out], `nnConv.bias` [out] as in PyTorch-Geometric 1.3.2's NNConv).  The forward is the
LDS-resident-weight-table kernel of csrc/nnconv.hip instead of PyG's MessagePassing.propagate."""
import math

import torch
import torch.nn as nn

from ... import ops
from .. import _graph_cache
from .._tracking import BatchNorm1d, Tracked
from .util import MLP


class NNConv(Tracked, nn.Module):
    """Drop-in for torch_geometric.nn.conv.nn_conv.NNConv as the reference uses it
    (in == out == network_width, aggr='mean', root weight and bias on)."""

    def __init__(self, in_channels, out_channels, nn_module, aggr="add", root_weight=True, bias=True):
        super().__init__()
        if aggr != "mean":
            raise ValueError("tilingnn_amd.NNConv implements aggr='mean' (edge_conv.py:10,18) only")
        if not (root_weight and bias):
            raise ValueError("tilingnn_amd.NNConv needs root_weight=True and bias=True (the reference's defaults)")
        self.in_channels, self.out_channels, self.aggr = in_channels, out_channels, aggr
        self.nn = nn_module
        bound = 1.0 / math.sqrt(in_channels)                  # PyG's uniform(size=in_channels, tensor)
        self.root = nn.Parameter(torch.empty(in_channels, out_channels).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))

    def _edge_mlp_params(self):
        layers = list(self.nn.mlp)
        if len(layers) != 3 or any(l.batch_norm is not None for l in layers) or \
                any(not isinstance(l.activation, nn.Sigmoid) for l in layers):
            raise ValueError("the NNConv kernels expect GraphConv's edge MLP: 3 x (Linear, Sigmoid), no BN")
        out = []
        for l in layers:
            out += [l.linear.weight, l.linear.bias]
        return out

    def forward_fused(self, x, edge_index, edge_attr, act, partials=None):
        if self.in_channels != self.out_channels:
            raise ValueError("NNConv kernels need in_channels == out_channels")
        graph = _graph_cache.get_adj(int(x.shape[0]), edge_index, edge_attr)
        wtab = ops.edge_weight_table(edge_attr, graph, *self._edge_mlp_params(), self.in_channels)
        return ops.nnconv_mean(x, graph, wtab, self.root, self.bias, act=act, partials=partials)

    def forward(self, x, edge_index, edge_attr):
        return self.forward_fused(x, edge_index, edge_attr, ops.ACT_NONE)[0]


class GraphConv(Tracked, nn.Module):
    def __init__(self, edge_feature_dim, node_feature_in_dim, node_feature_out_dim, hidden_dims=[32, 64],
                 aggr="mean", batch_norm=True, mlp_activation=torch.nn.Sigmoid(),
                 final_activation=torch.nn.LeakyReLU()):
        super().__init__()
        self.mlp = MLP(in_dim=edge_feature_dim, out_dim=node_feature_in_dim * node_feature_out_dim,
                       hidden_layer_dims=list(hidden_dims), activation=mlp_activation, batch_norm=False)
        self.nnConv = NNConv(node_feature_in_dim, node_feature_out_dim, self.mlp, aggr=aggr)
        self.activation = final_activation
        self.batch_norm = BatchNorm1d(node_feature_out_dim) if batch_norm else None

    def forward(self, x, edge_index, edge_features):
        act = ops.act_code(self.activation)
        if act == ops.ACT_SIGMOID:
            raise ValueError("GraphConv kernels fuse None / LeakyReLU only")
        parts = ops.new_partials(self.nnConv.out_channels, x.device) if self.batch_norm is not None else None
        out, n_parts = self.nnConv.forward_fused(x, edge_index, edge_features, act, parts)
        if self.batch_norm is not None:
            out = ops.batch_norm(out, parts, n_parts, self.batch_norm)
        return out, edge_index, edge_features
