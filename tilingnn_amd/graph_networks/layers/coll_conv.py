"""`CollConv` (collision branch layer) and the `GINConv` it wraps.

Mirror of /root/reference/graph_networks/layers/coll_conv.py:6-30: GINConv(nn=MLP(C->32->64->C,
Sigmoid x3, no BN)) with `eps` a BUFFER [1] (train_eps=False), sum aggregation (the ctor's
aggr='mean' argument is accepted and ignored exactly as the reference ignores it, coll_conv.py:10,18),
self loops removed, then LeakyReLU and BatchNorm1d.  Forward = csrc/gin.hip."""
import torch
import torch.nn as nn

from ... import ops
from .. import _graph_cache
from .._tracking import BatchNorm1d, Tracked
from .util import MLP


class GINConv(Tracked, nn.Module):
    """Drop-in for torch_geometric.nn.GINConv as the reference uses it."""

    def __init__(self, nn_module=None, eps=0.0, train_eps=False, **kwargs):
        super().__init__()
        nn_module = kwargs.pop("nn", nn_module)               # PyG's keyword is `nn=`
        if train_eps:
            raise ValueError("train_eps=True is not used by the reference and not implemented")
        self.nn = nn_module
        self.register_buffer("eps", torch.Tensor([eps]))

    def _mlp_params(self):
        layers = list(self.nn.mlp)
        if len(layers) != 3 or any(l.batch_norm is not None for l in layers) or \
                any(not isinstance(l.activation, nn.Sigmoid) for l in layers):
            raise ValueError("the GIN kernels expect CollConv's MLP: 3 x (Linear, Sigmoid), no BN")
        out = []
        for l in layers:
            out += [l.linear.weight, l.linear.bias]
        return out

    def forward_fused(self, x, edge_index, act, partials=None):
        graph = _graph_cache.get_col(int(x.shape[0]), edge_index)
        return ops.gin(x, graph, self.eps, *self._mlp_params(), act=act, partials=partials)

    def forward(self, x, edge_index):
        return self.forward_fused(x, edge_index, ops.ACT_NONE)[0]


class CollConv(Tracked, nn.Module):
    def __init__(self, node_feature_in_dim, node_feature_out_dim, hidden_dims=[32, 64], aggr="mean",
                 batch_norm=True, mlp_activation=torch.nn.Sigmoid(), final_activation=torch.nn.LeakyReLU()):
        super().__init__()
        mlp = MLP(in_dim=node_feature_in_dim, out_dim=node_feature_out_dim, hidden_layer_dims=list(hidden_dims),
                  activation=mlp_activation, batch_norm=False)
        self.ginConv = GINConv(nn=mlp)
        self.activation = final_activation
        self.out_dim = node_feature_out_dim
        self.batch_norm = BatchNorm1d(node_feature_out_dim) if batch_norm else None

    def forward(self, x, edge_index):
        act = ops.act_code(self.activation)
        if act == ops.ACT_SIGMOID:
            raise ValueError("CollConv kernels fuse None / LeakyReLU only")
        parts = ops.new_partials(self.out_dim, x.device) if self.batch_norm is not None else None
        out, n_parts = self.ginConv.forward_fused(x, edge_index, act, parts)
        if self.batch_norm is not None:
            out = ops.batch_norm(out, parts, n_parts, self.batch_norm)
        return out, edge_index
