"""`Linear_trans` and `MLP` with the reference's constructor signatures and parameter names
(/root/reference/graph_networks/layers/util.py:4-37), executed by the MFMA dense kernel.

Linear_trans = Linear -> activation -> BatchNorm1d (POST-activation BN, util.py:31-37);
MLP applies the activation on every layer including the last (util.py:10-13).  Inside an MLP the
BatchNorm of layer l is applied while layer l+1 stages its A tile (in_stat), so normalised
activations are only materialised for the MLP's final output."""
import torch
import torch.nn as nn

from ... import ops
from .._tracking import BatchNorm1d, Linear, Tracked


class Linear_trans(Tracked, nn.Module):
    def __init__(self, in_dim, out_dim, activation=None, batch_norm=True):
        super().__init__()
        self.linear = Linear(in_dim, out_dim)
        self.activation = activation
        self.batch_norm = BatchNorm1d(out_dim) if batch_norm else None

    def _dense(self, x, in_stat=None):
        """-> (pre-BN output, partials, n_partials)"""
        parts = ops.new_partials(self.linear.out_features, x.device) if self.batch_norm is not None else None
        out, n_parts = ops.dense_act(x, self.linear.weight, self.linear.bias, ops.act_code(self.activation),
                                     in_stat=in_stat, partials=parts)
        return out, parts, n_parts

    def forward(self, x):
        out, parts, n_parts = self._dense(x)
        if self.batch_norm is not None:
            out = ops.batch_norm(out, parts, n_parts, self.batch_norm)
        return out


class MLP(Tracked, nn.Module):
    def __init__(self, in_dim, out_dim, hidden_layer_dims: list, activation, batch_norm=True):
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim
        dims = [in_dim] + list(hidden_layer_dims) + [out_dim]
        self.mlp = nn.Sequential(*[Linear_trans(dims[i], dims[i + 1], activation=activation, batch_norm=batch_norm)
                                   for i in range(len(dims) - 1)])

    def forward(self, x):
        if x.shape[-1] != self.in_dim:                       # util.py:16 asserts
            raise ValueError(f"MLP expected in_dim {self.in_dim}, got {x.shape[-1]}")
        stat = None
        for layer in self.mlp:
            x, parts, n_parts = layer._dense(x, in_stat=stat)
            stat = None
            if layer.batch_norm is not None:
                bn = layer.batch_norm
                n = int(x.shape[0])
                if bn.training or not bn.track_running_stats:
                    if n < 2:
                        raise ValueError("Expected more than 1 value per channel when training")
                    stat = ops.bn_finalize(parts, n_parts, n, bn, update_running=True, mode=0)
                else:
                    stat = ops.bn_finalize(parts, max(n_parts, 1), max(n, 1), bn, update_running=False, mode=3)
        return ops.bn_apply(x, stat) if stat is not None else x
