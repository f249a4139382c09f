"""`Losses.calculate_unsupervised_loss` -- the loss ML_Solver.predict evaluates to pick the best probability map
(/root/reference/solver/ml_solver/losses.py:48-116, called through get_best_prob_map, ml_solver.py:46,133-136),
on the GPU through `tgnn_unsupervised_loss` (csrc/loss.hip).  Same arguments and the same three return values as the
reference: (min loss as a 0-dim tensor, arg-min as numpy, all losses as numpy).  When `probs` requires grad (the
training step, trainer.py:76-80) the returned loss is differentiable: its backward is `tgnn_unsupervised_loss_bwd`
(csrc/backward.hip) on the arg-min map -- the path torch.min's gradient takes in the reference (losses.py:108)."""
import ctypes as C
import math

import numpy as np
import torch

from ... import _lib, ops
from ..._lib import check, lib, ptr


def loss_weights():
    """(COLLISION_WEIGHT, ALIGN_LENGTH_WEIGHT, AVG_AREA_WEIGHT): the live values of `inputs.config` when the package is
    used inside the reference tree, else the reference's defaults (inputs/config.py:49-51)."""
    try:
        import inputs.config as config                      # noqa: the reference's configuration module
        return float(config.COLLISION_WEIGHT), float(config.ALIGN_LENGTH_WEIGHT), float(config.AVG_AREA_WEIGHT)
    except Exception:
        return 1.0 / math.log(1.0 + 1e-1), 0.02, 1.0


class Losses:
    @staticmethod
    def unsupervised_losses(probs, node_feature, collide_edge_index, adj_edges_index, adj_edge_features, weights=None):
        """-> (losses [M] float64 on the device, terms [M, 3] float64: the three logarithmic terms)."""
        if not probs.is_cuda:
            raise RuntimeError("tilingnn_amd has no CPU path: the loss runs on the GPU the probabilities live on")
        p = ops._f32c(probs, "probs")
        x = ops._f32c(node_feature, "node_feature")
        if p.dim() != 2 or x.dim() != 2 or x.shape[0] != p.shape[0]:
            raise ValueError(f"probs must be [N, M] and node_feature [N, Fx], got {tuple(p.shape)} / {tuple(x.shape)}")
        n, m = int(p.shape[0]), int(p.shape[1])
        # losses.py:54-55: `len(index) > 0` -- an empty edge set switches its term off
        e_col = int(collide_edge_index.shape[1]) if collide_edge_index.numel() > 0 else 0
        e_adj = int(adj_edges_index.shape[1]) if adj_edges_index.numel() > 0 else 0
        col = ops._check_edge_index(collide_edge_index, "collide_edge_index") if e_col else None
        adj = ops._check_edge_index(adj_edges_index, "adj_edges_index") if e_adj else None
        attr = ops._f32c(adj_edge_features, "adj_edge_features") if e_adj else None
        if e_adj and (attr.dim() != 2 or attr.shape[0] != e_adj or attr.shape[1] < 2):
            raise ValueError(f"adj_edge_features must be [Ea, Fe >= 2], got {tuple(attr.shape)}")
        wc, wl, wa = weights if weights is not None else loss_weights()
        dev = p.device
        losses = torch.empty(m, dtype=torch.float64, device=dev)
        terms = torch.empty(m, 3, dtype=torch.float64, device=dev)
        ws_bytes = lib.tgnn_unsupervised_loss_workspace_bytes(m)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        fx = int(x.shape[1])
        area_ptr = C.c_void_p(x.data_ptr() + 4 * (fx - 1))                       # node_feature[:, -1]
        len_ptr = C.c_void_p(attr.data_ptr() + 4) if e_adj else None             # adj_edge_features[:, 1]
        check(lib.tgnn_unsupervised_loss(ptr(p), m, m, area_ptr, fx, n, ptr(col) if e_col else None, e_col,
                                         ptr(adj) if e_adj else None, e_adj, len_ptr, int(attr.shape[1]) if e_adj else 1,
                                         wc, wl, wa, ptr(losses), ptr(terms), ptr(ws), ws_bytes, _lib.current_stream(dev)))
        return losses, terms

    @staticmethod
    def calculate_unsupervised_loss(probs, node_feature, collide_edge_index, adj_edges_index, adj_edge_features):
        if torch.is_grad_enabled() and probs.requires_grad:
            return _differentiable_loss(probs, node_feature, collide_edge_index, adj_edges_index, adj_edge_features)
        return Losses._calculate(probs, node_feature, collide_edge_index, adj_edges_index, adj_edge_features)[:3]

    @staticmethod
    def _calculate(probs, node_feature, collide_edge_index, adj_edges_index, adj_edge_features):
        losses, terms = Losses.unsupervised_losses(probs, node_feature, collide_edge_index, adj_edges_index,
                                                   adj_edge_features)
        host_terms = terms.cpu().numpy()
        # the reference asserts these signs (losses.py:100-102,108)
        assert (host_terms <= 0).all(), "loss terms must be non-positive"
        host = losses.cpu().numpy()
        assert (host >= 1.0).all()
        min_index = np.argmin(host)
        return losses[int(min_index)].to(probs.dtype), np.asarray(min_index), host.astype(np.float32), terms


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs, node_feature, collide_edge_index, adj_edges_index, adj_edge_features, box):
        loss, min_index, host, terms = Losses._calculate(probs.detach(), node_feature, collide_edge_index, adj_edges_index,
                                                         adj_edge_features)
        box.extend([min_index, host])
        ctx.k, ctx.terms, ctx.weights = int(min_index), terms, loss_weights()
        ctx.save_for_backward(probs, node_feature, collide_edge_index, adj_edges_index, adj_edge_features)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        probs, x, col, adj, attr = ctx.saved_tensors
        p = ops._f32c(probs.detach(), "probs")
        xf = ops._f32c(x, "node_feature")
        n, m, fx = int(p.shape[0]), int(p.shape[1]), int(xf.shape[1])
        e_col = int(col.shape[1]) if col.numel() > 0 else 0
        e_adj = int(adj.shape[1]) if adj.numel() > 0 else 0
        colc = ops._check_edge_index(col, "collide_edge_index") if e_col else None
        adjc = ops._check_edge_index(adj, "adj_edges_index") if e_adj else None
        attrc = ops._f32c(attr, "adj_edge_features") if e_adj else None
        dev = p.device
        dprobs = torch.zeros(n, m, dtype=torch.float32, device=dev)
        ws = torch.empty(n, dtype=torch.float64, device=dev)
        gout = grad_loss.detach().to(torch.float32).reshape(1).contiguous()
        wc, wl, wa = ctx.weights
        k = ctx.k
        check(lib.tgnn_unsupervised_loss_bwd(
            C.c_void_p(p.data_ptr() + 4 * k), m, C.c_void_p(xf.data_ptr() + 4 * (fx - 1)), fx, n,
            ptr(colc) if e_col else None, e_col, ptr(adjc) if e_adj else None, e_adj,
            C.c_void_p(attrc.data_ptr() + 4) if e_adj else None, int(attrc.shape[1]) if e_adj else 1, wc, wl, wa,
            C.c_void_p(ctx.terms.data_ptr() + 24 * k), ptr(gout), C.c_void_p(dprobs.data_ptr() + 4 * k), m, ptr(ws),
            n * 8, _lib.current_stream(dev)))
        return dprobs.to(probs.dtype), None, None, None, None, None


def _differentiable_loss(probs, node_feature, collide_edge_index, adj_edges_index, adj_edge_features):
    box = []
    loss = _LossFn.apply(probs, node_feature, collide_edge_index, adj_edges_index, adj_edge_features, box)
    return loss, box[0], box[1]
