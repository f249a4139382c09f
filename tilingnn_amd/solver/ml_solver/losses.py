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
        if np.isnan(host).any():                                # the kernel's report of an edge end outside [0, N)
            raise IndexError(f"edge index out of range [0, {int(probs.shape[0])}) in collide_edge_index / adj_edges_index "
                             "(torch.gather raises here in the reference, losses.py:70-73,85-88)")
        assert (host >= 1.0).all()
        min_index = np.argmin(host)
        return losses[int(min_index)].to(probs.dtype), np.asarray(min_index), host.astype(np.float32), terms


    # to evaluate the quality of a collision-free solution
    @staticmethod
    def solution_score(predict, brick_layout, super_contour_area=None, device=None):
        """losses.py:120-148, as `create_solution` (util/algorithms.py:210-220) calls it at the end of every greedy
        solve:  AVG_AREA_WEIGHT * filled_area + ALIGN_LENGTH_WEIGHT * (aligned length / perimeter of the selected tiles).
        The two dot products and the perimeter sum run on the GPU (`tgnn_solution_score_sums`, csrc/loss.hip).  The one
        shapely number of the reference, `brick_layout.get_super_contour_poly().area` (a polygon union), is taken from
        `super_contour_area`, else `brick_layout.super_contour_area`, else the layout's own `get_super_contour_poly()`
        when it is the reference's class.  Perimeters come from the tiles' vertex rings (`Tile.get_perimeter`)."""
        if device is None:
            device = torch.device("cuda")
        predict_h = np.asarray(predict, dtype=np.float64)
        x, adj_edge_index, adj_edge_features, _, _ = brick_layout.get_data_as_torch_tensor(device)
        n = int(x.shape[0])
        if predict_h.shape != (n,):
            raise ValueError(f"predict must have one entry per layout node ({n}), got {predict_h.shape}")
        if super_contour_area is None:
            super_contour_area = getattr(brick_layout, "super_contour_area", None)
        if super_contour_area is None and hasattr(brick_layout, "get_super_contour_poly"):
            super_contour_area = brick_layout.get_super_contour_poly().area
        if super_contour_area is None:
            raise ValueError("solution_score needs the area of the layout's super contour (a shapely polygon union in the "
                             "reference, brick_layout.py:180-188): pass super_contour_area= or set "
                             "brick_layout.super_contour_area")
        cg = brick_layout.complete_graph
        perims = getattr(brick_layout, "_tile_perimeters", None)
        if perims is None or perims.shape[0] != n:
            inv = brick_layout.inverse_index
            perims = np.array([cg.tiles[inv[i]].get_perimeter() for i in range(n)], dtype=np.float64)
            try:
                brick_layout._tile_perimeters = perims
            except AttributeError:
                pass
        p = torch.from_numpy(predict_h).float().to(device)                       # :122
        xf = ops._f32c(x, "node_feature")
        fx = int(xf.shape[1])
        e_adj = int(adj_edge_features.shape[0]) if adj_edge_features.numel() > 0 else 0
        adj = ops._check_edge_index(adj_edge_index, "adj_edge_index") if e_adj else None
        attr = ops._f32c(adj_edge_features, "adj_edge_features") if e_adj else None
        per = torch.from_numpy(perims).float().to(device)
        sums = torch.empty(3, dtype=torch.float64, device=device)
        ws_bytes = lib.tgnn_unsupervised_loss_workspace_bytes(1)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            check(lib.tgnn_solution_score_sums(ptr(p), C.c_void_p(xf.data_ptr() + 4 * (fx - 1)), fx, ptr(per), n,
                                               ptr(adj) if e_adj else None, e_adj,
                                               C.c_void_p(attr.data_ptr() + 4) if e_adj else None,
                                               int(attr.shape[1]) if e_adj else 1, ptr(sums), ptr(ws), ws_bytes,
                                               _lib.current_stream(device)))
        s0, s1, s2 = sums.cpu().tolist()
        if math.isnan(s0) or math.isnan(s1):
            raise IndexError(f"edge index out of range [0, {n}) in the layout's align_edge_index")
        filled_area = s0 * float(cg.max_area) / float(super_contour_area)                     # :126
        assert -1e-7 <= filled_area <= 1 + 1e-7, filled_area                                  # :127
        loss_align_length = s1 * float(cg.max_align_length) if e_adj else 0.0                 # :131-141
        all_edge_length = s2                                                                  # :143-144
        ratio = loss_align_length / all_edge_length                                           # (ZeroDivisionError on an empty selection, as in the reference)
        assert -1e-7 < ratio < 1 + 1e-7, ratio                                                # :146
        wc, wl, wa = loss_weights()
        return float(wa * filled_area + wl * ratio)                                           # :148


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
