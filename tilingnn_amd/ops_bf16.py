"""Torch-tensor front end of the bf16-storage path (BASELINE config 3: network_width 64; csrc/bf16_path.hip).

Same role as `ops.py` for the fp32 path: validation, allocation, stream plumbing around one C-ABI call each.  Activations
that cross HBM between kernels are `torch.bfloat16` tensors [N, 64]; weights stay fp32 module parameters and are rounded
to bf16 operand images inside the library; BatchNorm partial sums are fp64, stat records fp32.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _lib, ops
from ._lib import check, lib, ptr

Tensor = torch.Tensor
WIDTH = 64


def _bf16c(t: Tensor, name: str) -> Tensor:
    ops._need_gpu(t, name)
    if t.dtype != torch.bfloat16:
        raise ValueError(f"`{name}` must be bfloat16, got {t.dtype}")
    if t.dim() != 2 or t.shape[1] != WIDTH:
        raise ValueError(f"`{name}` must be [N, {WIDTH}], got {tuple(t.shape)}")
    return t if t.is_contiguous() else t.contiguous()


def to_bf16(t: Tensor) -> Tensor:
    """Round-to-nearest-even conversion on the device (tgnn_f32_to_bf16)."""
    t = ops._f32c(t, "x")
    out = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    check(lib.tgnn_f32_to_bf16(ptr(t), t.numel(), ptr(out), _lib.current_stream(t.device)))
    return out


def max_types() -> int:
    return 16           # (T + 1) x 8 KB of weight image + the per-wave staging must fit 160 KB of LDS


def nnconv64(h: Tensor, graph: ops.PreparedGraph, wtab: Tensor, root: Tensor, bias: Tensor, act: int = ops.ACT_NONE,
             partials: Optional[Tensor] = None, kernel: Optional[str] = None) -> Tuple[Tensor, int]:
    """kernel: None = over the structure the layout carries (edge groups where it has them, as tgnn_forward_bf16 does), "eg" /
    "cols" = over its edge groups / type columns (built on first use)."""
    h = _bf16c(h, "x")
    n = graph.n_nodes
    if tuple(root.shape) != (WIDTH, WIDTH) or tuple(bias.shape) != (WIDTH,):
        raise ValueError("NNConv root/bias shape mismatch")
    out = torch.empty(n, WIDTH, dtype=torch.bfloat16, device=h.device)
    wimg = torch.empty(lib.tgnn_nnconv64_image_elems(graph.n_types), dtype=torch.bfloat16, device=h.device)
    npart = C.c_int32(0)
    if kernel == "eg" or (kernel is None and graph.groups is not None):
        grp = ops.graph_groups(graph)
        if grp is None:
            raise ValueError("the layout has more edge types than the edge-group structure takes")
        check(lib.tgnn_nnconv64_bf16_eg_fwd(ptr(h), int(h.shape[0]), ptr(grp.tile_grp_ptr), ptr(grp.grp), ptr(ops._f32c(wtab, "wtab")),
                                            graph.n_types, ptr(ops._f32c(root, "root")), ptr(ops._f32c(bias, "bias")), n, act,
                                            ptr(out), ptr(wimg), ptr(partials), C.byref(npart), _lib.current_stream(h.device)))
        return out, npart.value
    if ops.graph_columns(graph) is None:
        raise ValueError("the bf16 NNConv runs on the type-column structure (prepare_graph(columns=True))")
    tl = ops.graph_columns(graph)
    check(lib.tgnn_nnconv64_bf16_fwd(ptr(h), int(h.shape[0]), ptr(tl.tile_col_ptr), ptr(tl.col_meta), ptr(tl.col_src),
                                     ptr(ops._f32c(wtab, "wtab")), graph.n_types, ptr(ops._f32c(root, "root")),
                                     ptr(ops._f32c(bias, "bias")), n, act, ptr(out), ptr(wimg), ptr(partials),
                                     C.byref(npart), _lib.current_stream(h.device)))
    return out, npart.value


def gin64(a: Tensor, graph: ops.PreparedGraph, eps: Tensor, w1, b1, w2, b2, w3, b3, act: int = ops.ACT_NONE,
          in_stat: Optional[Tensor] = None, partials: Optional[Tensor] = None) -> Tuple[Tensor, int]:
    a = _bf16c(a, "x")
    n = graph.n_nodes
    if tuple(w1.shape) != (32, WIDTH) or tuple(w2.shape) != (64, 32) or tuple(w3.shape) != (WIDTH, 64):
        raise ValueError("GIN MLP shape mismatch")
    out = torch.empty(n, WIDTH, dtype=torch.bfloat16, device=a.device)
    z = torch.empty(n, WIDTH, dtype=torch.bfloat16, device=a.device)
    npart = C.c_int32(0)
    ps = [ops._f32c(p, "gin parameter") for p in (eps, w1, b1, w2, b2, w3, b3)]
    check(lib.tgnn_gin64_bf16_fwd(ptr(a), ptr(in_stat), ptr(graph.col_rowptr), ptr(graph.col_src), *[ptr(p) for p in ps], n,
                                  act, ptr(out), ptr(z), ptr(partials), C.byref(npart), _lib.current_stream(a.device)))
    return out, npart.value


def collconv64(h2: Tensor, graph: ops.PreparedGraph, eps: Tensor, w1, b1, w2, b2, w3, b3, bn: torch.nn.BatchNorm1d,
               update_running: bool = True) -> Tensor:
    """CollConv.forward incl. its train-mode BatchNorm -> the normalised output, bf16 [N, 64] (one pass over the MLP leaves the
    statistics and the fp32 rows; an element-wise pass normalises and rounds them)."""
    h2 = _bf16c(h2, "x")
    n = graph.n_nodes
    dev = h2.device
    out = torch.empty(n, WIDTH, dtype=torch.bfloat16, device=dev)
    z = torch.empty(n, WIDTH, dtype=torch.bfloat16, device=dev)
    pre = torch.empty(n, WIDTH, dtype=torch.float32, device=dev)
    stat = torch.empty(4 * WIDTH, dtype=torch.float32, device=dev)
    parts = ops.new_partials(WIDTH, dev)
    ps = [ops._f32c(p, "gin parameter") for p in (eps, w1, b1, w2, b2, w3, b3)]
    upd = update_running and bn.track_running_stats
    check(lib.tgnn_collconv64_bf16_fwd(ptr(h2), ptr(graph.col_rowptr), ptr(graph.col_src), *[ptr(p) for p in ps],
                                       ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean if upd else None),
                                       ptr(bn.running_var if upd else None), ptr(bn.num_batches_tracked if upd else None), n,
                                       ptr(out), ptr(z), ptr(pre), ptr(stat), ptr(parts), _lib.current_stream(dev)))
    return out


def merge(a1: Tensor, stat1: Tensor, a2: Tensor, stat2: Optional[Tensor], resid: Optional[Tensor]) -> Tensor:
    a1, a2 = _bf16c(a1, "a1"), _bf16c(a2, "a2")
    out = torch.empty_like(a1)
    check(lib.tgnn_merge_bf16_fwd(ptr(a1), ptr(stat1), ptr(a2), ptr(stat2), ptr(_bf16c(resid, "resid") if resid is not None else None),
                                  int(a1.shape[0]), WIDTH, ptr(out), _lib.current_stream(a1.device)))
    return out


def dense_slots(mid: Tensor, weight: Tensor, bias: Tensor, act: int, partials: Optional[Tensor] = None) -> Tuple[Tensor, int]:
    """act(cat(mid slots) @ weight.T + bias) for the bf16 skip buffer [S, N, 64] -> fp32 [N, out]."""
    ops._need_gpu(mid, "mid")
    if mid.dtype != torch.bfloat16 or mid.dim() != 3 or mid.shape[2] != WIDTH or not mid.is_contiguous():
        raise ValueError(f"mid must be a contiguous bfloat16 [S, N, {WIDTH}] tensor")
    s, n = int(mid.shape[0]), int(mid.shape[1])
    m = int(weight.shape[0])
    if int(weight.shape[1]) != s * WIDTH:
        raise ValueError(f"Linear expects in_dim {int(weight.shape[1])}, got {s * WIDTH}")
    out = torch.empty(n, m, dtype=torch.float32, device=mid.device)
    wb = torch.empty(m * s * WIDTH, dtype=torch.bfloat16, device=mid.device)
    npart = C.c_int32(0)
    check(lib.tgnn_dense_bf16_slots_fwd(ptr(mid), n * WIDTH, s, ptr(ops._f32c(weight, "weight")), ptr(ops._f32c(bias, "bias")),
                                        n, m, act, ptr(out), ptr(wb), ptr(partials), C.byref(npart),
                                        _lib.current_stream(mid.device)))
    return out, npart.value


def forward(net, x: Tensor, adj_e_index: Tensor, adj_e_features: Tensor, col_e_idx: Tensor, graph=None) -> Tensor:
    """`tgnn_forward_bf16`: the whole network with bf16 activation storage -> probs fp32 [N, output_dim]."""
    if net.network_width != WIDTH:
        raise ValueError(f"bf16 activation storage is built for network_width {WIDTH} (BASELINE config 3), "
                         f"this network has {net.network_width}")
    if not net.training:
        raise ValueError("the bf16-storage forward runs BatchNorm with batch statistics (train mode, as the reference's "
                         "solver does, ml_solver.py:131)")
    table, dev = net._param_table()
    n = int(x.shape[0])
    dims = net._dims()
    xf = ops._f32c(x, "x")
    ws = None
    if graph is None:
        # [r6] a NEW layout: the init MLP needs nothing of the graph -- queued on the side stream in front of the preparation, it
        # runs beside it (tgnn_forward_bf16_begin; the workspace does not depend on the type count)
        side = _lib.side_stream_torch(dev) if n >= 2 and os.environ.get("TGNN_BF16_BEGIN", "1") == "1" else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(dev))
            ws_bytes = lib.tgnn_forward_bf16_workspace_bytes(C.byref(dims), n, 0)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            check(lib.tgnn_forward_bf16_begin(C.byref(dims), table, ptr(xf), n, 1, ptr(ws), ws_bytes, _lib.side_stream(dev)))
        try:
            graph = ops.prepare_graph(n, adj_e_index, adj_e_features, col_e_idx)
        except Exception:
            if side is not None:                                   # (begin's launches write the workspace freed below)
                torch.cuda.current_stream(dev).wait_stream(side)
            raise
    if (graph.cols is None and graph.groups is None) or graph.n_types > max_types():
        if ws is not None:
            torch.cuda.current_stream(dev).wait_stream(_lib.side_stream_torch(dev))
        raise ValueError(f"the bf16 path needs the NNConv type columns or edge groups and at most {max_types()} edge types")
    if ws is None:
        ws_bytes = lib.tgnn_forward_bf16_workspace_bytes(C.byref(dims), n, graph.n_types)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    probs = torch.empty(n, net.output_dim, dtype=torch.float32, device=dev)
    g = graph.c_struct()
    check(lib.tgnn_forward_bf16(C.byref(dims), table, ptr(xf), ptr(ops._f32c(adj_e_features, "adj_e_features")),
                                C.byref(g), 1, ptr(probs), ptr(ws), ws_bytes, _lib.current_stream(dev), _lib.side_stream(dev)))
    return probs
