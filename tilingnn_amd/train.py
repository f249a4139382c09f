"""The training step of TilinGNN on the GPU: forward in train mode with what the backward needs kept, and the backward
itself, scheduled kernel by kernel over the C ABI.

Reference: `Trainer.train` (/root/reference/solver/ml_solver/trainer.py:68-84):
    probs = network(x, adj_e_index, adj_e_features, col_e_idx); loss = Losses.calculate_unsupervised_loss(probs, ...);
    loss.backward(); optimizer.step()
torch.autograd records the graph there.  Here the network is ONE autograd node (`TrainStep`): its forward is the fused
inference forward with the pre-BatchNorm activations, the BatchNorm records and the GIN aggregates kept
(`tgnn_forward_train`); its backward walks the layers in reverse with the adjoint kernels of csrc/backward.hip.  Nothing is
differentiated by torch: `Function.backward` hands the finished parameter gradients to autograd, which only stores
them in `.grad` for the caller's optimizer (the reference passes one in, network_train.py).

Adjoints, per forward kernel (C = 32, T edge types, D layers):
  Linear_trans     dz = BN/activation backward (tgnn_bn_bwd_reduce/_apply); dW = dz^T x (tgnn_wgrad); db = colsum;
                   dx = dz W (tgnn_dense_act_fwd with W^T).
  merge            tgnn_merge_bwd_reduce: dy1 = dh BN2(a2), dy2 = dh BN1(a1) + carry, residual slot += dh, and the six
                   column sums of both BatchNorm backward passes in the same sweep.
  NNConv (mean)    with g = dz / deg:  S' = per-node sums of gathered g rows per edge type over the TRANSPOSED graph
                   (tgnn_nnconv_type_sum, root slot = dz) gives the input gradient  dh = S' [W_t^T; root^T]  as ONE dense
                   product and all weight gradients  [dW_0 .. dW_{T-1}, d root] = h^T S'  as ONE weight-gradient product.
                   The edge MLP (T rows) is back-propagated with the generic dense pieces.
  GIN              t1, t2 re-derived from the kept aggregate; three sigmoid/Linear adjoints; the aggregation's adjoint is
                   the aggregation on the transposed collision graph (tgnn_gin_aggregate).
Width 32 only (the reference's network_width, inputs/config.py:38); other widths raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch

from . import _lib, ops
from ._lib import ACT_LEAKY_RELU, ACT_NONE, ACT_SIGMOID, check, lib, ptr

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------ thin wrappers
def _s(t: Tensor):
    return _lib.current_stream(t.device)


class _Scratch:
    """Reduction / weight-gradient workspaces, grown on demand, one set per device (stream-ordered reuse)."""
    _bufs: Dict = {}

    @classmethod
    def get(cls, kind: str, nbytes: int, device) -> Tensor:
        key = (kind, torch.device(device).index)
        buf = cls._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = cls._bufs[key] = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        return buf


_zero_bufs: Dict = {}


def _zeros(n: int, device) -> Tensor:
    """A shared all-zero bias (never written)."""
    key = torch.device(device).index
    buf = _zero_bufs.get(key)
    if buf is None or buf.numel() < n:
        buf = _zero_bufs[key] = torch.zeros(max(n, 1024), dtype=torch.float32, device=device)
    return buf[:n]


def colsum(x: Tensor) -> Tensor:
    n, c = int(x.shape[0]), int(x.shape[1])
    out = torch.empty(c, dtype=torch.float32, device=x.device)
    nb = lib.tgnn_reduce_workspace_bytes(c)
    ws = _Scratch.get("red", nb, x.device)
    check(lib.tgnn_colsum(ptr(x), x.stride(0), n, c, ptr(out), ptr(ws), nb, _s(x)))
    return out


def wgrad(dz: Tensor, x: Tensor, slot_major: bool = False, with_bias: bool = False):
    """dz^T . x -> [cout, cin] (with_bias: also the column sums of dz, from the same pass).
    slot_major: x is the [S, N, C] skip buffer read as [N, S * C]."""
    n, cout = int(dz.shape[0]), int(dz.shape[1])
    if slot_major:
        cin, ld_x, kb = int(x.shape[0]) * int(x.shape[2]), int(x.shape[2]), int(x.shape[1]) * int(x.shape[2])
        if int(x.shape[2]) != 32:
            raise NotImplementedError("slot-major weight gradient: width 32 only")
    else:
        cin, ld_x, kb = int(x.shape[1]), x.stride(0), 0
    out = torch.empty(cout, cin, dtype=torch.float32, device=dz.device)
    dbias = torch.empty(cout, dtype=torch.float32, device=dz.device) if with_bias else None
    nb = lib.tgnn_wgrad_workspace_bytes(n, cout, cin)
    ws = _Scratch.get("wgrad", nb, dz.device)
    check(lib.tgnn_wgrad(ptr(dz), dz.stride(0), ptr(x), ld_x, kb, n, cout, cin, ptr(out), ptr(dbias), ptr(ws), nb, _s(dz)))
    return (out, dbias) if with_bias else out


def transpose(w: Tensor) -> Tensor:
    out = torch.empty(int(w.shape[1]), int(w.shape[0]), dtype=torch.float32, device=w.device)
    check(lib.tgnn_transpose(ptr(w), int(w.shape[0]), int(w.shape[1]), ptr(out), _s(w)))
    return out


def dense_dx(dz: Tensor, weight: Tensor) -> Tensor:
    """dz [N, out] . weight [out, in] -> [N, in]: the forward dense kernel on the transposed weight."""
    wt = transpose(weight)
    return ops.dense_act(dz, wt, _zeros(int(wt.shape[0]), dz.device), ACT_NONE)[0]


def sigmoid_bwd(d: Tensor, t: Tensor) -> Tensor:
    n, c = int(d.shape[0]), int(d.shape[1])
    out = torch.empty(n, c, dtype=torch.float32, device=d.device)
    check(lib.tgnn_sigmoid_bwd(ptr(d), d.stride(0), ptr(t), t.stride(0), n, c, ptr(out), c, _s(d)))
    return out


def add_into(src: Tensor, dst: Tensor) -> None:
    check(lib.tgnn_add_into(ptr(src), src.stride(0), int(src.shape[0]), int(src.shape[1]), ptr(dst), dst.stride(0),
                            _s(src)))


def bn_bwd(dy: Tensor, a: Tensor, stat: Tensor, eps: float, act: int, row_scale: Optional[Tensor] = None):
    """-> (dz, dgamma, dbeta[, dz * row_scale])."""
    n, f = int(a.shape[0]), int(a.shape[1])
    dev = a.device
    coef = torch.empty(2, f, dtype=torch.float32, device=dev)
    dgamma = torch.empty(f, dtype=torch.float32, device=dev)
    dbeta = torch.empty(f, dtype=torch.float32, device=dev)
    nb = lib.tgnn_reduce_workspace_bytes(f)
    ws = _Scratch.get("red", nb, dev)
    check(lib.tgnn_bn_bwd_reduce(ptr(dy), dy.stride(0), ptr(a), a.stride(0), ptr(stat), n, f, eps, ptr(coef), ptr(dgamma),
                                 ptr(dbeta), ptr(ws), nb, _s(a)))
    return (*bn_bwd_apply(dy, a, stat, coef, act, row_scale), dgamma, dbeta)


def bn_bwd_apply(dy: Tensor, a: Tensor, stat: Tensor, coef: Tensor, act: int, row_scale: Optional[Tensor] = None):
    n, f = int(a.shape[0]), int(a.shape[1])
    dz = torch.empty(n, f, dtype=torch.float32, device=a.device)
    scaled = torch.empty(n, f, dtype=torch.float32, device=a.device) if row_scale is not None else None
    check(lib.tgnn_bn_bwd_apply(ptr(dy), dy.stride(0), ptr(a), a.stride(0), ptr(stat), ptr(coef), n, f, act, ptr(dz), f,
                                ptr(row_scale), ptr(scaled), f, _s(a)))
    return dz, scaled


def type_sum(rows: Tensor, own: Tensor, root_scale: Optional[Tensor], rowptr: Tensor, src: Tensor, typ: Tensor,
             n: int, n_types: int) -> Tensor:
    out = torch.empty(n, (n_types + 1) * 32, dtype=torch.float32, device=rows.device)
    check(lib.tgnn_nnconv_type_sum(ptr(rows), rows.stride(0), ptr(own), own.stride(0), ptr(root_scale), ptr(rowptr),
                                   ptr(src), ptr(typ), n, n_types, 32, ptr(out), _s(rows)))
    return out


def gin_aggregate(a: Tensor, rowptr: Tensor, src: Tensor, eps: Tensor, n: int) -> Tensor:
    z = torch.empty(n, 32, dtype=torch.float32, device=a.device)
    check(lib.tgnn_gin_aggregate(ptr(a), a.stride(0), None, ptr(rowptr), ptr(src), ptr(eps), n, 32, ptr(z), _s(a)))
    return z


# ------------------------------------------------------------------------------------------------ graph, both ways
class TrainGraph:
    """The prepared graph of the forward plus what only the backward reads: CSR of the TRANSPOSED edge sets (gathers
    along out-edges: the adjoint of a gather along in-edges) and the in-degrees of the mean."""

    def __init__(self, graph: ops.PreparedGraph, adj_e_index: Tensor, col_e_idx: Tensor):
        n = graph.n_nodes
        self.g = graph
        dev = graph.adj_rowptr.device
        adj_t = adj_e_index.flip(0).contiguous()                     # edge e becomes (dst_e -> src_e)
        col_t = col_e_idx.flip(0).contiguous()
        self.adjT_rowptr, self.adjT_src, eid, _ = ops.build_csr(adj_t, n, False)
        ea = graph.n_adj_edges
        self.adjT_type = torch.empty(max(ea, 1), dtype=torch.int32, device=dev)
        check(lib.tgnn_gather_i32(ptr(graph.edge_type), ea, ptr(eid), ea, ptr(self.adjT_type), _lib.current_stream(dev)))
        self.colT_rowptr, self.colT_src, _, _ = ops.build_csr(col_t, n, True)
        self.deg = torch.empty(n, dtype=torch.float32, device=dev)
        self.inv_deg = torch.empty(n, dtype=torch.float32, device=dev)
        check(lib.tgnn_csr_degree(ptr(graph.adj_rowptr), n, ptr(self.deg), ptr(self.inv_deg), _lib.current_stream(dev)))


def _train_graph(net, n: int, adj_e_index: Tensor, adj_e_features: Tensor, col_e_idx: Tensor) -> TrainGraph:
    from .graph_networks import _graph_cache
    graph = _graph_cache.get_full(n, adj_e_index, adj_e_features, col_e_idx) if net.cache_graph else \
        ops.prepare_graph(n, adj_e_index, adj_e_features, col_e_idx)
    tg = graph.__dict__.get("_train")
    if tg is None:
        tg = TrainGraph(graph, adj_e_index, col_e_idx)
        if net.cache_graph:
            graph.__dict__["_train"] = tg
    return tg


# ------------------------------------------------------------------------------------------------ forward, keeping
class _Saved:
    pass


def forward_train(net, x: Tensor, adj_e_index: Tensor, adj_e_features: Tensor, col_e_idx: Tensor):
    """The forward of the training step: `tgnn_forward_train` = the fused inference forward (same kernels, same two-stream
    schedule) writing what the backward reads into buffers that outlive it.  -> (probs, saved)."""
    c, depth = net.network_width, net.network_depth
    if c != 32:
        raise NotImplementedError("the training path is built for network_width = 32 (inputs/config.py:38)")
    n = int(x.shape[0])
    if n < 2:
        raise ValueError("Expected more than 1 value per channel when training")
    table, dev = net._param_table()
    xf, ea = ops._f32c(x, "x"), ops._f32c(adj_e_features, "adj_e_features")
    tg = _train_graph(net, n, adj_e_index, adj_e_features, col_e_idx)
    g = tg.g
    T = g.n_types
    if T > 63:
        raise NotImplementedError("the training path holds at most 63 distinct edge-attribute rows")
    sv = _Saved()
    sv.tg, sv.x, sv.ea, sv.n = tg, xf, ea, n
    f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
    sv.init_a, sv.init_stat = [f(n, c), f(n, c)], [f(4, c), f(4, c)]
    a1, a2, u, st1, st2 = f(depth, n, c), f(depth, n, c), f(depth, n, c), f(depth, 4, c), f(depth, 4, c)
    fin_dims = [int(l.linear.out_features) for l in net.final_mlp[0].mlp]
    sv.fin_a, sv.fin_stat = [f(n, d) for d in fin_dims], [f(4, d) for d in fin_dims]
    sv.skip = f(depth + 1, n, c)
    wtab = f(depth, max(T, 1), c, c)
    probs = f(n, net.output_dim)
    keep = _lib.TrainSave()
    for k in range(2):
        keep.init_a[k], keep.init_stat[k] = sv.init_a[k].data_ptr(), sv.init_stat[k].data_ptr()
    for k in range(4):
        keep.fin_a[k], keep.fin_stat[k] = sv.fin_a[k].data_ptr(), sv.fin_stat[k].data_ptr()
    keep.a1, keep.a2, keep.u, keep.stat1, keep.stat2 = a1.data_ptr(), a2.data_ptr(), u.data_ptr(), st1.data_ptr(), st2.data_ptr()
    keep.skip, keep.wtab = sv.skip.data_ptr(), wtab.data_ptr()
    dims = net._dims()
    ws_bytes = lib.tgnn_forward_workspace_bytes(C.byref(dims), n, T)
    ws = _Scratch.get("fwd", ws_bytes, dev)
    gs = g.c_struct()
    check(lib.tgnn_forward_train(C.byref(dims), table, ptr(xf), ptr(ea), C.byref(gs), C.byref(keep), ptr(probs), ptr(ws),
                                 ws_bytes, _lib.current_stream(dev), _lib.side_stream(dev)))
    sv.keep, sv._keep_alive = keep, (a1, a2, u, st1, st2, wtab)
    sv.a1, sv.a2, sv.u = list(a1), list(a2), list(u)
    sv.stat1, sv.stat2 = list(st1), list(st2)
    sv.wtab = [wtab[i, :T] for i in range(depth)]
    sv.probs = probs
    return probs, sv


# ------------------------------------------------------------------------------------------------ backward
def _mlp_backward(layers, acts, stats, first_input, first_slot_major, dy, grads, prefix, need_dx):
    """Backward through a Linear_trans stack with BatchNorm.  dy: gradient at the stack's (normalised) output."""
    for k in range(len(layers) - 1, -1, -1):
        layer = layers[k]
        dz, _, dgamma, dbeta = bn_bwd(dy, acts[k], stats[k], float(layer.batch_norm.eps), ops.act_code(layer.activation))
        grads[f"{prefix}.mlp.{k}.batch_norm.weight"], grads[f"{prefix}.mlp.{k}.batch_norm.bias"] = dgamma, dbeta
        if k == 0:
            inp, slot = first_input, first_slot_major
        else:
            inp, slot = ops.bn_apply(acts[k - 1], stats[k - 1]), False
        grads[f"{prefix}.mlp.{k}.linear.weight"], grads[f"{prefix}.mlp.{k}.linear.bias"] = \
            wgrad(dz, inp, slot_major=slot, with_bias=True)
        if k > 0 or need_dx:
            dy = dense_dx(dz, layer.linear.weight)
    return dy


def sigmoid_mlp_backward(weights, x: Tensor, t3: Tensor, d_out: Tensor, grads: Dict[str, Tensor], names, need_dx: bool):
    """Three Linear + Sigmoid layers without BatchNorm (GraphConv's edge MLP, GINConv's MLP) in one library call
    (tgnn_sigmoid_mlp_bwd).  weights = [w1, b1, w2, b2, w3, b3]; t3 = the MLP's output; names = the `...mlp.k` prefixes."""
    n, d0 = int(x.shape[0]), int(x.shape[1])
    d1, d2, d3 = int(weights[0].shape[0]), int(weights[2].shape[0]), int(weights[4].shape[0])
    dev = x.device
    gw = [torch.empty_like(weights[2 * k]) for k in range(3)]
    gb = [torch.empty_like(weights[2 * k + 1]) for k in range(3)]
    dx = torch.empty(n, d0, dtype=torch.float32, device=dev) if need_dx else None
    nb = lib.tgnn_sigmoid_mlp_bwd_workspace_bytes(n, d0, d1, d2, d3)
    ws = _Scratch.get("mlp_bwd", nb, dev)
    if not (x.is_contiguous() and t3.is_contiguous()):
        raise ValueError("sigmoid_mlp_backward: x and t3 must be contiguous")
    check(lib.tgnn_sigmoid_mlp_bwd(ptr(x), n, d0, d1, d2, d3, ptr(weights[0]), ptr(weights[1]), ptr(weights[2]),
                                   ptr(weights[3]), ptr(weights[4]), ptr(t3), ptr(d_out), d_out.stride(0), ptr(gw[0]),
                                   ptr(gb[0]), ptr(gw[1]), ptr(gb[1]), ptr(gw[2]), ptr(gb[2]), ptr(dx), ptr(ws), nb, _s(x)))
    for k in range(3):
        grads[names[k] + ".linear.weight"], grads[names[k] + ".linear.bias"] = gw[k], gb[k]
    return dx


def nnconv_backward(conv, prefix: str, tg: TrainGraph, wtab: Tensor, h: Tensor, dz: Tensor, g_scaled: Tensor,
                    edge_attr: Tensor, grads: Dict[str, Tensor]) -> Tensor:
    """Adjoint of NNConv mean (edge_conv.py:25).  wtab [T, C, C]: the edge-type matrices the forward used; h: the layer's
    input; dz: gradient at the conv's output; g_scaled = dz / deg.  Fills the gradients of root, bias and the edge MLP
    under `prefix`; returns the gradient at h."""
    g = tg.g
    n, T, c = g.n_nodes, g.n_types, 32
    dev = h.device
    # ONE gather pass serves both gradients: S'[j][t] = sum over the out-edges (j -> v) of type t of g[v]  (type sums over
    # the TRANSPOSED graph; root slot = g[j] deg[j] = dz[j]).
    #   input gradient:   dh[j]  = sum_t S'[j][t] W_t^T + dz[j] root^T              = S' . [W_t^T; root^T]   (dense)
    #   weight gradients: dW_t   = sum_e h[src_e]^T g[dst_e] = sum_j h[j]^T S'[j][t];  d root = h^T dz       = h^T . S'
    s_bwd = type_sum(g_scaled, g_scaled, tg.deg, tg.adjT_rowptr, tg.adjT_src, tg.adjT_type, n, T)   # [N, (T+1) C]
    wd = torch.empty(c, (T + 1) * c, dtype=torch.float32, device=dev)                             # [in][t][out]; t = T: root
    check(lib.tgnn_swap_leading(ptr(wtab), T, c, c, ptr(wd), T + 1, _s(wd)))
    check(lib.tgnn_swap_leading(ptr(conv.root), 1, c, c, C.c_void_p(wd.data_ptr() + 4 * T * c), T + 1, _s(wd)))
    dh = ops.dense_act(s_bwd, wd, _zeros(c, dev), ACT_NONE)[0]
    dw_in_major = wgrad(h, s_bwd)                                                                 # [in][t][out]
    dwcat = torch.empty((T + 1) * c, c, dtype=torch.float32, device=dev)                          # [t][in][out]
    check(lib.tgnn_swap_leading(ptr(dw_in_major), c, T + 1, c, ptr(dwcat), c, _s(dwcat)))
    grads[prefix + ".nnConv.root"] = dwcat[T * c:]
    grads[prefix + ".nnConv.bias"] = colsum(dz)
    # the edge MLP behind the T weight matrices (edge_conv.py:17-18), on the T distinct attribute rows
    ew = conv._edge_mlp_params()
    names = [f"{prefix}.mlp.mlp.{k}" for k in range(3)]
    if T:
        fe = int(edge_attr.shape[1])
        rows = torch.empty(T, fe, dtype=torch.float32, device=dev)
        check(lib.tgnn_rows_gather(ptr(edge_attr), fe, ptr(g.type_rep_edge), T, fe, ptr(rows), fe, _s(rows)))
        sigmoid_mlp_backward(ew, rows, wtab.reshape(T, c * c), dwcat[:T * c].reshape(T, c * c), grads, names, False)
    else:
        for k in range(3):
            grads[names[k] + ".linear.weight"] = torch.zeros_like(ew[2 * k])
            grads[names[k] + ".linear.bias"] = torch.zeros_like(ew[2 * k + 1])
    return dh


def gin_backward(conv, prefix: str, tg: TrainGraph, u: Tensor, t3: Tensor, dz: Tensor, grads: Dict[str, Tensor]) -> Tensor:
    """Adjoint of GINConv (coll_conv.py:25): u = the aggregate the MLP read, t3 = the MLP's output (= the layer's
    pre-BatchNorm activation: LeakyReLU is the identity on a sigmoid), dz: gradient at t3.  Returns the gradient at the
    conv's input: the aggregation run over the transposed collision graph."""
    du = sigmoid_mlp_backward(conv._mlp_params(), u, t3, dz, grads, [f"{prefix}.ginConv.nn.mlp.{k}" for k in range(3)], True)
    return gin_aggregate(du, tg.colT_rowptr, tg.colT_src, conv.eps, tg.g.n_nodes)


def backward_train(net, sv, dprobs: Tensor) -> Dict[str, Tensor]:
    c, depth, n = net.network_width, net.network_depth, sv.n
    tg = sv.tg
    g = tg.g
    T = g.n_types
    dev = sv.x.device
    grads: Dict[str, Tensor] = {}
    dprobs = ops._f32c(dprobs, "grad of probs")

    # ---- final Linear_trans (32 -> out, Sigmoid, no BatchNorm), then the final MLP
    last = net.final_mlp[1]
    dl = sigmoid_bwd(dprobs, sv.probs)
    y = ops.bn_apply(sv.fin_a[-1], sv.fin_stat[-1])
    grads["final_mlp.1.linear.weight"], grads["final_mlp.1.linear.bias"] = wgrad(dl, y, with_bias=True)
    dy = dense_dx(dl, last.linear.weight)
    dcat = _mlp_backward(list(net.final_mlp[0].mlp), sv.fin_a, sv.fin_stat, sv.skip, True, dy, grads, "final_mlp.0", True)
    ld = int(dcat.shape[1])                                          # (D + 1) * C: slot s = columns [s C, (s + 1) C)

    def slot(s):
        return dcat[:, s * c:(s + 1) * c]

    nb = lib.tgnn_reduce_workspace_bytes(c)
    ws = _Scratch.get("red", nb, dev)
    carry = None
    for i in range(depth - 1, -1, -1):
        l1, l2 = net.brch_1_graph_conv_layers[i], net.brch_2_coll_conv_layers[i]
        p1, p2 = f"brch_1_graph_conv_layers.{i}", f"brch_2_coll_conv_layers.{i}"
        a1, a2, st1, st2 = sv.a1[i], sv.a2[i], sv.stat1[i], sv.stat2[i]
        dy1 = torch.empty(n, c, dtype=torch.float32, device=dev)
        dy2 = torch.empty(n, c, dtype=torch.float32, device=dev)
        coef = torch.empty(2, 2, c, dtype=torch.float32, device=dev)
        dgb = torch.empty(4, c, dtype=torch.float32, device=dev)
        dh = slot(i + 1)
        resid = slot(i - 2) if i >= 2 else None
        check(lib.tgnn_merge_bwd_reduce(ptr(dh), ld, ptr(a1), ptr(st1), ptr(a2), ptr(st2), ptr(carry), n, c,
                                        float(l1.batch_norm.eps), float(l2.batch_norm.eps), ptr(dy1), ptr(dy2),
                                        ptr(resid), ld, ptr(coef[0]), ptr(dgb[0]), ptr(dgb[1]), ptr(coef[1]), ptr(dgb[2]),
                                        ptr(dgb[3]), ptr(ws), nb, _s(dy1)))
        grads[p1 + ".batch_norm.weight"], grads[p1 + ".batch_norm.bias"] = dgb[0], dgb[1]
        grads[p2 + ".batch_norm.weight"], grads[p2 + ".batch_norm.bias"] = dgb[2], dgb[3]
        dz1, gsc = bn_bwd_apply(dy1, a1, st1, coef[0], ACT_LEAKY_RELU, row_scale=tg.inv_deg)      # gsc = dz1 / deg
        dz2, _ = bn_bwd_apply(dy2, a2, st2, coef[1], ACT_LEAKY_RELU)

        dh1 = nnconv_backward(l1.nnConv, p1, tg, sv.wtab[i], sv.skip[i], dz1, gsc, sv.ea, grads)
        add_into(dh1, slot(i))
        carry = gin_backward(l2.ginConv, p2, tg, sv.u[i], a2, dz2, grads)
    add_into(carry, slot(0))                                       # h2 of layer 0 is the init output (TilinGNN.py:55)

    _mlp_backward(list(net.init_node_feature_trans.mlp), sv.init_a, sv.init_stat, sv.x, False, slot(0), grads,
                  "init_node_feature_trans", False)
    return grads


# The backward schedule runs inside the library (csrc/train.hip: tgnn_backward) by default; backward_train above is the
# same schedule spelled out call by call -- kept as the readable statement and as the checker (tests compare the two bit
# for bit).  TGNN_PY_BACKWARD=1 or train.USE_LIBRARY_BACKWARD = False selects it.
USE_LIBRARY_BACKWARD = os.environ.get("TGNN_PY_BACKWARD", "0") != "1"


def backward_library(net, sv, dprobs: Tensor) -> Dict[str, Tensor]:
    """One call: tgnn_backward.  -> {parameter name: gradient}."""
    tg = sv.tg
    g = tg.g
    dev = sv.x.device
    table, _ = net._param_table()
    dims = net._dims()
    layout = net.__dict__.get("_tgnn_grad_layout")
    if layout is None or layout[0] != id(table):
        # once per parameter table: which table entries are parameters, and where each sits in ONE flat gradient buffer
        names = _lib.param_names(dims)
        params = dict(net.named_parameters())
        entries, off = [], 0
        for k, name in enumerate(names):
            p_ = params.get(name)
            if p_ is not None:
                entries.append((k, name, off, p_.numel(), tuple(p_.shape)))
                off += (p_.numel() + 63) // 64 * 64                  # 256-byte aligned slices
        layout = net.__dict__["_tgnn_grad_layout"] = (id(table), len(names), entries, off)
    _, n_names, entries, total = layout
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    base = flat.data_ptr()
    grads, gtable = {}, (C.c_void_p * n_names)()
    for k, name, off, numel, shape in entries:
        grads[name] = flat[off:off + numel].view(shape)
        gtable[k] = base + 4 * off
    keep = sv.keep
    tgd = _lib.TrainGraphDesc(tg.adjT_rowptr.data_ptr(), tg.adjT_src.data_ptr(), tg.adjT_type.data_ptr(),
                              tg.colT_rowptr.data_ptr(), tg.colT_src.data_ptr(), tg.deg.data_ptr(), tg.inv_deg.data_ptr())
    nb = lib.tgnn_backward_workspace_bytes(C.byref(dims), sv.n, g.n_types)
    ws = _Scratch.get("bwd", nb, dev)
    gs = g.c_struct()
    dp = ops._f32c(dprobs, "grad of probs")
    check(lib.tgnn_backward(C.byref(dims), table, gtable, ptr(sv.x), ptr(sv.ea), C.byref(gs), C.byref(tgd), C.byref(keep),
                            ptr(sv.probs), ptr(dp), ptr(ws), nb, _lib.current_stream(dev)))
    return grads


class TrainStep(torch.autograd.Function):
    """probs = TrainStep.apply(net, x, adj_e_index, adj_e_features, col_e_idx, *net.parameters())"""

    @staticmethod
    def forward(ctx, net, x, adj_e_index, adj_e_features, col_e_idx, *params):
        with _lib.pinned_stream(x.device):
            probs, sv = forward_train(net, x, adj_e_index, adj_e_features, col_e_idx)
        ctx.net, ctx.sv = net, sv
        return probs

    @staticmethod
    def backward(ctx, dprobs):
        with _lib.pinned_stream(dprobs.device):
            run = backward_library if USE_LIBRARY_BACKWARD else backward_train
            grads = run(ctx.net, ctx.sv, dprobs.contiguous())
        ctx.sv = None
        out = []
        for name, p in ctx.net.named_parameters():
            if name not in grads:
                raise KeyError(f"no gradient was produced for `{name}`")
            out.append(grads[name].reshape(p.shape))
        return (None, None, None, None, None, *out)


def forward_with_grad(net, x, adj_e_index, adj_e_features, col_e_idx):
    params = [p for _, p in net.named_parameters()]
    return TrainStep.apply(net, x, adj_e_index, adj_e_features, col_e_idx, *params)
