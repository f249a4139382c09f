"""Torch-tensor front end of the libtgnn C ABI: validation, workspace allocation, stream plumbing.

PyTorch is used here for device memory and streams only; every function below ends in exactly
one (or a few) C-ABI calls into the hand-written gfx950 kernels -- there is no torch compute and
no CPU path.  Shape / dtype / device errors raise ValueError before anything is launched
(the reference's `assert x.shape[-1] == in_dim`, layers/util.py:16, becomes a ValueError).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ACT_LEAKY_RELU, ACT_NONE, ACT_SIGMOID, BN_MAX_PARTIALS, check, lib, ptr

Tensor = torch.Tensor


def _need_gpu(t: Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"tilingnn_amd: `{name}` lives on {t.device}; this package only runs on an AMD GPU "
            "(no CPU fallback exists by design -- move the module and its inputs to 'cuda').")


def _f32c(t: Tensor, name: str) -> Tensor:
    _need_gpu(t, name)
    if t.dtype != torch.float32:
        raise ValueError(f"`{name}` must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _stream(t: Tensor) -> C.c_void_p:
    return _lib.current_stream(t.device)


def act_code(activation) -> int:
    """Map the reference's activation modules (torch.nn.LeakyReLU() / torch.nn.Sigmoid() / None)."""
    if activation is None:
        return ACT_NONE
    if isinstance(activation, torch.nn.LeakyReLU):
        if abs(activation.negative_slope - 0.01) > 1e-12:
            raise ValueError("only the default LeakyReLU slope 0.01 is built into the kernels")
        return ACT_LEAKY_RELU
    if isinstance(activation, torch.nn.Sigmoid):
        return ACT_SIGMOID
    raise ValueError(f"unsupported activation {activation!r} (kernels provide None / LeakyReLU() / Sigmoid())")


# ----------------------------------------------------------------------------------------------
# graph preparation
# ----------------------------------------------------------------------------------------------
@dataclass
class PreparedGraph:
    """CSR-by-destination of both edge sets + edge-type ids; device int32 tensors."""
    n_nodes: int
    n_adj_edges: int
    n_col_edges: int          # after GINConv's self-loop removal
    n_types: int
    adj_rowptr: Tensor
    adj_src: Tensor
    adj_eid: Tensor
    adj_type: Tensor          # CSR order
    edge_type: Tensor         # original edge order
    type_rep_edge: Tensor
    col_rowptr: Tensor
    col_src: Tensor
    col_eid: Tensor
    cols: Optional["NNConvColumns"] = None     # matrix-core NNConv column structure (None: CSR kernel is used)
    max_in_degree: int = 0                     # largest adjacency in-degree (0 = not known: no small-layout kernel)
    mid: Optional["NNConvBatches"] = None      # NNConv batches of the mid-size persistent layer loop (None: general schedule)
    groups: Optional["NNConvGroups"] = None    # NNConv edge groups (layouts of the general schedule; None: the type columns are used)

    def c_struct(self, defer_late_check: bool = False) -> _lib.Graph:
        """defer_late_check: the caller queues its launches first and asks `late_words_failed()` behind them (the mid-size
        batches' verdict is the preparation's LAST result word: waiting for it here would wait for the whole preparation)."""
        if not defer_late_check and "_late_words" in self.__dict__:
            self.late_words_failed()
        if defer_late_check and self.mid is not None and "_late_words" in self.__dict__:
            # [r6] the optimistic struct carries the DEVICE address of the verdict (result word 9): the persistent kernels read
            # it first and leave without a trace when the batches turn out not to fit (tgnn_graph.nn_mid_verdict)
            hit = self.__dict__.get("_c_struct_unverified")
            if hit is None:
                res = self.__dict__["_late_words"][2]
                self.__dict__["_verdict_words"] = res                # (kept alive as long as the struct may be in use)
                hit = self._build_c_struct()
                hit.nn_mid_verdict = res.data_ptr() + 9 * 4
                self.__dict__["_c_struct_unverified"] = hit
            return hit
        hit = self.__dict__.get("_c_struct")          # (the tensors of a prepared graph are never replaced)
        if hit is None:
            hit = self.__dict__["_c_struct"] = self._build_c_struct()
        return hit

    def ensure_columns(self) -> None:
        """A graph prepared with edge groups only gets its type columns too (built on first use, kept): what a forward that
        cannot take the fp16-pair path -- eval-mode BatchNorm, tgnn_set_split_precision(0), the all-reduce shard scheme -- runs its
        NNConv on; without them it would fall through to the scalar CSR kernel (3.8 x slower at 100 000 nodes)."""
        if self.cols is None and self.groups is not None:
            cols = graph_columns(self)
            if cols is not None:
                self.cols = cols
                self.__dict__.pop("_c_struct", None)

    def late_words_failed(self) -> bool:
        """True ONCE if the preparation's last result words say that the mid-size batches do not fit (a tile with more batches
        than the persistent layer loop takes): `mid` is dropped, the C struct rebuilt -- a forward that was queued on the
        optimistic struct ran the persistent kernel on truncated batches (memory-safe, wrong) and has to be repeated."""
        late = self.__dict__.pop("_late_words", None)
        if late is None:
            return False
        late[1].synchronize()
        if int(late[0][9]) == 0:
            return False
        self.mid = None
        self.__dict__.pop("_c_struct", None)
        return True

    def _build_c_struct(self) -> _lib.Graph:
        t, st, gr = self.cols, self.mid, self.groups
        return _lib.Graph(self.n_nodes, self.n_adj_edges, self.n_col_edges, self.n_types,
                          self.adj_rowptr.data_ptr(), self.adj_src.data_ptr(), self.adj_type.data_ptr(),
                          self.type_rep_edge.data_ptr(), self.col_rowptr.data_ptr(), self.col_src.data_ptr(),
                          *((t.tile_col_ptr.data_ptr(), t.col_meta.data_ptr(), t.col_src.data_ptr())
                            if t is not None else (None,) * 3), self.max_in_degree,
                          *((st.tile_nb.data_ptr(), st.ent.data_ptr()) if st is not None else (None,) * 2),
                          *((gr.tile_grp_ptr.data_ptr(), gr.grp.data_ptr()) if gr is not None else (None,) * 2))


@dataclass
class NNConvColumns:
    """Per-16-row tiles of type-sorted source columns (tgnn_nnconv_cols_build, include/tgnn.h)."""
    tile_col_ptr: Tensor      # int32 [ceil(N/16) + 1]
    col_meta: Tensor          # int32 [cap]: type | first << 8 | last << 9 | end-of-tile << 10
    col_src: Tensor           # int32 [16 * cap]: source row, -1 = none; root columns: float bits of max(deg,1)


@dataclass
class NNConvBatches:
    """Per 16-row tile its in-edges packed by edge type for the mid-size persistent layer loop (tgnn_mid_entries_build,
    include/tgnn.h; csrc/forward_mid.hip)."""
    tile_nb: Tensor           # int32 [ceil(N/16)]
    ent: Tensor               # int32 (uint32 bits) [ceil(N/16) * 24 * 36]


def runs_general_schedule(n_nodes: int) -> bool:
    """True when tgnn_forward takes a layout of this size through the launch-per-op schedule (neither persistent kernel)."""
    lo_mid, hi_mid = 4096, int(lib.tgnn_get_mid_layout_limit())
    if n_nodes <= lo_mid:
        return n_nodes > int(lib.tgnn_get_small_layout_limit())
    return n_nodes > hi_mid


def mid_layout_range() -> Tuple[int, int]:
    """(lo, hi]: node counts whose layer loop runs as the mid-size persistent kernel (layouts of up to 4 096 nodes belong to the
    small-layout kernel, or to the general schedule when that one is switched off)."""
    return 4096, int(lib.tgnn_get_mid_layout_limit())


def build_nnconv_batches(n_nodes: int, cols: "NNConvColumns") -> Optional[NNConvBatches]:
    """The batches from the column structure (synchronises: a tile that does not fit sends the layout to the general schedule)."""
    dev = cols.tile_col_ptr.device
    ntiles = (n_nodes + 15) // 16
    mid = NNConvBatches(torch.empty(ntiles, dtype=torch.int32, device=dev),
                        torch.empty(int(lib.tgnn_mid_entries_words(n_nodes)), dtype=torch.int32, device=dev))
    res = torch.zeros(2, dtype=torch.int32, device=dev)
    check(lib.tgnn_mid_entries_build(ptr(cols.tile_col_ptr), ptr(cols.col_meta), ptr(cols.col_src), n_nodes, None, ptr(mid.tile_nb),
                                     ptr(mid.ent), ptr(res), _stream(cols.tile_col_ptr)))
    return mid if res.cpu().tolist()[1] == 0 else None


def build_nnconv_columns(n_nodes: int, n_edges: int, n_types: int, rowptr: Tensor, col_src: Tensor,
                         col_type: Tensor) -> Optional[NNConvColumns]:
    """None when the layout has more edge types than the matrix-core kernel's LDS weight image holds."""
    if n_types > lib.tgnn_nnconv_cols_max_types():
        return None
    dev = rowptr.device
    cap = int(lib.tgnn_nnconv_cols_max_columns(n_nodes, n_edges))
    ntiles = (n_nodes + 15) // 16
    cols = NNConvColumns(torch.empty(ntiles + 1, dtype=torch.int32, device=dev),
                         torch.empty(cap, dtype=torch.int32, device=dev),
                         torch.empty(cap * 16, dtype=torch.int32, device=dev))
    ws_bytes = lib.tgnn_nnconv_cols_workspace_bytes(n_nodes)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.tgnn_nnconv_cols_build(ptr(rowptr), ptr(col_src), ptr(col_type), n_nodes, n_types,
                                     ptr(cols.tile_col_ptr), ptr(cols.col_meta), ptr(cols.col_src),
                                     ptr(ws), ws_bytes, _stream(rowptr)))
    return cols


@dataclass
class NNConvGroups:
    """Per-16-row tiles of edge groups: up to 16 in-edges of one type per group (tgnn_nnconv_eg_build, include/tgnn.h)."""
    tile_grp_ptr: Tensor      # int32 [ceil(N/16) + 1]
    grp: Tensor               # int32 [16 * cap, 2]: (source row of slot k, -1 = none; root groups: float bits of max(deg,1) |
                              #                      word j: mask of the slots that end in row j | (type | root << 8) << 16)


def build_nnconv_groups(n_nodes: int, n_edges: int, n_types: int, rowptr: Tensor, col_src: Tensor,
                        col_type: Tensor) -> Optional[NNConvGroups]:
    """None when the layout has more edge types than the structure takes."""
    if n_types > lib.tgnn_nnconv_cols_max_types() or n_types > 40:
        return None
    dev = rowptr.device
    cap = int(lib.tgnn_nnconv_eg_max_groups(n_nodes, n_edges, n_types))
    ntiles = (n_nodes + 15) // 16
    grp = NNConvGroups(torch.empty(ntiles + 1, dtype=torch.int32, device=dev),
                       torch.empty(cap * 16, 2, dtype=torch.int32, device=dev))
    ws_bytes = lib.tgnn_nnconv_cols_workspace_bytes(n_nodes)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.tgnn_nnconv_eg_build(ptr(rowptr), ptr(col_src), ptr(col_type), n_nodes, n_types,
                                   ptr(grp.tile_grp_ptr), ptr(grp.grp), ptr(ws), ws_bytes, _stream(rowptr)))
    return grp


def _check_edge_index(ei: Tensor, name: str) -> Tensor:
    _need_gpu(ei, name)
    if ei.dtype != torch.int64:
        raise ValueError(f"`{name}` must be int64 (torch .long()), got {ei.dtype}")
    if ei.numel() == 0:
        return ei.reshape(2, 0)
    if ei.dim() != 2 or ei.shape[0] != 2:
        raise ValueError(f"`{name}` must have shape [2, E], got {tuple(ei.shape)}")
    return ei if ei.is_contiguous() else ei.contiguous()


def build_csr(edge_index: Tensor, n_nodes: int, drop_self_loops: bool,
              n_src_nodes: Optional[int] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """-> (rowptr [N+1], col_src [E], col_eid [E], err_flag [1]) int32; asynchronous.
    n_src_nodes (default n_nodes): sources may index halo rows stored behind the N destination rows."""
    ei = _check_edge_index(edge_index, "edge_index")
    dev, e = ei.device, int(ei.shape[1])
    rowptr = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
    col_src = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    col_eid = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    ws_bytes = lib.tgnn_csr_workspace_bytes(n_nodes, e)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.tgnn_csr_build(ptr(ei), e, n_nodes, n_nodes if n_src_nodes is None else n_src_nodes,
                             int(drop_self_loops), ptr(rowptr), ptr(col_src), ptr(col_eid), ptr(err), ptr(ws), ws_bytes,
                             _stream(ei)))
    return rowptr, col_src, col_eid, err


def dedup_edge_types(edge_attr: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (edge_type [E], type_rep_edge [E] (first T valid), n_types [1]) int32; asynchronous."""
    ea = _f32c(edge_attr, "edge_attr")
    if ea.dim() != 2:
        raise ValueError(f"`edge_attr` must be [E, Fe], got {tuple(ea.shape)}")
    e, fe = int(ea.shape[0]), int(ea.shape[1])
    dev = ea.device
    edge_type = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    rep = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    n_types = torch.zeros(1, dtype=torch.int32, device=dev)
    ws_bytes = lib.tgnn_edge_dedup_workspace_bytes(e, fe)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.tgnn_edge_type_dedup(ptr(ea), e, fe, ptr(edge_type), ptr(rep), ptr(n_types), ptr(ws), ws_bytes,
                                   _stream(ea)))
    return edge_type, rep, n_types


# Layouts up to this many nodes would run NNConv on the CSR / LDS-weight-table kernel (one half wave per destination row)
# instead of the type-column matrix-core kernel (one wave per 16-row tile).  0 = never: once the column kernel spreads a
# small layout's tiles one per SIMD (csrc/nnconv_cols.hip: launch_cols_t) it is as fast or faster at every size measured
# -- cached-layout forward on MI355X, columns / CSR-only, ms: labyrinth (1 254 nodes) 0.74 / 0.89, synthetic 300 nodes
# 0.80 / 0.76, 2 500: 0.94 / 0.90, 5 000: 0.90 / 0.93, 10 000: 0.91 / 1.04, 20 000: 1.03 / 1.31, 100 000: 2.26 / 3.77
# (scratch/cols_ab.py).  The knob stays for experiments and for the tests that pin the CSR kernel end to end.
COLS_MIN_NODES = int(os.environ.get("TGNN_COLS_MIN_NODES", "0"))
GROUPS = os.environ.get("TGNN_GROUPS", "1") != "0"     # layouts of the general schedule are prepared with NNConv edge groups


def _small_prep_limits(_cache=[]):
    if not _cache:
        _cache.append((int(lib.tgnn_graph_prep_small_max_nodes()), int(lib.tgnn_graph_prep_small_max_edges())))
    return _cache[0]


def _small_prep_counters(dev, _cache={}):
    """Barrier counters of tgnn_graph_prep_small: zero before the first call, left at zero by every call; one pair per
    (device, stream) so that preparations on different streams cannot meet in them."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _cache:
        _cache[key] = torch.zeros(2, dtype=torch.int32, device=dev)
    return _cache[key]


SMALL_PREP = True      # one library call per layout (up to 4 096 nodes: one launch, tgnn_graph_prep_small); False: the separate calls


_pinned_words = {}


def _pinned_result_words(dev):
    import threading
    key = (threading.get_ident(), dev.index, "early")
    host = _pinned_words.get(key)
    if host is None:
        host = _pinned_words[key] = torch.empty(32, dtype=torch.int32, pin_memory=True)
    return host


def _read_back(res: Tensor):
    """The preparation's 32 result words on the host: an asynchronous copy into a pinned buffer of this thread + one stream
    synchronise (`.cpu()` allocates a pageable tensor and stages the copy: ~10 us more per call)."""
    import threading
    key = (threading.get_ident(), res.device.index)
    host = _pinned_words.get(key)
    if host is None:
        host = _pinned_words[key] = torch.empty(32, dtype=torch.int32, pin_memory=True)
    host.copy_(res, non_blocking=True)
    torch.cuda.current_stream(res.device).synchronize()
    return host.tolist()


def _prepare_graph_fused(n_nodes: int, adj: Tensor, attr: Tensor, col: Tensor, small: bool,
                         n_src_nodes: Optional[int] = None, groups: Optional[bool] = None, after_enqueue=None) -> Optional[PreparedGraph]:
    """prepare_graph as ONE library call + the one sync: `small`: tgnn_graph_prep_small (one launch); else tgnn_graph_prep
    (the launches of the separate calls, queued by the library without a host round trip).  None = fall back."""
    ea, ec = int(adj.shape[1]), int(col.shape[1])
    if attr.dim() != 2 or attr.shape[1] < 1:
        return None
    attr = _f32c(attr, "adj_e_features")
    dev = adj.device
    ntiles = (n_nodes + 15) // 16
    e1, c1 = max(ea, 1), max(ec, 1)
    fe = int(attr.shape[1])
    ws_ints = int(lib.tgnn_graph_prep_small_tmp_ints(n_nodes, ea, ec)) if small else \
        (int(lib.tgnn_graph_prep_workspace_bytes(n_nodes, ea, ec, fe)) + 3) // 4 + 64
    lo_mid, hi_mid = mid_layout_range()
    # layouts of the general schedule carry the NNConv edge groups INSTEAD of the type columns (whoever needs the other
    # structure later builds it: graph_columns / graph_groups)
    # (a shard's layout always runs the general schedule)
    want_eg = (not small and groups) if groups is not None else \
        (not small and GROUPS and (n_src_nodes is not None or runs_general_schedule(n_nodes)))
    both = groups == "both" and not small                      # (tests: tgnn_graph_prep building the two structures in one call)
    want_cols = both or not want_eg
    want_mid = not small and want_cols and n_src_nodes is None and lo_mid < n_nodes <= hi_mid   # (built from the columns)
    mid_words = int(lib.tgnn_mid_entries_words(n_nodes)) if want_mid else 0
    cap = int(lib.tgnn_nnconv_cols_max_columns(n_nodes, ea)) if want_cols else 0
    gcap = int(lib.tgnn_nnconv_eg_max_groups(n_nodes, ea, lib.tgnn_nnconv_cols_max_types())) if want_eg else 0
    # the persistent outputs share ONE long-lived allocation; the scratch (CSR / de-dup tables, scan workspaces) and the result
    # words are tensors of their own, freed after the read-back -- a cached graph does not pin hundreds of MB of scratch
    sizes = [n_nodes + 1, e1, e1, e1, e1, e1, n_nodes + 1, c1, c1, ntiles + 1, cap, cap * 16,
             ntiles if want_mid else 0, mid_words, ntiles + 1 if want_eg else 0, gcap * 32]
    offs, at = [], 0
    for sz in sizes:
        offs.append(at)
        at += (sz + 63) // 64 * 64                     # 256-byte aligned pieces of ONE allocation
    buf = torch.empty(at, dtype=torch.int32, device=dev)
    v = [buf[o:o + sz] for o, sz in zip(offs, sizes)]
    (a_rowptr, a_src, a_eid, adj_type, edge_type, rep, c_rowptr, c_src, c_eid, tile_col_ptr, col_meta, col_slot_src,
     mid_nb, mid_ent, tile_grp_ptr, grp) = v
    res = torch.empty(32, dtype=torch.int32, device=dev)
    tmp = torch.empty(ws_ints, dtype=torch.int32, device=dev)
    head = (ptr(adj), ea, ptr(attr), fe, ptr(col), ec, n_nodes) + (() if small else (n_src_nodes or n_nodes,)) + (ptr(a_rowptr), ptr(a_src), ptr(a_eid), ptr(adj_type), ptr(edge_type),
            ptr(rep), ptr(c_rowptr), ptr(c_src), ptr(c_eid),
            *((ptr(tile_col_ptr), ptr(col_meta), ptr(col_slot_src)) if want_cols else (None,) * 3))
    small_words = None
    if small:
        # [r6] the kernel stores its result words into this thread's pinned buffer, tgnn_graph_prep_wait polls word 31
        small_words = _pinned_result_words(dev) if lib.tgnn_set_prep_words_poll(-1) else None
        check(lib.tgnn_graph_prep_small(*head, ptr(tmp), ptr(res), ptr(_small_prep_counters(dev)),
                                        C.c_void_p(small_words.data_ptr()) if small_words is not None else None, _stream(adj)))
    else:
        mid_args = (ptr(mid_nb), ptr(mid_ent)) if want_mid else (None,) * 2
        eg_args = (ptr(tile_grp_ptr), ptr(grp)) if want_eg else (None,) * 2
        # the words read below are final before the NNConv structure is built: without the mid-size batches (whose result words
        # come last) the library copies them out early and the forward is queued while the structure's launches still run
        early = _pinned_result_words(dev)
        check(lib.tgnn_graph_prep(*head, *mid_args, *eg_args, ptr(tmp), ws_ints * 4, ptr(res),
                                  C.c_void_p(early.data_ptr()) if early is not None else None, _stream(adj)))
    late = None
    if not small and early is not None:
        if want_mid:
            # the mid-size batches' verdict (result[9]: a tile that does not fit sends the layout to the general schedule) comes
            # with the LAST launch: copied behind it, looked at when the graph's C struct is first built (c_struct)
            words = torch.empty(32, dtype=torch.int32, pin_memory=True)
            words.copy_(res, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            late = (words, ev, res)
        if after_enqueue is not None:
            # (the caller's launches that need nothing of the graph -- or, of it, only what is on the device already: the types'
            #  representative edges and the result words, whose word 0 is the type count)
            after_enqueue({"type_rep_edge": rep, "result": res})
            after_enqueue = None
        check(lib.tgnn_graph_prep_wait(_stream(adj)))                                        # the one sync: the copy of the words alone
        host = early.tolist()
        host[5] = host[10] = int(host[0] <= lib.tgnn_nnconv_cols_max_types() and not host[6])
        host[9] = 0                                                              # (optimistic: see `late`)
    else:
        if after_enqueue is not None:
            after_enqueue({"type_rep_edge": rep, "result": res} if small else None)
        if small_words is not None:
            check(lib.tgnn_graph_prep_wait(_stream(adj)))                        # the one sync: a poll of pinned memory
            host = small_words.tolist()
        else:
            host = _read_back(res)                                               # the one sync
    if host[1] or host[2]:
        raise IndexError(f"edge index out of range [0, {n_nodes}) in {'adj_e_index' if host[1] else 'col_e_idx'}")
    if host[6]:
        return None
    cols = NNConvColumns(tile_col_ptr, col_meta, col_slot_src) if host[5] and want_cols else None
    mid = NNConvBatches(mid_nb, mid_ent) if want_mid and cols is not None and host[9] == 0 else None
    groups_ = NNConvGroups(tile_grp_ptr, grp.view(-1, 2)) if want_eg and host[10] else None
    g = PreparedGraph(n_nodes, ea, int(host[3]), int(host[0]), a_rowptr, a_src, a_eid, adj_type, edge_type, rep,
                      c_rowptr, c_src, c_eid, cols, int(host[4]), mid, groups_)
    if late is not None and mid is not None:
        g.__dict__["_late_words"] = late
    return g


def prepare_graph(n_nodes: int, adj_e_index: Tensor, adj_e_features: Tensor, col_e_idx: Tensor,
                  tile_width: int = 32, n_src_nodes: Optional[int] = None, columns: Optional[bool] = None,
                  groups=None, after_enqueue=None) -> PreparedGraph:
    """Everything the 20 layers share: CSR of both edge sets, edge-type ids in CSR order, and (columns: None = for
    layouts above COLS_MIN_NODES) the NNConv column structure -- or (groups: None = for layouts of the general schedule,
    i.e. above the mid-size limit, when GROUPS is on; "both": columns, mid-size batches AND groups from the one
    tgnn_graph_prep call) the NNConv edge groups in its place; `graph_columns` / `graph_groups` build the other structure
    for whoever needs it.
    Synchronises once (the type count and the self-loop-free collision edge count are read back).
    after_enqueue: called once behind the preparation's launches and in front of that synchronisation (or first thing on the paths
    that synchronise more than once): launches of the caller's that need nothing of the graph go there.  Its one argument is None
    or, from the one-call preparation, {"type_rep_edge", "result"}: device tensors that are filled by the launches just queued."""
    done = [False]
    if after_enqueue is not None:
        user_cb = after_enqueue

        def after_enqueue(info=None):
            if not done[0]:
                done[0] = True
                user_cb(info)
    adj = _check_edge_index(adj_e_index, "adj_e_index")
    col = _check_edge_index(col_e_idx, "col_e_idx")
    ea, ec = int(adj.shape[1]), int(col.shape[1])
    if adj_e_features.shape[0] != ea:
        raise ValueError(f"adj_e_features has {adj_e_features.shape[0]} rows for {ea} edges")
    if SMALL_PREP and tile_width == 32 and columns in (None, True) and COLS_MIN_NODES == 0 and n_nodes >= 1:
        # (a shard's layout -- sources behind the destination rows -- goes through the any-size call)
        small = n_src_nodes is None and n_nodes <= _small_prep_limits()[0] and max(ea, ec) <= _small_prep_limits()[1] and \
            not (groups or (groups is None and GROUPS and runs_general_schedule(n_nodes)))   # (the one-launch preparation builds columns)
        g = _prepare_graph_fused(n_nodes, adj, adj_e_features, col, small, n_src_nodes, groups, after_enqueue)
        if g is not None:
            return g
    if after_enqueue is not None:
        after_enqueue()
    a_rowptr, a_src, a_eid, a_err = build_csr(adj, n_nodes, False, n_src_nodes)
    c_rowptr, c_src, c_eid, c_err = build_csr(col, n_nodes, True, n_src_nodes)        # GINConv strips self loops
    edge_type, rep, n_types = dedup_edge_types(adj_e_features)
    adj_type = torch.empty(max(ea, 1), dtype=torch.int32, device=adj.device)
    check(lib.tgnn_gather_i32(ptr(edge_type), ea, ptr(a_eid), ea, ptr(adj_type), _stream(adj)))
    # (measured at 100k nodes: the collision CSR on a stream of its own beside the adjacency side changes nothing, 0.425 vs
    #  0.427 ms -- the ~38 launches of a preparation are bound by the host's launch path and by returning atomics, not by
    #  idle CUs)
    # largest adjacency in-degree: the small-layout kernel's limit and the bound of the fp16-pair NNConv operands
    if n_nodes > 0:
        max_deg = (a_rowptr[1:n_nodes + 1] - a_rowptr[:n_nodes]).max().reshape(1).to(torch.int32)
    else:
        max_deg = torch.zeros_like(n_types)
    host = torch.cat([n_types, a_err, c_err, c_rowptr[n_nodes:n_nodes + 1], max_deg]).cpu().tolist()   # the one sync
    if host[1] or host[2]:
        raise IndexError(f"edge index out of range [0, {n_nodes}) in "
                         f"{'adj_e_index' if host[1] else 'col_e_idx'}")
    n_types = int(host[0])
    if columns is None:
        columns = n_nodes > COLS_MIN_NODES
    lo_mid, hi_mid = mid_layout_range()
    want_eg = tile_width == 32 and (groups if groups is not None else
                                    (columns and GROUPS and (n_src_nodes is not None or runs_general_schedule(n_nodes))))
    grp = build_nnconv_groups(n_nodes, ea, n_types, a_rowptr, a_src, adj_type) if want_eg else None
    cols = build_nnconv_columns(n_nodes, ea, n_types, a_rowptr, a_src, adj_type) if tile_width == 32 and columns and grp is None else None
    mid = None
    if cols is not None and n_src_nodes is None and lo_mid < n_nodes <= hi_mid:
        mid = build_nnconv_batches(n_nodes, cols)
    return PreparedGraph(n_nodes, ea, int(host[3]), n_types, a_rowptr, a_src, a_eid, adj_type, edge_type, rep,
                         c_rowptr, c_src, c_eid, cols, int(host[4]), mid, grp)


def graph_columns(graph: PreparedGraph) -> Optional[NNConvColumns]:
    """The graph's type columns, built on first use when the preparation left them out (layouts that carry edge groups)."""
    if graph.cols is not None:
        return graph.cols
    hit = graph.__dict__.get("_lazy_cols", False)
    if hit is False:
        hit = graph.__dict__["_lazy_cols"] = build_nnconv_columns(graph.n_nodes, graph.n_adj_edges, graph.n_types, graph.adj_rowptr,
                                                                  graph.adj_src, graph.adj_type)
    return hit


def graph_groups(graph: PreparedGraph) -> Optional[NNConvGroups]:
    """The graph's edge groups, built on first use when the preparation left them out."""
    if graph.groups is not None:
        return graph.groups
    hit = graph.__dict__.get("_lazy_groups", False)
    if hit is False:
        hit = graph.__dict__["_lazy_groups"] = build_nnconv_groups(graph.n_nodes, graph.n_adj_edges, graph.n_types, graph.adj_rowptr,
                                                                   graph.adj_src, graph.adj_type)
    return hit


# ----------------------------------------------------------------------------------------------
# per-op wrappers (layer seams of the reference)
# ----------------------------------------------------------------------------------------------
def new_partials(f: int, device) -> Tensor:
    return torch.empty(BN_MAX_PARTIALS * 2 * f, dtype=torch.float64, device=device)


def edge_weight_table(edge_attr: Tensor, graph: PreparedGraph, w1, b1, w2, b2, w3, b3, c: int) -> Tensor:
    """[T, C, C] NNConv matrices: GraphConv's edge MLP on the T distinct attribute rows."""
    ea = _f32c(edge_attr, "edge_attr")
    t, fe = graph.n_types, int(ea.shape[1])
    if tuple(w1.shape) != (32, fe) or tuple(w2.shape) != (64, 32) or tuple(w3.shape) != (c * c, 64):
        raise ValueError(f"edge-MLP shapes {tuple(w1.shape)} {tuple(w2.shape)} {tuple(w3.shape)} do not match "
                         f"Fe={fe}, C={c}")
    wtab = torch.empty(max(t, 1), c, c, dtype=torch.float32, device=ea.device)
    ws = [_f32c(p, "edge mlp parameter") for p in (w1, b1, w2, b2, w3, b3)]
    check(lib.tgnn_edge_weight_table(ptr(ea), ptr(graph.type_rep_edge), t, fe, *[ptr(p) for p in ws], c, ptr(wtab),
                                     _stream(ea)))
    return wtab[:t]


def nnconv_mean(h: Tensor, graph: PreparedGraph, wtab: Tensor, root: Tensor, bias: Tensor, act: int = ACT_NONE,
                partials: Optional[Tensor] = None, force_csr_kernel: bool = False,
                kernel: Optional[str] = None, max_in_degree: int = 0) -> Tuple[Tensor, int]:
    """NNConv mean: the matrix-core column kernel when the graph carries the column structure, else the CSR /
    LDS-weight-table kernel (any type count that fits LDS) or the generic one.  kernel: None = that order; "cols_f16" = the
    column kernel with the fp16 x 2 split as tgnn_forward runs it (max_in_degree: the bound to scale by, default the
    layout's)."""
    h = _f32c(h, "x")
    c = int(h.shape[1])
    n = graph.n_nodes                      # destination rows; x may carry extra (halo) rows behind them
    if int(h.shape[0]) < n:
        raise ValueError(f"x has {int(h.shape[0])} rows, the graph {graph.n_nodes} nodes")
    if tuple(root.shape) != (c, c) or tuple(bias.shape) != (c,):
        raise ValueError("NNConv root/bias shape mismatch")
    out = torch.empty(n, c, dtype=torch.float32, device=h.device)
    npart = C.c_int32(0)
    wt = _f32c(wtab, "wtab")
    if kernel is None and graph.groups is not None and c == 32 and not force_csr_kernel and 1 <= graph.max_in_degree <= 2048 \
            and int(h.shape[0]) * c * 4 < 2 ** 31:
        kernel = "eg"                          # a layout that carries edge groups: the kernel tgnn_forward runs on it
    if kernel == "eg":
        grp = graph_groups(graph)
        if grp is None or c != 32:
            raise ValueError("the edge-group kernel needs <= 40 edge types and width 32")
        wimg = torch.empty(lib.tgnn_nnconv_weight_image_floats(graph.n_types), dtype=torch.float32, device=h.device)
        bounds = torch.empty(2, dtype=torch.int32, device=h.device)
        check(lib.tgnn_nnconv_mean_eg_fwd(ptr(h), c, int(h.shape[0]), ptr(grp.tile_grp_ptr), ptr(grp.grp), ptr(wt), graph.n_types, ptr(_f32c(root, "root")), ptr(_f32c(bias, "bias")), n, act,
                                          ptr(out), ptr(wimg), ptr(bounds), ptr(partials), C.byref(npart), _stream(h)))
        return out, npart.value
    tl = graph_columns(graph) if (kernel == "cols_f16" or (graph.groups is not None and not force_csr_kernel)) else graph.cols
    if kernel == "cols_f16":
        if tl is None or c != 32:
            raise ValueError("the fp16-pair column kernel needs the column structure and width 32")
        wimg = torch.empty(lib.tgnn_nnconv_weight_image_floats(graph.n_types), dtype=torch.float32, device=h.device)
        bounds = torch.empty(2, dtype=torch.int32, device=h.device)
        check(lib.tgnn_nnconv_mean_cols_f16_fwd(ptr(h), c, int(h.shape[0]), ptr(tl.tile_col_ptr), ptr(tl.col_meta), ptr(tl.col_src),
                                                ptr(wt), graph.n_types, ptr(_f32c(root, "root")), ptr(_f32c(bias, "bias")), n,
                                                max_in_degree if max_in_degree else graph.max_in_degree, act, ptr(out), ptr(wimg),
                                                ptr(bounds), ptr(partials), C.byref(npart), _stream(h)))
    elif tl is not None and c == 32 and not force_csr_kernel and int(h.shape[0]) * c * 4 < 2 ** 31:
        wimg = torch.empty(lib.tgnn_nnconv_weight_image_floats(graph.n_types), dtype=torch.float32, device=h.device)
        check(lib.tgnn_nnconv_mean_cols_fwd(ptr(h), c, ptr(tl.tile_col_ptr), ptr(tl.col_meta), ptr(tl.col_src), ptr(wt),
                                            graph.n_types, ptr(_f32c(root, "root")), ptr(_f32c(bias, "bias")), n, c, act,
                                            ptr(out), ptr(wimg), ptr(partials), C.byref(npart), _stream(h)))
    else:
        check(lib.tgnn_nnconv_mean_fwd(ptr(h), c, ptr(graph.adj_rowptr), ptr(graph.adj_src), ptr(graph.adj_type), ptr(wt),
                                       graph.n_types, ptr(_f32c(root, "root")), ptr(_f32c(bias, "bias")), n, c, act,
                                       ptr(out), ptr(partials), C.byref(npart), _stream(h)))
    return out, npart.value


def gin(a: Tensor, graph: PreparedGraph, eps: Tensor, w1, b1, w2, b2, w3, b3, act: int = ACT_NONE,
        in_stat: Optional[Tensor] = None, partials: Optional[Tensor] = None) -> Tuple[Tensor, int]:
    a = _f32c(a, "x")
    c = int(a.shape[1])
    n = graph.n_nodes                      # destination rows; x may carry extra (halo) rows behind them
    if int(a.shape[0]) < n:
        raise ValueError(f"x has {int(a.shape[0])} rows, the graph {graph.n_nodes} nodes")
    if tuple(w1.shape) != (32, c) or tuple(w2.shape) != (64, 32) or tuple(w3.shape) != (c, 64):
        raise ValueError("GIN MLP shape mismatch")
    out = torch.empty(n, c, dtype=torch.float32, device=a.device)
    z_scratch = torch.empty(n, c, dtype=torch.float32, device=a.device)
    npart = C.c_int32(0)
    ps = [_f32c(p, "gin parameter") for p in (eps, w1, b1, w2, b2, w3, b3)]
    check(lib.tgnn_gin_fwd(ptr(a), c, ptr(in_stat), ptr(graph.col_rowptr), ptr(graph.col_src), *[ptr(p) for p in ps],
                           n, c, act, ptr(out), ptr(z_scratch), ptr(partials), C.byref(npart), _stream(a)))
    return out, npart.value


def dense_act(a: Tensor, weight: Tensor, bias: Tensor, act: int, in_stat: Optional[Tensor] = None,
              partials: Optional[Tensor] = None, slot_major: bool = False, f16_split: bool = False) -> Tuple[Tensor, int]:
    """act(BN_in(a) @ weight.T + bias) for a row-major [N, in] matrix, or (slot_major) for the
    [S, N, C] skip-connection buffer read as the concatenation of its S slots (torch.cat never happens; C a
    multiple of 32).  f16_split (slot-major, C = 32, no input BatchNorm, >= 64 outputs): the fp16 x 2 split kernel as
    tgnn_forward runs it."""
    a = _f32c(a, "x")
    if slot_major:
        if a.dim() != 3 or a.shape[2] % 32 != 0:
            raise ValueError(f"slot-major input must be [S, N, C] with C a multiple of 32, got {tuple(a.shape)}")
        n, cw = int(a.shape[1]), int(a.shape[2])
        k = cw * int(a.shape[0])
        m = int(weight.shape[0])
        if int(weight.shape[1]) != k:
            raise ValueError(f"Linear expects in_dim {int(weight.shape[1])}, got {k}")      # layers/util.py:16
        out = torch.empty(n, m, dtype=torch.float32, device=a.device)
        npart = C.c_int32(0)
        if f16_split:
            if in_stat is not None or cw != 32:
                raise ValueError("the fp16-pair dense kernel takes 32-channel slots and no input BatchNorm")
            bounds = torch.empty(int(a.shape[0]) + 1, dtype=torch.int32, device=a.device)
            # the rows-per-wave kernel (what tgnn_forward runs on the final MLP) where its shape fits; f16_split="tile": the block-tile one
            # (the library takes the rows kernel from 49 152 rows on; below that wimg is built and ignored)
            wimg = torch.empty(k * m, dtype=torch.float32, device=a.device) if (f16_split != "tile" and m in (64, 128, 256)) else None
            check(lib.tgnn_dense_act_slots_f16_fwd(ptr(a), cw, n * cw, ptr(_f32c(weight, "weight")), ptr(_f32c(bias, "bias")), n, k, m,
                                                   act, ptr(out), m, ptr(bounds), ptr(wimg), ptr(partials), C.byref(npart), _stream(a)))
            return out, npart.value
        check(lib.tgnn_dense_act_slots_fwd(ptr(a), cw, n * cw, ptr(in_stat), ptr(_f32c(weight, "weight")),
                                           ptr(_f32c(bias, "bias")), n, k, m, act, ptr(out), m, ptr(partials),
                                           C.byref(npart), _stream(a)))
        return out, npart.value
    else:
        if a.dim() != 2:
            raise ValueError(f"expected a [N, F] matrix, got {tuple(a.shape)}")
        n, k = int(a.shape[0]), int(a.shape[1])
        lda, kb = k, 32
    m = int(weight.shape[0])
    if int(weight.shape[1]) != k:
        raise ValueError(f"Linear expects in_dim {int(weight.shape[1])}, got {k}")      # layers/util.py:16
    out = torch.empty(n, m, dtype=torch.float32, device=a.device)
    npart = C.c_int32(0)
    check(lib.tgnn_dense_act_fwd(ptr(a), lda, kb, ptr(in_stat), ptr(_f32c(weight, "weight")), ptr(_f32c(bias, "bias")),
                                 n, k, m, act, ptr(out), m, ptr(partials), C.byref(npart), _stream(a)))
    return out, npart.value


def bn_finalize(partials: Tensor, n_partials: int, n_rows: int, bn: torch.nn.BatchNorm1d, update_running: bool,
                mode: int = 0, sums: Optional[Tensor] = None) -> Optional[Tensor]:
    """Train-mode statistics -> stat record [4, F]  (mode 1 returns None and fills `sums`)."""
    f = bn.num_features
    dev = bn.weight.device
    stat = torch.empty(4, f, dtype=torch.float32, device=dev) if mode != 1 else None
    upd = update_running and bn.track_running_stats and mode != 1
    check(lib.tgnn_bn_finalize(mode, ptr(partials), n_partials, ptr(sums), f, n_rows, ptr(bn.weight), ptr(bn.bias),
                               float(bn.eps), float(bn.momentum if bn.momentum is not None else 0.1),
                               ptr(bn.running_mean if upd or mode == 3 else None),
                               ptr(bn.running_var if upd or mode == 3 else None),
                               ptr(bn.num_batches_tracked if upd else None), ptr(stat), _stream(bn.weight)))
    return stat


def bn_apply(v: Tensor, stat: Tensor) -> Tensor:
    v = _f32c(v, "x")
    n, f = int(v.shape[0]), int(v.shape[1])
    out = torch.empty_like(v)
    check(lib.tgnn_bn_apply(ptr(v), f, ptr(stat), n, f, ptr(out), f, _stream(v)))
    return out


def batch_norm(v: Tensor, partials: Tensor, n_partials: int, bn: torch.nn.BatchNorm1d) -> Tensor:
    """nn.BatchNorm1d forward on a [N, F] activation whose column sums are in `partials`:
    batch statistics (and running-stat update) in train mode, running statistics in eval mode."""
    n = int(v.shape[0])
    if bn.training or not bn.track_running_stats:
        if n < 2:
            raise ValueError("Expected more than 1 value per channel when training")     # torch's own message
        stat = bn_finalize(partials, n_partials, n, bn, update_running=True, mode=0)
    else:
        stat = bn_finalize(partials, max(n_partials, 1), max(n, 1), bn, update_running=False, mode=3)
    return bn_apply(v, stat)


def merge(a1: Tensor, stat1: Tensor, a2: Tensor, stat2: Tensor, resid: Optional[Tensor],
          out: Optional[Tensor] = None, want_h2: bool = True) -> Tuple[Tensor, Optional[Tensor]]:
    """BN1(a1) * BN2(a2) (+ resid) -> out (a dense [N, C] tensor, e.g. a slot of the skip buffer)."""
    n, c = int(a1.shape[0]), int(a1.shape[1])
    if out is None:
        out = torch.empty_like(a1)
    elif not out.is_contiguous() or tuple(out.shape) != (n, c):
        raise ValueError("merge: `out` must be a contiguous [N, C] tensor")
    h2 = torch.empty_like(a1) if want_h2 else None
    check(lib.tgnn_merge_fwd(ptr(a1), ptr(stat1), ptr(a2), ptr(stat2), ptr(resid), n, c, ptr(out), ptr(h2),
                             _stream(a1)))
    return out, h2


def rows_gather(src: Tensor, idx: Tensor, out: Tensor, col_offset: int = 0) -> None:
    """out[i, col_offset : col_offset + C] = src[idx[i], :C]  (halo send packing)."""
    c = int(src.shape[1])
    check(lib.tgnn_rows_gather(ptr(src), int(src.stride(0)), ptr(idx), int(idx.shape[0]), c,
                               C.c_void_p(out.data_ptr() + 4 * col_offset), int(out.stride(0)), _stream(src)))


def bn_sums(partials: Tensor, n_partials: int, f: int) -> Tensor:
    """Per-block partial rows -> [2, F] fp64 column sums (the quantity ranks all-reduce)."""
    sums = torch.empty(2 * f, dtype=torch.float64, device=partials.device)
    check(lib.tgnn_bn_finalize(1, ptr(partials), n_partials, ptr(sums), f, 1, None, None, 1e-5, 0.1, None, None, None,
                               None, _stream(partials)))
    return sums


def bn_stat_from_sums(sums: Tensor, n_rows_total: int, bn: torch.nn.BatchNorm1d, update_running: bool) -> Tensor:
    """[2, F] global column sums -> stat record [4, F] (+ running-stat update)."""
    f = bn.num_features
    stat = torch.empty(4, f, dtype=torch.float32, device=sums.device)
    upd = update_running and bn.track_running_stats
    check(lib.tgnn_bn_finalize(2, None, 0, ptr(sums), f, n_rows_total, ptr(bn.weight), ptr(bn.bias), float(bn.eps),
                               float(bn.momentum if bn.momentum is not None else 0.1),
                               ptr(bn.running_mean if upd else None), ptr(bn.running_var if upd else None),
                               ptr(bn.num_batches_tracked if upd else None), ptr(stat), _stream(sums)))
    return stat
