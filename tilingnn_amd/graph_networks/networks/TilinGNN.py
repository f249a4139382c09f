"""`TilinGNN` -- the scoring network, same constructor / forward / state-dict as the reference
(/root/reference/graph_networks/networks/TilinGNN.py:13-78), executed by ONE C-ABI call
(`tgnn_forward`, csrc/forward.hip) into hand-written gfx950 kernels.

    network = TilinGNN(adj_edge_features_dim=..., network_depth=20, network_width=32).to("cuda")
    probs, *_ = network(x=x, adj_e_index=ei, adj_e_features=ea, col_e_idx=ci, col_e_features=cf)

drops into ML_Solver.predict (solver/ml_solver/ml_solver.py:39-43) unchanged.  Differences a
caller can observe: outputs carry no autograd graph unless `network.autograd = True` (the training step of
tilingnn_amd/train.py; off by default because the reference runs inference with autograd
recording too), and a CPU module raises instead of computing (no fallback by design).
BatchNorm follows module.training exactly like torch: batch statistics + running-stat updates in
train mode (which is how the reference runs inference, ml_solver.py:129-131), running statistics
in eval mode.
"""
import ctypes as C

import torch
import torch.nn as nn

from ... import _lib, ops
from ..._lib import check, lib, ptr
from .. import _graph_cache, _tracking
from .._tracking import Tracked
from ..layers.coll_conv import CollConv
from ..layers.edge_conv import GraphConv
from ..layers.util import MLP, Linear_trans


import os as _os
# tgnn_forward_begin's launches in front of the preparation's (default; same-box A/B at 100 000 nodes: 1.921 against 1.937 ms per step
# with them behind the preparation's launches -- ops.prepare_graph's after_enqueue hook --, profiles/r06_begin_order.txt)
_BEGIN_FIRST = _os.environ.get("TGNN_BEGIN_FIRST", "1") == "1"


def _default_node_features_dim():
    """The reference evaluates `environment.tile_count + 1` at import (TilinGNN.py:19).  When this
    module is dropped into the reference tree the same global is honoured."""
    try:
        from inputs.config import environment          # type: ignore
        return environment.tile_count + 1
    except Exception as exc:                            # noqa: BLE001
        raise ValueError("node_features_dim not given and inputs.config.environment is not importable; "
                         "pass node_features_dim=tile_count+1") from exc


class TilinGNN(Tracked, nn.Module):
    def __init__(self, adj_edge_features_dim, network_depth, network_width, output_dim=1, node_features_dim=None):
        super().__init__()
        if node_features_dim is None:
            node_features_dim = _default_node_features_dim()
        self.network_depth = network_depth
        self.network_width = network_width
        self.residual_skip_num = 2                                           # TilinGNN.py:25
        self.adj_edge_features_dim = adj_edge_features_dim
        self.node_features_dim = node_features_dim
        self.output_dim = output_dim
        self.brch_1_layer_feature_dims = [network_width] * (network_depth + 1)
        self.brch_2_layer_feature_dims = [network_width] * (network_depth + 1)

        self.init_node_feature_trans = MLP(in_dim=node_features_dim, out_dim=network_width,
                                           hidden_layer_dims=[network_width], activation=torch.nn.LeakyReLU(),
                                           batch_norm=True)
        self.brch_1_graph_conv_layers = nn.ModuleList(
            [GraphConv(edge_feature_dim=adj_edge_features_dim, node_feature_in_dim=network_width,
                       node_feature_out_dim=network_width) for _ in range(network_depth)])
        self.brch_2_coll_conv_layers = nn.ModuleList(
            [CollConv(node_feature_in_dim=network_width, node_feature_out_dim=network_width)
             for _ in range(network_depth)])
        self.final_mlp = nn.Sequential(
            MLP(in_dim=network_width * (network_depth + 1), out_dim=network_width, hidden_layer_dims=[256, 128, 64],
                activation=torch.nn.LeakyReLU()),
            Linear_trans(network_width, output_dim, activation=torch.nn.Sigmoid(), batch_norm=False))
        self.cache_graph = True          # reuse the prepared CSR / edge types while the edge tensors are unchanged
        # The reference runs inference with autograd recording (ml_solver.py:39 has no no_grad), so "grad enabled" cannot
        # mean "training" here.  The differentiable forward (tilingnn_amd/train.py: keeps activations, backward through
        # the adjoint kernels) is therefore opt-in: Trainer switches it on around its steps.
        self.autograd = False
        # torch.bfloat16: the activations that cross HBM between kernels are stored in bf16 (fp32 accumulation, fp64
        # BatchNorm sums) -- BASELINE config 3, network_width 64 only (tilingnn_amd/ops_bf16.py, csrc/bf16_path.hip)
        self.activation_dtype = torch.float32

    # ---- host-side table of device pointers -------------------------------------------------
    def _dims(self):
        key = (self.node_features_dim, self.adj_edge_features_dim, self.network_width, self.network_depth, self.output_dim)
        hit = self.__dict__.get("_tgnn_dims")
        if hit is None or hit[0] != key:
            hit = self.__dict__["_tgnn_dims"] = (key, _lib.ModelDims(*key))
        return hit[1]

    def _param_table(self, verify: bool = True):
        """(table, device).  verify=False: the cached table without the pointer comparison (~25 us for 400 entries) -- the
        caller runs `_param_table_stale()` AFTER it has queued its work and repeats the work if that says so; the cache holds
        the storages the table points into, so a stale pointer still points into live memory."""
        cache = self.__dict__.get("_tgnn_table")
        if cache is not None and cache[0] == _tracking.epoch():
            # the epoch covers attribute assignment, _apply and load_state_dict; storage swaps underneath a live
            # Parameter (`p.data = t`, torch.utils.swap_tensors) show up as a changed data_ptr
            if verify:
                if tuple(map(torch.Tensor.data_ptr, cache[3])) == cache[4]:
                    return cache[1], cache[2]
            elif tuple(map(torch.Tensor.data_ptr, cache[6])) == cache[7]:
                # (what the kernels WRITE -- the running statistics -- is compared up front even then: a stale read comes out of
                #  memory the cache keeps alive and is repeated, a stale write would land in a buffer somebody else may own)
                return cache[1], cache[2]
        dims = self._dims()
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        names = _lib.param_names(dims)
        table = (C.c_void_p * len(names))()
        keep = []
        dev = None
        for i, name in enumerate(names):
            t = sd[name]
            if not t.is_cuda:
                raise RuntimeError(f"tilingnn_amd.TilinGNN: parameter `{name}` is on {t.device}; call .to('cuda') "
                                   "(there is no CPU path)")
            want = torch.int64 if name.endswith("num_batches_tracked") else torch.float32
            if t.dtype != want:
                raise ValueError(f"parameter `{name}` must be {want}, got {t.dtype}")
            if not t.is_contiguous():
                raise ValueError(f"parameter `{name}` must be contiguous")
            dev = dev or t.device
            if t.device != dev:
                raise ValueError("all parameters must live on one device")
            table[i] = t.data_ptr()
            keep.append(t)
        written = [t for name, t in zip(names, keep) if name.endswith(("running_mean", "running_var", "num_batches_tracked"))]
        self.__dict__["_tgnn_table"] = (_tracking.epoch(), table, dev, keep, tuple(t.data_ptr() for t in keep),
                                        [t.untyped_storage() for t in keep], written, tuple(t.data_ptr() for t in written))
        return table, dev

    def _param_table_stale(self) -> bool:
        """True when a parameter's storage was swapped underneath the cached table (the cache is dropped)."""
        cache = self.__dict__.get("_tgnn_table")
        if cache is None or cache[0] != _tracking.epoch() or tuple(map(torch.Tensor.data_ptr, cache[3])) != cache[4]:
            self.__dict__.pop("_tgnn_table", None)
            return True
        return False

    def __getstate__(self):                      # copy.deepcopy(network) (ml_solver.py:26) and pickling
        state = self.__dict__.copy()
        state.pop("_tgnn_table", None)
        state.pop("_tgnn_dims", None)
        return state

    # ---- forward ----------------------------------------------------------------------------
    def forward_many(self, layouts, streams: int = 3):
        """K independent layouts, every one a batch of its own (own BatchNorm statistics: exactly what K forward() calls
        compute, bit for bit) scored SIDE BY SIDE -- the reference's crop loop (Tiling-Shape.py:52-64) and the first rounds of
        several greedy solves hand over layouts of ~1 000 nodes that fill a third of the chip each.  layouts: sequence of
        (x, adj_e_index, adj_e_features, col_e_idx); every layout's preparation and forward are queued on one of `streams`
        streams of this module's own, the persistent small-layout kernels of different streams run beside each other when
        they fit the device together (csrc/forward_small.hip: spin_kernel_chain).  More than two layouts at a time need more
        than HIP's default 4 hardware queues (current + side + K layout streams): GPU_MAX_HW_QUEUES=8 in the environment before
        the process's first GPU call; the streams used are measured to overlap (_lib.concurrent_streams).  Train mode: the running statistics are left
        untouched (K concurrent updates of the same buffers would race; they do not enter train-mode outputs).
        Returns the list of probs tensors, ready on the current stream."""
        table, dev = self._param_table()
        layouts = list(layouts)
        bn_train = self.training
        if (self.autograd and bn_train and torch.is_grad_enabled()) or self.activation_dtype != torch.float32 or not layouts:
            return [self._forward_one(*l, update_running=False)[0] for l in layouts]
        cur = torch.cuda.current_stream(dev)
        _lib.side_stream(dev)                                  # (the first of the measured set: the forward's side stream)
        lanes = _lib.concurrent_streams(dev, streams + 1)[1:]    # streams measured to run beside each other and the side stream
        k_n = len(layouts)
        used = lanes[:min(streams, k_n)]
        for st in used:
            st.wait_stream(cur)
        dims = self._dims()
        xs, attrs, graphs, outs, wss = [], [], [], [], []
        for k, (x, adj, attr, col) in enumerate(layouts):
            n = int(x.shape[0])
            if x.dim() != 2 or x.shape[1] != self.node_features_dim or attr.dim() != 2 or attr.shape[1] != self.adj_edge_features_dim:
                raise ValueError("forward_many: layout shapes (see forward)")
            if bn_train and n < 2:
                raise ValueError("Expected more than 1 value per channel when training")
            with torch.cuda.stream(lanes[k % streams]):          # preparation and buffers belong to the layout's stream
                xf, ea = ops._f32c(x, "x"), ops._f32c(attr, "adj_e_features")
                graph = _graph_cache.get_full(n, adj, attr, col) if self.cache_graph else ops.prepare_graph(n, adj, attr, col)
                ws = torch.empty(lib.tgnn_forward_workspace_bytes(C.byref(dims), n, graph.n_types), dtype=torch.uint8, device=dev)
                probs = torch.empty(n, self.output_dim, dtype=torch.float32, device=dev)
            probs.record_stream(cur)
            xs.append(xf); attrs.append(ea); graphs.append(graph); outs.append(probs); wss.append(ws)
        arr = lambda ts: (C.c_void_p * k_n)(*[t.data_ptr() for t in ts])
        gs = (_lib.Graph * k_n)(*[g.c_struct() for g in graphs])
        check(lib.tgnn_forward_many(C.byref(dims), table, k_n, arr(xs), arr(attrs), gs, 0, int(not bn_train), arr(outs), arr(wss),
                                    (C.c_size_t * k_n)(*[int(w.numel()) for w in wss]),
                                    (C.c_void_p * len(used))(*[st.cuda_stream for st in used]), len(used), _lib.side_stream(dev)))
        for st in used:
            cur.wait_stream(st)
        return outs

    def forward(self, x, adj_e_index, adj_e_features, col_e_idx, col_e_features=None):
        return self._forward_one(x, adj_e_index, adj_e_features, col_e_idx, col_e_features)

    def forward_checked(self, x, adj_e_index, adj_e_features, col_e_idx, col_e_features=None):
        """forward() + the health check of the persistent kernels (layouts of up to 65 536 nodes run their layers as ONE kernel
        whose blocks wait for each other: csrc/forward_small.hip, forward_mid.hip).  Every such wait is bounded; a kernel
        that was starved of compute units by another process gives up and leaves a word behind.  This call SYNCHRONISES the
        current stream, reads that word and, if it is set, runs the forward again on the general launch schedule (which the next
        256 forwards of the process take too: tgnn_persist_fallback).  ML_Solver.predict goes through here (it copies the probabilities to the
        host right away, so the synchronisation costs it nothing); forward() itself stays asynchronous."""
        try:
            out = self._forward_one(x, adj_e_index, adj_e_features, col_e_idx, col_e_features)
        except _lib.TgnnError as exc:
            # an EARLIER, unchecked forward's persistent kernel gave up and this entry was the first to see it: the library has cleared
            # the word and opened the fall-back window -- this call's own forward was not queued; queue it now, once
            if exc.code != _lib.ERR_STALE_RESULT:
                raise
            out = self._forward_one(x, adj_e_index, adj_e_features, col_e_idx, col_e_features)
        dev = x.device
        code = C.c_uint32(0)
        check(lib.tgnn_spin_error_poll(_lib.current_stream(dev), C.byref(code)))
        if code.value:
            import warnings
            # (the poll has opened the fallback window: the next 256 forwards of the process take the general schedule, then the
            #  persistent ones are tried again -- a transient neighbour does not cost every later predict 30-50 %)
            warnings.warn("tilingnn_amd: a persistent forward kernel gave up waiting for its blocks (another process holds compute "
                          f"units; reason bits {code.value}); the forward is repeated on the general launch schedule, which the "
                          "next forwards of this process take as well", RuntimeWarning)
            out = self._forward_one(x, adj_e_index, adj_e_features, col_e_idx, col_e_features)
        return out

    def _forward_one(self, x, adj_e_index, adj_e_features, col_e_idx, col_e_features=None, update_running=True):
        fast = (self.activation_dtype == torch.float32 and not (self.autograd and self.training and torch.is_grad_enabled()))
        # (the inference forward checks the cached pointer table AFTER it has queued its kernels: 25 us less in front of the
        #  first launch; a stale table -- `p.data = t` under a live Parameter -- is rebuilt and the forward repeated)
        table, dev = self._param_table(verify=not fast)
        for name, t in (("x", x), ("adj_e_index", adj_e_index), ("adj_e_features", adj_e_features),
                        ("col_e_idx", col_e_idx)):
            if t.device != dev:
                raise ValueError(f"`{name}` is on {t.device} but the network is on {dev}")
        if x.dim() != 2 or x.shape[1] != self.node_features_dim:
            raise ValueError(f"x must be [N, {self.node_features_dim}], got {tuple(x.shape)}")     # util.py:16
        if adj_e_features.dim() != 2 or adj_e_features.shape[1] != self.adj_edge_features_dim:
            raise ValueError(f"adj_e_features must be [Ea, {self.adj_edge_features_dim}], "
                             f"got {tuple(adj_e_features.shape)}")
        n = int(x.shape[0])
        bn_train = self.training
        if self.autograd and bn_train and torch.is_grad_enabled():
            from ... import train
            return train.forward_with_grad(self, x, adj_e_index, adj_e_features, col_e_idx), adj_e_features
        if bn_train and n < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        if self.activation_dtype == torch.bfloat16:
            from ... import ops_bf16
            graph = _graph_cache.get_full(n, adj_e_index, adj_e_features, col_e_idx) if self.cache_graph else None
            return ops_bf16.forward(self, x, adj_e_index, adj_e_features, col_e_idx, graph), adj_e_features
        if self.activation_dtype != torch.float32:
            raise ValueError("activation_dtype must be torch.float32 or torch.bfloat16")
        xf = ops._f32c(x, "x")
        ea = ops._f32c(adj_e_features, "adj_e_features")
        dims = self._dims()
        begun = False
        init_done = 0                     # bit 1 of tgnn_forward's update_running: the init MLP's running statistics have their update
        state = {}

        def prepare_new():
            # a NEW layout of the general schedule (cache off, or not in the cache): what the forward does in front of its first layer
            # without the graph (bounds, init MLP, the final MLP's operand images) is queued on the side stream BEFORE the preparation
            # and runs beside it (tgnn_forward_begin / tgnn_forward_resume; the workspace's layout does not depend on the type count
            # up to 16).  _BEGIN_FIRST = 0: behind the preparation's launches, in front of its one synchronisation (after_enqueue).
            def begin(info=None):
                state["bytes"] = lib.tgnn_forward_workspace_bytes(C.byref(dims), n, 0)
                state["ws"] = torch.empty(state["bytes"], dtype=torch.uint8, device=dev)
                state["rc"] = lib.tgnn_forward_begin(C.byref(dims), table, ptr(xf), n, int(update_running), ptr(state["ws"]), state["bytes"],
                                                     None, _lib.side_stream(dev))

            use_begin = bn_train and ops.runs_general_schedule(n) and n > 4096
            if use_begin:
                side = _lib.side_stream_torch(dev)
                if side is None:
                    use_begin = False
                else:
                    side.wait_stream(torch.cuda.current_stream(dev))          # x and the parameters are ready HERE: in front of the preparation
            if use_begin and _BEGIN_FIRST:
                begin()

            def weights(info=None):
                # behind the preparation's launches, before its one synchronisation: the edge weights, with the type count read on
                # the device (tgnn_forward_begin_weights) -- the launch tgnn_forward_resume would otherwise queue first
                if not _BEGIN_FIRST:
                    begin()
                if info is not None and state.get("rc") == 0:
                    rcw = lib.tgnn_forward_begin_weights(C.byref(dims), table, ptr(ea), ptr(info["type_rep_edge"]), ptr(info["result"]),
                                                         n, ptr(state["ws"]), state["bytes"], _lib.current_stream(dev))
                    if rcw not in (0, _lib.ERR_UNSUPPORTED):
                        check(rcw)
            # a small layout (one persistent kernel behind a pre-pass): the pre-pass -- edge weights, parameter pack -- behind the
            # one-launch preparation, the type count read on the device (tgnn_forward_small_prepass)
            use_small = (not use_begin and bn_train and fast and 2 <= n <= min(4096, int(lib.tgnn_get_small_layout_limit())))

            def small_pre(info=None):
                if info is None:
                    return
                nbytes = lib.tgnn_forward_workspace_bytes(C.byref(dims), n, 0)
                wsp = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                rcs = lib.tgnn_forward_small_prepass(C.byref(dims), table, ptr(ea), ptr(info["type_rep_edge"]), ptr(info["result"]), n,
                                                     ptr(wsp), nbytes, _lib.current_stream(dev))
                if rcs == 0:
                    state["small_ws"], state["small_bytes"] = wsp, nbytes
                elif rcs != _lib.ERR_UNSUPPORTED:
                    check(rcs)
            try:
                return ops.prepare_graph(n, adj_e_index, adj_e_features, col_e_idx,
                                         after_enqueue=weights if use_begin else small_pre if use_small else None)
            except Exception:
                side = _lib.side_stream_torch(dev) if state.get("rc") == 0 else None
                if side is not None:                                          # (begin's launches write the workspace freed below)
                    torch.cuda.current_stream(dev).wait_stream(side)
                raise

        graph = _graph_cache.get_full(n, adj_e_index, adj_e_features, col_e_idx, build=prepare_new) if self.cache_graph else prepare_new()
        begun = state.get("rc") == 0
        if begun:
            ws, ws_bytes = state["ws"], state["bytes"]
        elif state.get("rc") not in (None, _lib.ERR_UNSUPPORTED):
            check(state["rc"])
        if begun and graph.n_types > 16:
            ws_need = lib.tgnn_forward_workspace_bytes(C.byref(dims), n, graph.n_types)       # (a larger edge-weight table: rare)
            if ws_need > ws_bytes:
                side = _lib.side_stream_torch(dev)
                if side is not None:
                    torch.cuda.current_stream(dev).wait_stream(side)          # (begin's launches still write the smaller workspace)
                begun = False
                init_done = 2 if update_running else 0                        # (... and have updated the init MLP's running statistics)
        if not begun and "small_ws" in state and graph.n_types <= 16:
            ws, ws_bytes = state["small_ws"], state["small_bytes"]            # (the pre-pass is in it)
        elif not begun:
            ws_bytes = lib.tgnn_forward_workspace_bytes(C.byref(dims), n, graph.n_types)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        if graph.cols is None and graph.groups is not None and (not bn_train or lib.tgnn_set_split_precision(-1) == 0 or
                                                                 self.network_depth > 64):
            graph.ensure_columns()                                            # (no fp16-pair path: the column kernel, not the CSR one)
        probs = torch.empty(n, self.output_dim, dtype=torch.float32, device=dev)
        # One running-statistics update per forward() (the buffers are state: ml_solver.py:129-131 keeps the network in train mode):
        # a just-prepared mid-size layout's batches are verified by the preparation's LAST launch; the forward is queued behind it
        # without waiting, with the DEVICE address of that verdict in the graph struct -- the persistent kernels read it first and
        # leave without output or update when the batches do not fit (tgnn_graph.nn_mid_verdict); the host looks at the word behind
        # its launches and then runs the general schedule, whose update is this forward's one
        g = graph.c_struct(defer_late_check=True)
        if begun:
            check(lib.tgnn_forward_resume(C.byref(dims), table, ptr(xf), ptr(ea), C.byref(g), int(update_running), ptr(probs), ptr(ws),
                                          ws_bytes, _lib.current_stream(dev), _lib.side_stream(dev)))
            if self._param_table_stale():                                     # (the begin / resume pair read a stale table: repeat, plain;
                table, dev = self._param_table()                              #  the running statistics were updated by the first pass)
                check(lib.tgnn_forward(C.byref(dims), table, ptr(xf), ptr(ea), C.byref(g), 0, 0, ptr(probs), ptr(ws), ws_bytes,
                                       _lib.current_stream(dev), _lib.side_stream(dev)))
            return probs, adj_e_features
        rc = lib.tgnn_forward(C.byref(dims), table, ptr(xf), ptr(ea), C.byref(g), int(bn_train and update_running) | init_done,
                              int(not bn_train), ptr(probs), ptr(ws), ws_bytes, _lib.current_stream(dev), _lib.side_stream(dev))
        if rc == _lib.ERR_UNVERIFIED:                                         # (a mid-size forward that is not the two persistent
            g = graph.c_struct()                                              #  kernels alone: nothing was queued; wait for the words)
            rc = lib.tgnn_forward(C.byref(dims), table, ptr(xf), ptr(ea), C.byref(g), int(bn_train and update_running) | init_done,
                                  int(not bn_train), ptr(probs), ptr(ws), ws_bytes, _lib.current_stream(dev), _lib.side_stream(dev))
        check(rc)
        if graph.late_words_failed():                                         # (a just-prepared mid-size layout whose batches did not fit)
            g = graph.c_struct()
            check(lib.tgnn_forward(C.byref(dims), table, ptr(xf), ptr(ea), C.byref(g), int(bn_train and update_running),
                                   int(not bn_train), ptr(probs), ptr(ws), ws_bytes, _lib.current_stream(dev), _lib.side_stream(dev)))
        if self._param_table_stale():                                         # (see the top: checked behind the launches)
            # the repeat leaves the running statistics alone: the first pass -- on the swapped-out storages, which the cache kept
            # alive -- has applied this forward's ONE momentum update to the (verified) running buffers already
            table, dev = self._param_table()
            check(lib.tgnn_forward(C.byref(dims), table, ptr(xf), ptr(ea), C.byref(g), 0,
                                   int(not bn_train), ptr(probs), ptr(ws), ws_bytes, _lib.current_stream(dev), _lib.side_stream(dev)))
        return probs, adj_e_features                                          # TilinGNN.py:78
