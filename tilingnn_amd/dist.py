"""Node-range sharding of the TilinGNN forward over the GPUs of one node.

The reference is single-device (no distributed code at all); this is the build's own
multi-GPU design for the same forward, one process per GPU:

  * rank r owns the contiguous node range [N r / P, N (r+1) / P) and ALL in-edges of those nodes
    (CSR-by-destination is local); remote sources form the HALO, stored behind the owned rows of
    every feature buffer, grouped by owner rank and sorted by global id -- so an
    `all_to_all_single` lands directly in the halo rows, no unpack kernel;
  * per message-passing layer ONE halo exchange (the new adjacency-branch rows and the new pre-BN
    collision-branch rows travel together, 64 floats per halo row; RCCL all-to-all over xGMI) and ONE
    all-reduce of the fp64 BatchNorm column sums of both branches (train-mode BN = statistics over
    all N nodes, ml_solver.py:129-131) -- 20 + 20 + 6 collectives per forward, every one latency
    bound (<= 1 MB / 1 KB), which is why they are fused this way;
  * dense per-node work (GIN MLP, merge, init / final MLP) never leaves the rank.

The schedule is written ONCE as a generator that yields collective requests; three drivers run it:
  `TorchDistComm`   real ranks, torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" in CPU tests)
  `LocalSimComm`    P virtual ranks in one process on one device: lets a 1-GPU box (and the test
                    suite) execute the exact sharded HIP path and compare it with the unsharded one.
Compute goes through a backend object; the product backend is `HipBackend` (the libtgnn kernels via
tilingnn_amd.ops).  Tests pass a CPU backend built on the oracle to check the partition / halo / BN
logic under gloo without a GPU; nothing in this package provides or falls back to a CPU path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# partitioning (host, numpy)
# ----------------------------------------------------------------------------------------------
@dataclass
class Shard:
    rank: int
    world: int
    n_total: int
    lo: int                        # first owned global node id
    n_own: int
    halo_ids: np.ndarray           # global ids of halo rows, grouped by owner rank, sorted inside a group
    recv_counts: List[int]         # halo rows received from every rank (sum == len(halo_ids))
    x: np.ndarray                  # node features of the owned rows
    adj: np.ndarray                # [2, Ea_loc] int64, LOCAL ids (owned 0..n_own-1, halo n_own..)
    adj_attr: np.ndarray
    col: np.ndarray                # [2, Ec_loc] int64, local ids
    send_ids: Optional[List[np.ndarray]] = None   # owned LOCAL row ids every peer needs (filled by setup)
    bounds: Optional[np.ndarray] = None           # [world + 1] first global id of every rank's range (None: the even split of
                                                  # node_range; set by compact_shard, whose ranges follow the surviving nodes)

    @property
    def n_rows(self) -> int:
        return self.n_own + int(self.halo_ids.shape[0])


def node_range(n_total: int, rank: int, world: int, bounds: Optional[np.ndarray] = None):
    if bounds is not None:
        return int(bounds[rank]), int(bounds[rank + 1])
    return n_total * rank // world, n_total * (rank + 1) // world


def even_bounds(n_total: int, world: int) -> np.ndarray:
    return np.array([n_total * r // world for r in range(world + 1)], dtype=np.int64)


def owner_of(ids: np.ndarray, n_total: int, world: int, bounds: Optional[np.ndarray] = None) -> np.ndarray:
    """Rank owning each global node id (inverse of node_range)."""
    ends = np.asarray(bounds[1:] if bounds is not None else even_bounds(n_total, world)[1:], dtype=np.int64)
    return np.searchsorted(ends, ids, side="right")


def make_shard(node_feature: np.ndarray, adj: np.ndarray, adj_attr: np.ndarray, col: np.ndarray, rank: int,
               world: int) -> Shard:
    """Cut rank `rank`'s shard out of a global graph given in the reference's array layout
    (util/data_util.py:164-205).  Edge order inside the shard = global edge order (sums stay reproducible)."""
    n_total = int(node_feature.shape[0])
    lo, hi = node_range(n_total, rank, world)
    adj = adj.reshape(2, -1)
    col = col.reshape(2, -1)
    keep_a = (adj[1] >= lo) & (adj[1] < hi)
    keep_c = (col[1] >= lo) & (col[1] < hi)
    a, c = adj[:, keep_a], col[:, keep_c]
    srcs = np.concatenate([a[0], c[0]])
    remote = np.unique(srcs[(srcs < lo) | (srcs >= hi)])          # sorted global ids == grouped by owner
    owners = owner_of(remote, n_total, world)
    recv_counts = [int(np.count_nonzero(owners == r)) for r in range(world)]

    def localise(e):
        out = np.empty_like(e)
        out[1] = e[1] - lo
        own = (e[0] >= lo) & (e[0] < hi)
        out[0] = np.where(own, e[0] - lo, (hi - lo) + np.searchsorted(remote, e[0]))
        return out
    return Shard(rank, world, n_total, lo, hi - lo, remote, recv_counts, node_feature[lo:hi], localise(a),
                 adj_attr[keep_a], localise(c))


def exchange_send_lists(shards_halo_ids: Sequence[np.ndarray], n_total: int, world: int,
                        bounds: Optional[np.ndarray] = None) -> List[List[np.ndarray]]:
    """send_ids[r][p] = LOCAL owned row ids of rank r that rank p holds in its halo (what the setup
    all-to-all of id lists produces on real ranks)."""
    out = [[np.empty(0, dtype=np.int64) for _ in range(world)] for _ in range(world)]
    for p, ids in enumerate(shards_halo_ids):
        owners = owner_of(ids, n_total, world, bounds)
        for r in range(world):
            sel = ids[owners == r]
            out[r][p] = sel - node_range(n_total, r, world, bounds)[0]
    return out


class EmptyShardError(ValueError):
    """A rank's node range holds no node (check_no_empty_rank; raised on EVERY rank together)."""


def check_no_empty_rank(shard: Shard) -> None:
    """The HIP forward of a shard needs rows on EVERY rank (train-mode BatchNorm, tgnn_graph_prep and tgnn_forward_sharded take
    n_nodes >= 1), and a rank that raised alone would leave the others waiting in an all-to-all for ever.  Every rank holds the
    same `bounds` (compact_shard derives them from the global alive mask), so every rank raises HERE, together, before any
    collective is issued; callers re-shard the remainder over fewer ranks (solve_sharded) -- a late greedy round of a large
    layout is a small layout."""
    b = shard.bounds if shard.bounds is not None else even_bounds(shard.n_total, shard.world)
    sizes = np.diff(np.asarray(b, dtype=np.int64))
    if np.any(sizes < 1):
        empty = [int(r) for r in np.nonzero(sizes < 1)[0]]
        raise EmptyShardError(f"ranks {empty} own no node of the {shard.n_total}-node layout split over {shard.world} ranks: "
                              "re-shard over fewer ranks")


def compact_shard(shard: Shard, alive: np.ndarray) -> Shard:
    """A greedy round later (the reference's util/algorithms.py:18-62 scores, every round, the sub-layout of the still
    unlabelled nodes: tiling/brick_layout.py:248-286): this rank's shard of that sub-layout, cut LOCALLY from its shard of the
    round before -- mask -> local compact -> halo-list rebuild -- with no gather of the layout anywhere.
      alive: bool [shard.n_total] over the CURRENT global numbering (every rank holds it: the acceptance sweep is a host
      loop over the gathered probabilities).
    The sub-layout's nodes are numbered as compute_sub_layout numbers them (rank among the alive ones), every rank keeps the
    survivors of its own range (so the ranges stay contiguous: `bounds`), an edge survives iff both ends do, in its old order;
    the halo = the alive remote sources of the surviving edges.  The communicator's setup must run again (send lists).
    A range may come out EMPTY (n_own == 0): the Python schedule over a CPU backend runs such a shard, the HIP forward does not
    (check_no_empty_rank raises on every rank together)."""
    world, rank = shard.world, shard.rank
    alive = np.asarray(alive, dtype=bool)
    assert alive.shape[0] == shard.n_total
    old_bounds = shard.bounds if shard.bounds is not None else even_bounds(shard.n_total, world)
    prefix = np.concatenate([[0], np.cumsum(alive, dtype=np.int64)])          # new id of node g (if alive) = prefix[g]
    new_bounds = prefix[old_bounds]
    lo, hi = int(old_bounds[rank]), int(old_bounds[rank + 1])
    nlo, nhi = int(new_bounds[rank]), int(new_bounds[rank + 1])
    gid = np.concatenate([np.arange(lo, hi, dtype=np.int64), shard.halo_ids.astype(np.int64)])   # global id of every local row
    row_alive = alive[gid]
    new_gid = prefix[gid]

    def cut(e):
        keep = row_alive[e[0]] & row_alive[e[1]]
        return keep, new_gid[e[:, keep]]                                       # surviving edges in NEW global ids
    keep_a, a = cut(shard.adj)
    keep_c, c = cut(shard.col)
    srcs = np.concatenate([a[0], c[0]])
    remote = np.unique(srcs[(srcs < nlo) | (srcs >= nhi)])
    owners = np.searchsorted(new_bounds[1:], remote, side="right")
    recv_counts = [int(np.count_nonzero(owners == r)) for r in range(world)]

    def localise(e):
        out = np.empty_like(e)
        out[1] = e[1] - nlo
        own = (e[0] >= nlo) & (e[0] < nhi)
        out[0] = np.where(own, e[0] - nlo, (nhi - nlo) + np.searchsorted(remote, e[0]))
        return out
    return Shard(rank, world, int(prefix[-1]), nlo, nhi - nlo, remote, recv_counts, shard.x[alive[lo:hi]], localise(a),
                 shard.adj_attr[keep_a], localise(c), None, new_bounds)


def compact_shard_device(shard: Shard, inputs: Dict[str, Tensor], alive) -> "tuple[Shard, Dict[str, Tensor]]":
    """compact_shard with the shard's arrays where they live -- in HBM (`inputs` = HipBackend.upload(shard) or the result of an
    earlier call): the same sub-layout shard, bit for bit, cut by the device (csrc/graph_prep.hip: tgnn_shard_alive_rows marks the
    surviving owned rows and the halo rows that are still somebody's source, tgnn_sublayout_compact -- the kernel of the
    single-GPU loop -- re-indexes rows and edges over the shard's local row space).  The host does what is O(world) or O(halo):
    the ranges' new bounds from the alive mask it holds anyway, the new halo list (global numbers) for the send-list exchange.
    One read-back: the three counts + the surviving halo rows.  Returns (shard', inputs'); shard'.x / .adj / .col stay None (the
    data is inputs'); the communicator's setup must run again and inputs'["send_idx"] be rebuilt (refresh_send_idx)."""
    import ctypes as C
    from . import _lib
    lib, check, ptr = _lib.lib, _lib.check, _lib.ptr
    world, rank = shard.world, shard.rank
    alive_h = np.asarray(alive.cpu().numpy() if torch.is_tensor(alive) else alive).astype(bool)
    assert alive_h.shape[0] == shard.n_total
    dev = inputs["x"].device
    old_bounds = shard.bounds if shard.bounds is not None else even_bounds(shard.n_total, world)
    prefix = np.concatenate([[0], np.cumsum(alive_h, dtype=np.int64)])
    new_bounds = prefix[old_bounds]
    lo, hi = int(old_bounds[rank]), int(old_bounds[rank + 1])
    nlo, nhi = int(new_bounds[rank]), int(new_bounds[rank + 1])
    n_own, n_rows = shard.n_own, shard.n_rows
    n_halo = n_rows - n_own
    x, adj, attr, col = inputs["x"], inputs["adj"], inputs["attr"], inputs["col"]
    fx = int(x.shape[1])
    ea, ec, fe = int(adj.shape[1]), int(col.shape[1]), int(attr.shape[1])   # (the width also of an EMPTY attribute array: every rank keeps [*, Fe])
    stream = _lib.current_stream(dev)
    alive_d = (alive if torch.is_tensor(alive) and alive.is_cuda else torch.from_numpy(alive_h.astype(np.int32)).to(dev)).to(torch.int32)
    gid = torch.cat([torch.arange(lo, hi, dtype=torch.int64, device=dev),
                     torch.from_numpy(np.ascontiguousarray(shard.halo_ids, dtype=np.int64)).to(dev)])
    alive_local = torch.empty(max(n_rows, 1), dtype=torch.int32, device=dev)
    tail = torch.zeros(4, dtype=torch.int64, device=dev)        # counts [3] | error flag
    err = tail[3:].view(torch.int32)[:1]
    check(lib.tgnn_shard_alive_rows(ptr(alive_d), ptr(gid), n_own, n_rows, ptr(adj) if ea else None, ea, ptr(col) if ec else None, ec,
                                    ptr(alive_local), ptr(err), stream))
    x_pad = torch.cat([x, torch.zeros(n_halo, fx, dtype=torch.float32, device=dev)]) if n_halo else x
    x_out = torch.empty(n_rows, fx, dtype=torch.float32, device=dev)
    inverse = torch.empty(n_rows, dtype=torch.int64, device=dev)
    adj_out = torch.empty(2 * max(ea, 1), dtype=torch.int64, device=dev)
    attr_out = torch.empty(max(ea, 1) * fe, dtype=torch.float32, device=dev)
    col_out = torch.empty(2 * max(ec, 1), dtype=torch.int64, device=dev)
    ws_bytes = int(lib.tgnn_sublayout_workspace_bytes(n_rows, ea, ec))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.tgnn_sublayout_compact(ptr(alive_local), n_rows, ptr(x_pad), fx, ptr(adj) if ea else None, ea, ptr(attr) if ea else None, fe,
                                     ptr(col) if ec else None, ec, ptr(x_out), ptr(inverse), ptr(adj_out), ptr(attr_out), ptr(col_out),
                                     ptr(tail[:3]), ptr(err), ptr(ws), ws_bytes, stream))
    n_own2 = nhi - nlo
    rows2, ea2, ec2, bad = tail.cpu().tolist()                  # the one sync
    if bad:
        raise IndexError("edge index out of range in the shard")
    kept_halo = inverse[n_own2:rows2].cpu().numpy() - n_own     # old halo slots that survive, ascending
    halo_ids2 = prefix[np.asarray(shard.halo_ids, dtype=np.int64)[kept_halo]] if rows2 > n_own2 else np.empty(0, dtype=np.int64)
    owners = np.searchsorted(new_bounds[1:], halo_ids2, side="right")
    recv_counts = [int(np.count_nonzero(owners == r)) for r in range(world)]
    new = Shard(rank, world, int(prefix[-1]), nlo, n_own2, halo_ids2, recv_counts, None, None, None, None, None, new_bounds)
    new_inputs = {"x": x_out[:n_own2], "adj": adj_out[:2 * ea2].view(2, ea2), "attr": attr_out[:ea2 * fe].view(ea2, fe),
                  "col": col_out[:2 * ec2].view(2, ec2), "send_idx": torch.empty(0, dtype=torch.int32, device=dev)}
    return new, new_inputs


def refresh_send_idx(shard: Shard, inputs: Dict[str, Tensor]) -> None:
    """inputs["send_idx"] from shard.send_ids (after the communicator's setup ran on a compacted shard)."""
    send = shard.send_ids if shard.send_ids is not None else []
    flat = np.concatenate(send) if len(send) else np.empty(0)
    inputs["send_idx"] = torch.from_numpy(np.ascontiguousarray(flat)).to(torch.int32).to(inputs["x"].device)


# ----------------------------------------------------------------------------------------------
# compute backends
# ----------------------------------------------------------------------------------------------
class HipBackend:
    """The product backend: libtgnn kernels through tilingnn_amd.ops.  Needs a GPU."""

    def __init__(self, device):
        from . import ops
        self.ops, self.device = ops, torch.device(device)

    def tensor(self, a: np.ndarray, dtype) -> Tensor:
        return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(self.device)

    def upload(self, shard: Shard) -> Dict[str, Tensor]:
        """Shard inputs -> HBM, once per layout (the arrays ML_Solver.predict would hand over)."""
        send = shard.send_ids if shard.send_ids is not None else []
        return {"x": self.tensor(shard.x, torch.float32), "adj": self.tensor(shard.adj, torch.int64),
                "attr": self.tensor(shard.adj_attr, torch.float32), "col": self.tensor(shard.col, torch.int64),
                "send_idx": self.tensor(np.concatenate(send) if len(send) else np.empty(0), torch.int32)}

    def prepare(self, shard: Shard, inputs: Dict[str, Tensor]):
        """Per-forward graph preparation on the device-resident inputs (CSR, edge types, NNConv tiles)."""
        check_no_empty_rank(shard)
        g = self.ops.prepare_graph(shard.n_own, inputs["adj"], inputs["attr"], inputs["col"], n_src_nodes=shard.n_rows)
        return g, inputs["attr"]

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32, device=self.device)

    def edge_tables(self, graph, attr, net) -> list:
        return [self.ops.edge_weight_table(attr, graph, *l.nnConv._edge_mlp_params(), net.network_width)
                for l in net.brch_1_graph_conv_layers]

    def dense(self, a, lin, act, in_stat=None, slot_major=False):
        f = lin.out_features
        parts = self.ops.new_partials(f, self.device)
        out, n_parts = self.ops.dense_act(a, lin.weight, lin.bias, act, in_stat=in_stat, partials=parts,
                                          slot_major=slot_major)
        return out, self.ops.bn_sums(parts, n_parts, f)

    def nnconv(self, h_rows, graph, wtab, conv, act):
        parts = self.ops.new_partials(conv.out_channels, self.device)
        out, n_parts = self.ops.nnconv_mean(h_rows, graph, wtab, conv.root, conv.bias, act=act, partials=parts)
        return out, self.ops.bn_sums(parts, n_parts, conv.out_channels)

    def gin(self, a_rows, graph, conv, act, in_stat):
        c = int(a_rows.shape[1])
        parts = self.ops.new_partials(c, self.device)
        out, n_parts = self.ops.gin(a_rows, graph, conv.eps, *conv._mlp_params(), act=act, in_stat=in_stat,
                                    partials=parts)
        return out, self.ops.bn_sums(parts, n_parts, c)

    def bn_stat(self, sums, n_total, bn, update_running):
        return self.ops.bn_stat_from_sums(sums, n_total, bn, update_running)

    def bn_apply(self, v, stat):
        return self.ops.bn_apply(v, stat)

    def merge(self, a1, stat1, a2, stat2, resid, out):
        self.ops.merge(a1, stat1, a2, stat2, resid, out=out, want_h2=False)

    def pack_rows(self, src, idx, out, col_offset):
        self.ops.rows_gather(src, idx, out, col_offset)


# ----------------------------------------------------------------------------------------------
# the sharded forward, written once
# ----------------------------------------------------------------------------------------------
class ShardProgram:
    """One rank's share of TilinGNN.forward (graph_networks/networks/TilinGNN.py:51-78).
    `run()` is a generator: it yields ("allreduce", fp64 tensor) / ("alltoall", send, send_splits,
    recv_view, recv_splits) requests and is resumed once the collective has completed."""

    def __init__(self, net, shard: Shard, backend, update_running: bool = True, inputs=None):
        self.net, self.shard, self.be, self.update_running = net, shard, backend, update_running
        send = shard.send_ids
        assert send is not None, "shard.send_ids not set: run setup (exchange_send_lists / TorchDistComm.setup)"
        self.inputs = inputs if inputs is not None else backend.upload(shard)
        self.graph, self.attr = backend.prepare(shard, self.inputs)
        self.x = self.inputs["x"]
        self.send_splits = [int(s.shape[0]) for s in send]
        self.send_idx = self.inputs["send_idx"]
        self.recv_splits = list(shard.recv_counts)
        c = net.network_width
        n_rows, n_send = shard.n_rows, int(self.send_idx.shape[0])
        self.h1 = backend.zeros(n_rows, c)                 # adjacency-branch features: owned rows then halo
        self.a2 = backend.zeros(n_rows, c)                 # pre-BN collision-branch activations, same layout
        self.sendbuf = backend.zeros(max(n_send, 1), 2 * c)
        self.recvbuf = backend.zeros(max(n_rows - shard.n_own, 1), 2 * c)

    def _exchange(self, with_a2: bool):
        """Halo rows of h1 (and a2): pack the rows every peer needs, one all-to-all, drop into the halo rows."""
        c, n_own = self.net.network_width, self.shard.n_own
        n_halo, n_send = self.shard.n_rows - n_own, int(self.send_idx.shape[0])
        self.be.pack_rows(self.h1, self.send_idx, self.sendbuf, 0)
        if with_a2:
            self.be.pack_rows(self.a2, self.send_idx, self.sendbuf, c)
        send, recv = self.sendbuf[:n_send], self.recvbuf[:n_halo]
        yield ("alltoall", send, self.send_splits, recv, self.recv_splits)
        if n_halo:
            self.h1[n_own:] = recv[:, :c]
            if with_a2:
                self.a2[n_own:] = recv[:, c:]

    def run(self):
        net, be, sh = self.net, self.be, self.shard
        ACT_LEAKY, ACT_SIGMOID = 1, 2
        n_own, n_total, c, depth = sh.n_own, sh.n_total, net.network_width, net.network_depth

        def bn(sums, module):
            return be.bn_stat(sums, n_total, module, self.update_running)

        # ---- init MLP (TilinGNN.py:54): two Linear -> LeakyReLU -> BN, statistics over ALL nodes
        l0, l1 = net.init_node_feature_trans.mlp[0], net.init_node_feature_trans.mlp[1]
        t0, s = be.dense(self.x, l0.linear, ACT_LEAKY)
        yield ("allreduce", s)
        st0 = bn(s, l0.batch_norm)
        t1, s = be.dense(t0, l1.linear, ACT_LEAKY, in_stat=st0)
        yield ("allreduce", s)
        st1 = bn(s, l1.batch_norm)
        mid = be.zeros(depth + 1, n_own, c)                # slot-major skip buffer (TilinGNN.py:58,71,74)
        mid[0] = be.bn_apply(t1, st1)
        self.h1[:n_own] = mid[0]
        yield from self._exchange(with_a2=False)
        tables = be.edge_tables(self.graph, self.attr, net)

        # ---- message passing layers (TilinGNN.py:59-71)
        stat2 = None
        for i in range(depth):
            l1m, l2m = net.brch_1_graph_conv_layers[i], net.brch_2_coll_conv_layers[i]
            a1, s1 = be.nnconv(self.h1, self.graph, tables[i], l1m.nnConv, ACT_LEAKY)
            gin_in = self.h1 if i == 0 else self.a2          # layer 0: both branches start from middle[0] (:55)
            a2, s2 = be.gin(gin_in, self.graph, l2m.ginConv, ACT_LEAKY, stat2)
            both = torch.cat([s1, s2])
            yield ("allreduce", both)                       # one message for both BatchNorms of the layer
            stat1 = bn(both[: 2 * c], l1m.batch_norm)
            stat2 = bn(both[2 * c:], l2m.batch_norm)
            resid = mid[i - 2] if i >= 2 else None           # residual_skip_num = 2 (:25,67-69)
            be.merge(a1, stat1, a2, stat2, resid, mid[i + 1])
            if i + 1 < depth:
                self.h1[:n_own] = mid[i + 1]
                self.a2[:n_own] = a2
                yield from self._exchange(with_a2=True)

        # ---- final MLP over the concatenation (TilinGNN.py:74-76); K block k = middle[k]
        fm = net.final_mlp[0].mlp
        v, s = be.dense(mid, fm[0].linear, ACT_LEAKY, slot_major=True)
        yield ("allreduce", s)
        stat = bn(s, fm[0].batch_norm)
        for layer in list(fm)[1:]:
            v, s = be.dense(v, layer.linear, ACT_LEAKY, in_stat=stat)
            yield ("allreduce", s)
            stat = bn(s, layer.batch_norm)
        probs, _ = be.dense(v, net.final_mlp[1].linear, ACT_SIGMOID, in_stat=stat)
        return probs


# ----------------------------------------------------------------------------------------------
# drivers
# ----------------------------------------------------------------------------------------------
class TorchDistComm:
    """Real ranks: one process per GPU, torch.distributed ("nccl" is RCCL on ROCm; "gloo" in CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group

    def setup(self, shard: Shard) -> None:
        """Tell every owner which of its rows this rank keeps as halo (ids travel once per layout)."""
        dist, world = self.dist, shard.world
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        counts_out = torch.tensor(shard.recv_counts, dtype=torch.int64, device=dev)
        counts_in = torch.empty(world, dtype=torch.int64, device=dev)
        dist.all_to_all_single(counts_in, counts_out, group=self.group)
        counts_in = counts_in.cpu().tolist()
        ids_out = torch.from_numpy(shard.halo_ids.astype(np.int64)).to(dev)
        ids_in = torch.empty(int(sum(counts_in)), dtype=torch.int64, device=dev)
        dist.all_to_all_single(ids_in, ids_out, output_split_sizes=counts_in, input_split_sizes=shard.recv_counts,
                               group=self.group)
        ids_in = ids_in.cpu().numpy() - shard.lo
        offs = np.concatenate([[0], np.cumsum(counts_in)]).astype(np.int64)
        shard.send_ids = [ids_in[offs[p]:offs[p + 1]] for p in range(world)]

    def run(self, program: ShardProgram) -> Tensor:
        gen = program.run()
        try:
            req = next(gen)
            while True:
                if req[0] == "allreduce":
                    self.dist.all_reduce(req[1], group=self.group)
                else:
                    _, send, send_splits, recv, recv_splits = req
                    self.dist.all_to_all_single(recv, send, output_split_sizes=recv_splits,
                                                input_split_sizes=send_splits, group=self.group)
                req = gen.send(None)
        except StopIteration as stop:
            return stop.value


class LocalSimComm:
    """P virtual ranks in ONE process / on ONE device, stepped in lock-step: every collective is carried out
    with plain tensor copies.  Same ShardProgram, same kernels, same halo layout as the real thing."""

    @staticmethod
    def setup(shards: Sequence[Shard]) -> None:
        lists = exchange_send_lists([s.halo_ids for s in shards], shards[0].n_total, shards[0].world, shards[0].bounds)
        for r, s in enumerate(shards):
            s.send_ids = lists[r]

    @staticmethod
    def run(programs: Sequence[ShardProgram]) -> List[Tensor]:
        gens = [p.run() for p in programs]
        world = len(gens)
        reqs, results = [None] * world, [None] * world
        for r, g in enumerate(gens):
            reqs[r] = next(g)
        while any(q is not None for q in reqs):
            kinds = {q[0] for q in reqs if q is not None}
            assert len(kinds) == 1 and all(q is not None for q in reqs), "ranks diverged"
            if kinds == {"allreduce"}:
                total = torch.stack([q[1] for q in reqs]).sum(0)
                for q in reqs:
                    q[1].copy_(total)
            else:
                for dst in range(world):
                    _, _, _, recv, recv_splits = reqs[dst]
                    off = 0
                    for src in range(world):
                        k = recv_splits[src]
                        if k:
                            _, send, send_splits, _, _ = reqs[src]
                            s0 = sum(send_splits[:dst])
                            assert send_splits[dst] == k
                            recv[off:off + k].copy_(send[s0:s0 + k])
                        off += k
            for r, g in enumerate(gens):
                try:
                    reqs[r] = g.send(None)
                except StopIteration as stop:
                    reqs[r], results[r] = None, stop.value
        return results


class FusedShardForward:
    """One shard's forward through `tgnn_forward_sharded` (csrc/forward.hip): the whole schedule of ShardProgram in
    ONE library call that enqueues ~150 launches and calls back for the collectives -- the per-op Python
    schedule above costs ~2x the GPU time in host overhead at 100k nodes.  `comm` supplies
    allreduce(tensor) and alltoall(send, send_splits, recv, recv_splits).
    fused=True (default): one all-to-all per message-passing layer carries the raw halo rows of both branches and
    the local BatchNorm sums to every peer (27 collectives per forward); fused=False: all-reduce of the sums, then
    all-to-all of the merged rows (47)."""

    def __init__(self, net, shard: Shard, device, comm, inputs=None, fused: bool = True, rccl: "LibraryRccl" = None):
        import ctypes as C
        from . import _lib, ops
        self._C, self._lib, self._ops = C, _lib, ops
        self.net, self.shard, self.comm, self.dev = net, shard, comm, torch.device(device)
        assert shard.send_ids is not None, "shard.send_ids not set: run the communicator's setup first"
        check_no_empty_rank(shard)
        self.inputs = inputs if inputs is not None else HipBackend(device).upload(shard)
        c = net.network_width
        self.n_send = int(self.inputs["send_idx"].shape[0])
        self.n_halo = shard.n_rows - shard.n_own
        self.send_splits = [int(x.shape[0]) for x in shard.send_ids]
        self.recv_splits = list(shard.recv_counts)
        world = shard.world
        self.sum_buf = torch.zeros(max(1024, 128 * (world + 2)), dtype=torch.float64, device=self.dev)
        self.send_buf = torch.zeros(max(self.n_send + 4 * world, 1) * 2 * c, dtype=torch.float32, device=self.dev)
        self.recv_buf = torch.zeros(max(self.n_halo + 4 * world, 1) * 2 * c, dtype=torch.float32, device=self.dev)
        self._error = None
        # rccl given: the library issues the collectives itself (csrc/rccl_comm.hip) -- `comm` is then only the fallback
        self.rccl = rccl
        self._counts = ((C.c_int64 * world)(*self.send_splits), (C.c_int64 * world)(*self.recv_splits))
        # fused layers (one all-to-all per layer, include/tgnn.h): per peer its rows, then 4 rows of BatchNorm sums
        send_ext, recv_ext, halo_at = [], [], 0
        for p_ in range(world):
            send_ext.append(np.asarray(shard.send_ids[p_], dtype=np.int32))
            send_ext.append(np.array([-1, -2, -3, -4], dtype=np.int32))
            k = int(shard.recv_counts[p_])
            recv_ext.append(np.arange(halo_at, halo_at + k, dtype=np.int32))
            recv_ext.append(np.array([-1 - (4 * p_ + j) for j in range(4)], dtype=np.int32))
            halo_at += k
        self.fused = fused and c == 32
        # True: the neighbourhood sum of a layer's collision branch on the side stream (tgnn_shard.side_stream), beside the merge of
        # the layer before and the head of the NNConv.  Measured at world 1, 100k nodes, per step: one stream 3.29 ms; the whole
        # GIN there 3.65 ms (its MLP finds no CU beside an NNConv block: 52 us instead of 19); only the sum there 3.49 ms -- GIN_i
        # needs the exchange of layer i-1, the chain cannot run ahead as it does unsharded, and every cross-queue dependency on
        # the critical path costs ~10-15 us (scratch/time_sharded.py, scratch/sharded_trace.sh)
        self.two_streams = False
        msg_rows = np.concatenate(send_ext)
        self.send_idx_fused = torch.from_numpy(msg_rows).to(self.dev)
        self.recv_idx_fused = torch.from_numpy(np.concatenate(recv_ext)).to(self.dev)
        # [r6] the inverse of the message's row list (tgnn_shard.send_row_ptr / send_row_slot): with it the NNConv writes the adjacency
        # branch's halo message itself -- no pack launch on the step's critical chain
        slots = np.nonzero(msg_rows >= 0)[0].astype(np.int32)
        order = np.argsort(msg_rows[slots], kind="stable")
        counts = np.bincount(msg_rows[slots], minlength=shard.n_own).astype(np.int64)
        ptr_ = np.zeros(shard.n_own + 1, dtype=np.int32)
        np.cumsum(counts, out=ptr_[1:])
        self.send_row_ptr = torch.from_numpy(ptr_).to(self.dev)
        self.send_row_slot = torch.from_numpy(slots[order] if slots.size else np.zeros(1, dtype=np.int32)).to(self.dev)
        self.pack_in_nnconv = True

        self._ext_streams = {}

        def on_stream(stream_ptr):
            """The collective goes onto the stream the library names (the side stream for the collision branch's exchange)."""
            import contextlib
            if not stream_ptr or self.dev.type != "cuda":
                return contextlib.nullcontext()
            st = self._ext_streams.get(stream_ptr)
            if st is None:
                st = self._ext_streams[stream_ptr] = torch.cuda.ExternalStream(stream_ptr, device=self.dev)
            return torch.cuda.stream(st)

        def allreduce_cb(_ctx, _buf, count, stream_ptr):
            try:
                with on_stream(stream_ptr):
                    self.comm.allreduce(self.sum_buf[:count])
                return 0
            except BaseException as exc:                      # never let an exception cross the C frame
                self._error = exc
                return 1

        def alltoall_cb(_ctx, send_ptr, recv_ptr, row_floats, extra_rows, stream_ptr):
            try:
                n_out = self.n_send + extra_rows * self.shard.world
                n_in = self.n_halo + extra_rows * self.shard.world
                # (the split exchange hands over the second halves of the two buffers for the collision branch)
                so = (send_ptr - self.send_buf.data_ptr()) // 4 if send_ptr else 0
                ro = (recv_ptr - self.recv_buf.data_ptr()) // 4 if recv_ptr else 0
                send = self.send_buf[so: so + n_out * row_floats].view(n_out, row_floats)
                recv = self.recv_buf[ro: ro + n_in * row_floats].view(n_in, row_floats)
                with on_stream(stream_ptr):
                    self.comm.alltoall(send, [k + extra_rows for k in self.send_splits], recv,
                                       [k + extra_rows for k in self.recv_splits])
                return 0
            except BaseException as exc:
                self._error = exc
                return 1

        self._cbs = (_lib.ALLREDUCE_CB(allreduce_cb), _lib.ALLTOALL_CB(alltoall_cb))   # keep them alive

    def step(self) -> Tensor:
        """Graph preparation + forward of this shard; returns probs of the owned rows."""
        C, _lib, ops, sh, net = self._C, self._lib, self._ops, self.shard, self.net
        inp = self.inputs
        graph = ops.prepare_graph(sh.n_own, inp["adj"], inp["attr"], inp["col"], n_src_nodes=sh.n_rows)
        if not self.fused or _lib.lib.tgnn_set_split_precision(-1) == 0:
            graph.ensure_columns()         # (the all-reduce + all-to-all scheme stays on bf16 x 3: the column kernel, not the CSR one)
        dims = net._dims()
        table, _ = net._param_table()
        ws_bytes = _lib.lib.tgnn_forward_sharded_workspace_bytes(C.byref(dims), sh.n_own, sh.n_rows, graph.n_types)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.dev)
        probs = torch.empty(sh.n_own, net.output_dim, dtype=torch.float32, device=self.dev)
        desc = _lib.ShardDesc(sh.n_own, sh.n_rows, sh.n_total, inp["send_idx"].data_ptr(), self.n_send,
                              self.sum_buf.data_ptr(), self.send_buf.data_ptr(), self.recv_buf.data_ptr(),
                              self._cbs[0], self._cbs[1], None, sh.world, sh.rank,
                              self.send_idx_fused.data_ptr() if self.fused else None,
                              self.recv_idx_fused.data_ptr() if self.fused else None,
                              _lib.side_stream(self.dev) if self.two_streams else None,
                              self.rccl.comm if self.rccl else None, self.rccl.comm_side if self.rccl else None,
                              C.cast(self._counts[0], C.c_void_p), C.cast(self._counts[1], C.c_void_p),
                              self.send_row_ptr.data_ptr() if self.fused and self.pack_in_nnconv else None,
                              self.send_row_slot.data_ptr() if self.fused and self.pack_in_nnconv else None)
        g = graph.c_struct()
        self._error = None
        rc = _lib.lib.tgnn_forward_sharded(C.byref(dims), table, ops.ptr(inp["x"]), ops.ptr(inp["attr"]), C.byref(g),
                                           C.byref(desc), int(net.training), ops.ptr(probs), ops.ptr(ws), ws_bytes,
                                           _lib.current_stream(self.dev))
        if self._error is not None:
            raise self._error
        _lib.check(rc)
        return probs


class LibraryRccl:
    """Two RCCL communicators of the library's own (csrc/rccl_comm.hip: main chain / side stream) for FusedShardForward.  Every
    rank of the torch.distributed job constructs one (a collective): rank 0 draws the unique ids, torch.distributed carries
    the 128 bytes to the others -- its only part; the forward's collectives then never touch Python."""

    def __init__(self, device, group=None):
        import ctypes as C
        import torch.distributed as dist
        from . import _lib
        self._lib, self._C = _lib, C
        # The forward's side stream is picked (and bound to its hardware queue) BEFORE the communicators exist: RCCL creates
        # streams of its own, and a side stream that came to share the main stream's queue made the two chains run strictly
        # one after the other (rocprof: every kernel on one queue).  _lib.side_stream measures what it picks.
        _lib.side_stream(torch.device(device))
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        nbytes = int(_lib.lib.tgnn_rccl_unique_id_bytes())
        ids = [None, None]
        if rank == 0:
            for k in range(2):
                buf = (C.c_ubyte * nbytes)()
                _lib.check(_lib.lib.tgnn_rccl_unique_id(buf))
                ids[k] = bytes(buf)
        dist.broadcast_object_list(ids, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        handles = []
        with torch.cuda.device(device):
            for k in range(2):
                h = C.c_void_p()
                buf = (C.c_ubyte * nbytes).from_buffer_copy(ids[k])
                _lib.check(_lib.lib.tgnn_rccl_comm_create(buf, rank, world, C.byref(h)))
                handles.append(h)
        self.comm, self.comm_side = handles
        self.rank, self.world = rank, world

    @staticmethod
    def counters():
        import ctypes as C
        from . import _lib
        out = (C.c_int64 * 2)()
        _lib.lib.tgnn_rccl_counters(out)
        return int(out[0]), int(out[1])

    def close(self):
        for h in (self.comm, self.comm_side):
            if h:
                self._lib.lib.tgnn_rccl_comm_destroy(h)
        self.comm = self.comm_side = None


class TorchDistCollectives:
    """allreduce / alltoall of FusedShardForward over torch.distributed ("nccl" = RCCL)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.n_allreduce = self.n_alltoall = 0               # calls issued so far (bench.py reports them per forward)

    def allreduce(self, t: Tensor) -> None:
        self.n_allreduce += 1
        self.dist.all_reduce(t, group=self.group)

    def alltoall(self, send: Tensor, send_splits, recv: Tensor, recv_splits) -> None:
        self.n_alltoall += 1
        self.dist.all_to_all_single(recv, send, output_split_sizes=recv_splits, input_split_sizes=send_splits,
                                    group=self.group)

    def allgather(self, t: Tensor) -> List[Tensor]:
        """Every rank's 1-d tensor (lengths may differ), in rank order."""
        dist, world = self.dist, self.dist.get_world_size(self.group)
        lens = torch.zeros(world, dtype=torch.int64, device=t.device)
        dist.all_gather_into_tensor(lens, torch.tensor([t.numel()], dtype=torch.int64, device=t.device), group=self.group)
        lens = lens.cpu().tolist()
        pad = torch.zeros(max(lens), dtype=t.dtype, device=t.device)
        pad[: t.numel()] = t.reshape(-1)
        out = torch.empty(world * max(lens), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, pad, group=self.group)
        return [out[r * max(lens): r * max(lens) + lens[r]] for r in range(world)]


class ThreadSimCollectives:
    """P virtual ranks = P Python threads on ONE device and ONE stream (tests): a collective is a rendezvous on a
    barrier plus plain tensor copies.  ctypes releases the GIL for the duration of the library call and takes it
    again inside the callbacks, so the ranks really interleave at the collectives like separate processes do."""

    class Hub:
        def __init__(self, world: int):
            import threading
            self.world, self.barrier = world, threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, hub: "ThreadSimCollectives.Hub", rank: int):
        self.hub, self.rank = hub, rank

    def allreduce(self, t: Tensor) -> None:
        hub = self.hub
        hub.slots[self.rank] = t
        hub.barrier.wait()
        total = torch.stack(list(hub.slots)).sum(0) if self.rank == 0 else None
        if self.rank == 0:
            hub.total = total
        hub.barrier.wait()
        t.copy_(hub.total)
        hub.barrier.wait()

    def setup(self, shard: Shard) -> None:
        """The send-list exchange of TorchDistComm.setup among the thread-simulated ranks: every rank posts its halo list and
        picks, from every peer's list, the rows it owns."""
        hub = self.hub
        hub.slots[self.rank] = np.asarray(shard.halo_ids, dtype=np.int64)
        hub.barrier.wait()
        lo, hi = node_range(shard.n_total, shard.rank, shard.world, shard.bounds)
        shard.send_ids = [ids[(ids >= lo) & (ids < hi)] - lo for ids in hub.slots]
        hub.barrier.wait()

    def allgather(self, t: Tensor) -> List[Tensor]:
        hub = self.hub
        hub.slots[self.rank] = t
        hub.barrier.wait()
        out = [s.clone() for s in hub.slots]
        hub.barrier.wait()
        return out

    def alltoall(self, send: Tensor, send_splits, recv: Tensor, recv_splits) -> None:
        hub = self.hub
        hub.slots[self.rank] = (send, send_splits)
        hub.barrier.wait()
        off = 0
        for src in range(hub.world):
            k = recv_splits[src]
            if k:
                s_send, s_splits = hub.slots[src]
                s0 = sum(s_splits[: self.rank])
                assert s_splits[self.rank] == k
                recv[off:off + k].copy_(s_send[s0:s0 + k])
            off += k
        hub.barrier.wait()


class ShardedTilinGNN:
    """bench.py's handle on the multi-GPU path: rank `rank` of `world`, RCCL collectives."""

    def __init__(self, net, super_graph, rank: int, world: int, device, group=None):
        shard = make_shard(super_graph.node_feature, super_graph.align_edge_index, super_graph.align_edge_features,
                           super_graph.collide_edge_index, rank, world)
        self.comm = TorchDistComm(group)
        self.comm.setup(shard)
        self.shard, self.net, self.backend = shard, net, HipBackend(device)
        self.n_local, self.ea_local, self.ec_local = shard.n_own, int(shard.adj.shape[1]), int(shard.col.shape[1])
        self.inputs = self.backend.upload(shard)          # resident in HBM before any timed step
        self.program = None
        self.collectives = TorchDistCollectives(group)
        # the collectives of the fused forward: issued by the library over RCCL communicators of its own (default where the
        # job runs on "nccl" = RCCL and librccl is found; TGNN_LIBRARY_RCCL=0: host callbacks into torch.distributed) -- and
        # then with the split exchange (the collision branch's all-to-all on the side stream)
        import os
        import torch.distributed as dist
        from . import _lib
        self.rccl = None
        if (torch.device(device).type == "cuda" and dist.get_backend(group) == "nccl" and _lib.lib.tgnn_rccl_available()
                and os.environ.get("TGNN_LIBRARY_RCCL", "1") == "1"):
            try:
                self.rccl = LibraryRccl(device, group)
            except Exception as exc:                           # (no RCCL entry points, communicator creation refused, ...)
                import warnings
                warnings.warn(f"tilingnn_amd.dist: library-issued RCCL collectives unavailable ({exc}); "
                              "falling back to torch.distributed callbacks")
                self.rccl = None
        self.fused = FusedShardForward(net, shard, device, self.collectives, inputs=self.inputs, rccl=self.rccl)
        self.fused.two_streams = self.rccl is not None
        self._rccl0 = LibraryRccl.counters() if self.rccl else (0, 0)
        self.steps = 0

    @property
    def collectives_per_forward(self):
        """{all_to_all, all_reduce} issued per fused forward (counted, not assumed): by the library's RCCL calls or through
        torch.distributed."""
        k = max(self.steps, 1)
        if self.rccl:
            a2a, ar = LibraryRccl.counters()
            return {"all_to_all_single": (a2a - self._rccl0[0]) / k, "all_reduce": (ar - self._rccl0[1]) / k,
                    "issued_by": "library (ncclSend / ncclRecv groups, ncclAllReduce)"}
        return {"all_to_all_single": self.collectives.n_alltoall / k, "all_reduce": self.collectives.n_allreduce / k,
                "issued_by": "torch.distributed (host callbacks)"}

    def step(self, fused: bool = True) -> Tensor:
        """One forward of this rank's shard, graph preparation included (as in the 1-GPU benchmark).
        fused=False runs the per-op Python schedule (ShardProgram) instead of tgnn_forward_sharded."""
        if fused:
            self.steps += 1
            return self.fused.step()
        self.program = ShardProgram(self.net, self.shard, self.backend, inputs=self.inputs)
        return self.comm.run(self.program)


# ----------------------------------------------------------------------------------------------
# the greedy assembly loop on a layout that stays sharded
# ----------------------------------------------------------------------------------------------
def solve_sharded(net, shard: Shard, device, collectives, setup, collide_edge_index: np.ndarray, rccl: "LibraryRccl" = None,
                  min_nodes_per_rank: int = 64, on_round=None, uniform=None):
    """`solve_by_probablistic_greedy` (/root/reference/util/algorithms.py:18-62) for ONE RANK of a node-range sharded layout:
    per round   shard of the unlabelled sub-layout (compact_shard_device: cut locally, on the device)  ->  send lists
    (`setup(shard)`: the communicator's exchange of halo id lists)  ->  score (FusedShardForward: tgnn_forward_sharded)  ->
    all ranks' probabilities gathered (`collectives.allgather`)  ->  the reference's acceptance sweep, run by EVERY rank on the
    same probabilities with the same numpy seed, hence the same decisions everywhere (tilingnn_amd.util.algorithms.HostSweep; it
    needs the layout's collision edge list `collide_edge_index` [2, Ec] in ORIGINAL numbering on every rank's host)  ->  alive mask.
    The empty-edge early-out of ML_Solver.predict (ml_solver.py:31-32: no collision or no adjacency edge left -> every remaining
    tile gets probability 1) is taken when the sub-layout has none on ANY rank (an all-gathered count).
    When the sub-layout has shrunk below world * min_nodes_per_rank nodes, or a rank's range has emptied, the loop stops
    sharding: it returns with `sweep.unlabelled` non-empty and the caller finishes the few remaining rounds on one device
    (finish_on_one_device).  Returns (sweep, rounds): sweep.selection / .order / .unlabelled over the original numbering."""
    from .util.algorithms import HostSweep
    if int(getattr(net, "output_dim", 1)) != 1:
        raise ValueError("solve_sharded scores with probability map 0: the network must have output_dim == 1 (ML_Solver picks the "
                         "map by its loss; the sharded loop has no loss pass)")
    world = shard.world
    n = shard.n_total
    sweep = HostSweep(n, collide_edge_index, uniform)
    inputs = HipBackend(device).upload(shard)
    ids = np.arange(n)
    rounds = 0
    while sweep.unlabelled.any():
        sizes = np.diff(np.asarray(shard.bounds if shard.bounds is not None else even_bounds(shard.n_total, world)))
        if shard.n_total < world * min_nodes_per_rank or np.any(sizes < 2):
            break                                               # (every rank sees the same bounds: a collective decision)
        counts = collectives.allgather(torch.tensor([float(inputs["adj"].shape[1]), float(inputs["col"].shape[1])], device=device))
        tot = torch.stack(list(counts)).sum(0).cpu().tolist()
        if tot[0] == 0 or tot[1] == 0:
            prob = np.ones(shard.n_total)                        # ml_solver.py:31-32
        else:
            fwd = FusedShardForward(net, shard, device, collectives, inputs=inputs, rccl=rccl)
            fwd.two_streams = rccl is not None
            # (ML_Solver picks the probability map by its loss; with the reference's single map that is map 0)
            own = fwd.step()[:, 0].contiguous()
            prob = torch.cat([p.reshape(-1) for p in collectives.allgather(own)]).cpu().numpy()
        if on_round is not None:
            on_round(ids, prob)
        killed = sweep.round(ids, prob)
        rounds += 1
        new_ids = sweep.ids()
        if new_ids.size == 0:
            break
        alive_rel = np.isin(ids, new_ids, assume_unique=True)
        shard, inputs = compact_shard_device(shard, inputs, alive_rel)
        setup(shard)
        refresh_send_idx(shard, inputs)
        ids = new_ids
    return sweep, rounds


def finish_on_one_device(ml_solver, origin_layout, sweep):
    """The tail of a sharded solve: the rounds that are left once the sub-layout is too small to shard, on one device -- the
    single-GPU loop of tilingnn_amd.util.algorithms continued from `sweep`'s state (same RNG stream, same decisions on every rank
    that runs it)."""
    from .util.algorithms import DeviceLayout, SubLayoutBuilder
    origin = origin_layout if isinstance(origin_layout, DeviceLayout) else DeviceLayout.upload(origin_layout, ml_solver.device)
    builder = SubLayoutBuilder(origin)
    dev = origin.node_feature.device
    while sweep.unlabelled.any():
        alive = torch.from_numpy(sweep.unlabelled.astype(np.int32)).to(dev)
        sweep.round(sweep.ids(), ml_solver.predict(builder.build(alive)))
    return sweep
