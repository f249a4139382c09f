"""CPU compute backend for tilingnn_amd.dist.ShardProgram, built on the ORACLE -- test infrastructure only.
It lets the partition / halo-exchange / BatchNorm-all-reduce schedule run under gloo (or LocalSimComm)
without a GPU; the product backend is dist.HipBackend."""
import numpy as np
import torch

from oracle import tilingnn_oracle as orc


def _sums(v):
    v = v.double()
    return torch.cat([v.sum(0), (v * v).sum(0)])


class OracleBackend:
    def __init__(self, dtype=torch.float64, dedup_rows=False):
        # dedup_rows: the edge MLP on the DISTINCT attribute rows only, messages grouped by row class (the oracle's
        # nnconv_mean_dedup, pinned to the port by tests/test_oracle_vs_reference_golden.py) -- what makes a 20 000-node,
        # depth-20 run fit the CPU suite; the default materialises the reference's [Ea, C * C] tensor
        self.dtype, self.device, self.dedup_rows = dtype, torch.device("cpu"), dedup_rows

    def tensor(self, a, dtype):
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.to(self.dtype if dtype == torch.float32 else dtype)

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=self.dtype)

    def upload(self, shard):
        send = shard.send_ids if shard.send_ids is not None else []
        return {"x": self.tensor(shard.x, torch.float32),
                "send_idx": self.tensor(np.concatenate(send) if len(send) else np.empty(0), torch.int32)}

    def prepare(self, shard, inputs):
        return shard, self.tensor(shard.adj_attr, torch.float32)

    def edge_tables(self, graph, attr, net):
        return [None] * net.network_depth

    @staticmethod
    def _act(v, act):
        return {0: lambda t: t, 1: orc.leaky_relu, 2: orc.sigmoid}[act](v)

    def _bn_in(self, a, in_stat):
        return a if in_stat is None else (a - in_stat[0]) * in_stat[1] + in_stat[2]

    def dense(self, a, lin, act, in_stat=None, slot_major=False):
        if slot_major:
            a = torch.cat(list(a), dim=1)
        w, b = lin.weight.detach().to(self.dtype), lin.bias.detach().to(self.dtype)
        out = self._act(self._bn_in(a, in_stat) @ w.t() + b, act)
        return out, _sums(out)

    def nnconv(self, h_rows, shard, wtab, conv, act):
        adj = torch.from_numpy(shard.adj)
        attr = self.tensor(shard.adj_attr, torch.float32)
        layers = list(conv.nn.mlp)
        c = conv.in_channels
        if self.dedup_rows:
            uniq, inv = torch.unique(attr, dim=0, return_inverse=True)
            w = uniq
            for l in layers:
                w = orc.sigmoid(w @ l.linear.weight.detach().to(self.dtype).t() + l.linear.bias.detach().to(self.dtype))
            w = w.view(-1, c, c)
            xs = h_rows[adj[0]]
            msg = torch.zeros(adj.shape[1], c, dtype=self.dtype)
            for t in range(w.shape[0]):
                sel = (inv == t).nonzero().squeeze(1)
                if sel.numel():
                    msg.index_copy_(0, sel, xs.index_select(0, sel) @ w[t])
        else:
            w = attr
            for l in layers:
                w = orc.sigmoid(w @ l.linear.weight.detach().to(self.dtype).t() + l.linear.bias.detach().to(self.dtype))
            msg = torch.matmul(h_rows[adj[0]].unsqueeze(1), w.view(-1, c, c)).squeeze(1)
        agg = torch.zeros(shard.n_own, c, dtype=self.dtype).index_add_(0, adj[1], msg)
        cnt = torch.bincount(adj[1], minlength=shard.n_own).clamp(min=1).to(self.dtype)
        out = agg / cnt[:, None] + h_rows[: shard.n_own] @ conv.root.detach().to(self.dtype) + conv.bias.detach().to(self.dtype)
        out = self._act(out, act)
        return out, _sums(out)

    def gin(self, a_rows, shard, conv, act, in_stat):
        col = torch.from_numpy(shard.col)
        keep = col[0] != col[1]
        src, dst = col[0][keep], col[1][keep]
        x = self._bn_in(a_rows, in_stat)
        z = (1 + conv.eps.detach().to(self.dtype)) * x[: shard.n_own] + \
            torch.zeros(shard.n_own, x.shape[1], dtype=self.dtype).index_add_(0, dst, x[src])
        for l in conv.nn.mlp:
            z = orc.sigmoid(z @ l.linear.weight.detach().to(self.dtype).t() + l.linear.bias.detach().to(self.dtype))
        out = self._act(z, act)
        return out, _sums(out)

    def bn_stat(self, sums, n_total, bn, update_running):
        f = bn.num_features
        mean = sums[:f] / n_total
        var = (sums[f:] / n_total - mean * mean).clamp(min=0)
        g = bn.weight.detach().double() / torch.sqrt(var + bn.eps)
        return (mean.to(self.dtype), g.to(self.dtype), bn.bias.detach().to(self.dtype))

    def bn_apply(self, v, stat):
        return self._bn_in(v, stat)

    def merge(self, a1, stat1, a2, stat2, resid, out):
        r = self._bn_in(a1, stat1) * self._bn_in(a2, stat2)
        out.copy_(r if resid is None else r + resid)

    def pack_rows(self, src, idx, out, col_offset):
        if idx.numel():
            out[: idx.shape[0], col_offset:col_offset + src.shape[1]] = src[idx.long()]
