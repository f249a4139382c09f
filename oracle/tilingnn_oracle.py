"""CPU ORACLE for the TilinGNN graph-conv scoring forward.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (plain torch ops on CPU tensors, fp32 or
fp64) of the algorithm on the hot path named by BASELINE.json `north_star`:

    ML_Solver.predict -> TilinGNN.forward(x, adj_e_index, adj_e_features, col_e_idx)
    (/root/reference/solver/ml_solver/ml_solver.py:29-49,
     /root/reference/graph_networks/networks/TilinGNN.py:51-78)

Who may use it: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` -- as the CHECKER / the timed CPU baseline, never as the thing shipped.
Nothing under `tilingnn_amd/` imports this module; the product path is HIP only and
fails loudly when `libtgnn.so` is missing.

Parity status: PINNED.  The reference has no tests / golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself: `tests/golden/generate_golden.py` imports the reference's
graph_networks/* UNCHANGED in the build container (third-party PyG ops stood in as
described below) and stores its fp64 outputs under `tests/golden/`;
`tests/test_oracle_vs_reference_golden.py` checks this file against them.

Third-party arithmetic that is NOT under /root/reference: PyTorch-Geometric 1.3.2
(README.md:13 "tested with"; no lock file).  Its published semantics, which this
file restates (torch_geometric/nn/conv/{nn_conv,gin_conv,message_passing}.py @1.3.2,
torch_scatter.scatter_mean):
  * flow = source_to_target: x_j = x[edge_index[0]], aggregated at edge_index[1],
    dim_size = N.
  * NNConv(in, out, nn, aggr="mean"): parameters root [in, out], bias [out];
    message = matmul(x_j.unsqueeze(1), nn(edge_attr).view(-1, in, out)).squeeze(1);
    mean = scatter_add / count.clamp(min=1); update = aggr + x @ root + bias.
  * GINConv(nn, eps=0): eps is a BUFFER [1]; edge_index is stripped of self loops;
    out = nn((1 + eps) * x + scatter_add(x_j)).

State-dict key layout follows the reference modules exactly;
every function below takes the flat `sd` dict {key: tensor}.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

Tensor = torch.Tensor
SD = Dict[str, Tensor]

BN_EPS = 1e-5          # torch.nn.BatchNorm1d default (layers/util.py:28, edge_conv.py:22)
BN_MOMENTUM = 0.1      # torch default
LEAKY_SLOPE = 0.01     # torch.nn.LeakyReLU() default (TilinGNN.py:31)


# --------------------------------------------------------------------------------------
# element-wise pieces
# --------------------------------------------------------------------------------------
def leaky_relu(v: Tensor) -> Tensor:
    return torch.where(v >= 0, v, v * LEAKY_SLOPE)


def sigmoid(v: Tensor) -> Tensor:
    return 1.0 / (1.0 + torch.exp(-v))


def batch_norm_train(v: Tensor, sd: SD, prefix: str, update_running: bool = False) -> Tensor:
    """nn.BatchNorm1d in TRAIN mode -- batch statistics over all N rows.

    The reference never calls .eval(): ML_Solver.load_saved_network ends with
    network.train() (ml_solver.py:129-131).  mean / biased variance per column,
    y = gamma * (v - mean) / sqrt(var + eps) + beta; running stats use the UNBIASED
    variance with momentum 0.1 and num_batches_tracked += 1.
    """
    n = v.shape[0]
    mean = v.mean(dim=0)
    var = ((v - mean) ** 2).mean(dim=0)                    # biased
    y = (v - mean) / torch.sqrt(var + BN_EPS) * sd[prefix + ".weight"] + sd[prefix + ".bias"]
    if update_running:
        unbiased = var * (n / max(n - 1, 1))
        sd[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * sd[prefix + ".running_mean"] + BN_MOMENTUM * mean
        sd[prefix + ".running_var"] = (1 - BN_MOMENTUM) * sd[prefix + ".running_var"] + BN_MOMENTUM * unbiased
        sd[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    return y


class reference_ops:
    """Context: the three element-wise pieces above as the modules the REFERENCE calls -- torch.nn.Sigmoid (edge_conv.py:12,17;
    coll_conv.py:17), nn.LeakyReLU (TilinGNN.py:31), nn.BatchNorm1d in train mode (layers/util.py:28,35) -- one fused pass each
    instead of the decomposed forms (1/(1+exp(-v)) is four passes over the [Ea, 1024] tensor of the edge MLP).  Same values to
    fp32 rounding (tests/test_oracle_vs_reference_golden.py); the CHECKER keeps the decomposed forms (they are dtype-generic and
    were pinned in fp64), `bench.py`'s cpu_baseline TIMES this one: it is the reference's op sequence."""

    def __enter__(self):
        import torch.nn.functional as F
        g = globals()
        self._saved = {k: g[k] for k in ("leaky_relu", "sigmoid", "batch_norm_train")}

        def bn(v: Tensor, sd: SD, prefix: str, update_running: bool = False) -> Tensor:
            rm = sd[prefix + ".running_mean"] if update_running else None
            rv = sd[prefix + ".running_var"] if update_running else None
            if update_running:
                sd[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
            return F.batch_norm(v, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"], True, BN_MOMENTUM, BN_EPS)

        g["leaky_relu"] = lambda v: F.leaky_relu(v, LEAKY_SLOPE)
        g["sigmoid"] = torch.sigmoid
        g["batch_norm_train"] = bn
        return self

    def __exit__(self, *exc):
        globals().update(self._saved)
        return False


def linear(v: Tensor, sd: SD, prefix: str) -> Tensor:
    return v @ sd[prefix + ".weight"].t() + sd[prefix + ".bias"]


def linear_trans(v: Tensor, sd: SD, prefix: str, act, bn: bool, update_running: bool = False) -> Tensor:
    """Linear_trans.forward (layers/util.py:31-37): Linear -> activation -> BatchNorm (POST-activation)."""
    v = linear(v, sd, prefix + ".linear")
    if act is not None:
        v = act(v)
    if bn:
        v = batch_norm_train(v, sd, prefix + ".batch_norm", update_running)
    return v


def mlp(v: Tensor, sd: SD, prefix: str, n_layers: int, act, bn: bool, update_running: bool = False) -> Tensor:
    """MLP.forward (layers/util.py:15-17): a stack of Linear_trans; the activation is applied
    on EVERY layer including the last (layers/util.py:10-13)."""
    in_dim = sd[f"{prefix}.mlp.0.linear.weight"].shape[1]
    if v.shape[-1] != in_dim:                              # layers/util.py:16 (assert)
        raise ValueError(f"MLP {prefix}: expected in_dim {in_dim}, got {v.shape[-1]}")
    for i in range(n_layers):
        v = linear_trans(v, sd, f"{prefix}.mlp.{i}", act, bn, update_running)
    return v


# --------------------------------------------------------------------------------------
# message passing pieces (PyG 1.3.2 semantics, see module docstring)
# --------------------------------------------------------------------------------------
def edge_weight_matrices(edge_attr: Tensor, sd: SD, prefix: str, c_in: int, c_out: int) -> Tensor:
    """K1: the per-edge [C_in, C_out] matrices of NNConv -- GraphConv's edge MLP
    (edge_conv.py:17-18: hidden [32, 64], Sigmoid on all three layers, no BN), then
    .view(-1, C_in, C_out): flat index i * C_out + o."""
    w = mlp(edge_attr, sd, prefix, 3, sigmoid, bn=False)
    return w.view(-1, c_in, c_out)


def nnconv_mean(x: Tensor, edge_index: Tensor, edge_attr: Tensor, sd: SD, prefix: str) -> Tensor:
    """K1-K4: NNConv(C, C, mlp, aggr='mean') as called at edge_conv.py:25.

    `prefix` is the GraphConv prefix, e.g. 'brch_1_graph_conv_layers.3'; the edge MLP is
    read from '<prefix>.mlp' (the same tensors are aliased under '<prefix>.nnConv.nn').
    This is the reference's op sequence including the materialised [Ea, C*C] tensor."""
    root = sd[prefix + ".nnConv.root"]                     # [C_in, C_out]
    c_in, c_out = root.shape
    n = x.shape[0]
    src, dst = edge_index[0], edge_index[1]
    w = edge_weight_matrices(edge_attr, sd, prefix + ".mlp", c_in, c_out)          # [Ea, C, C]
    msg = torch.matmul(x.index_select(0, src).unsqueeze(1), w).squeeze(1)          # [Ea, C]
    agg = torch.zeros(n, c_out, dtype=x.dtype).index_add_(0, dst, msg)
    cnt = torch.zeros(n, dtype=x.dtype).index_add_(0, dst, torch.ones_like(dst, dtype=x.dtype))
    agg = agg / cnt.clamp(min=1).unsqueeze(1)
    return agg + x @ root + sd[prefix + ".nnConv.bias"]


def graph_conv(x: Tensor, edge_index: Tensor, edge_attr: Tensor, sd: SD, prefix: str,
               update_running: bool = False, pre: Optional[Tensor] = None) -> Tensor:
    """GraphConv.forward (edge_conv.py:24-30): NNConv -> LeakyReLU -> BatchNorm1d.
    `pre`: the NNConv's output when the caller has it already (tilingnn_forward's capture) -- the same tensor, not recomputed."""
    v = leaky_relu(nnconv_mean(x, edge_index, edge_attr, sd, prefix) if pre is None else pre)
    return batch_norm_train(v, sd, prefix + ".batch_norm", update_running)


def gin_conv(x: Tensor, edge_index: Tensor, sd: SD, prefix: str) -> Tensor:
    """K6-K7: GINConv(nn=MLP(C->32->64->C, Sigmoid x3, no BN)) as called at coll_conv.py:25.
    Self loops are removed; aggregation is a SUM (the ctor's aggr='mean' is ignored,
    coll_conv.py:10,18)."""
    n = x.shape[0]
    src, dst = edge_index[0], edge_index[1]
    keep = src != dst
    src, dst = src[keep], dst[keep]
    agg = torch.zeros(n, x.shape[1], dtype=x.dtype).index_add_(0, dst, x.index_select(0, src))
    z = (1.0 + sd[prefix + ".ginConv.eps"].to(x.dtype)) * x + agg
    return mlp(z, sd, prefix + ".ginConv.nn", 3, sigmoid, bn=False)


def coll_conv(x: Tensor, edge_index: Tensor, sd: SD, prefix: str, update_running: bool = False,
              pre: Optional[Tensor] = None) -> Tensor:
    """CollConv.forward (coll_conv.py:24-30): GINConv -> LeakyReLU -> BatchNorm1d.  `pre`: see graph_conv."""
    v = leaky_relu(gin_conv(x, edge_index, sd, prefix) if pre is None else pre)
    return batch_norm_train(v, sd, prefix + ".batch_norm", update_running)


# --------------------------------------------------------------------------------------
# the network
# --------------------------------------------------------------------------------------
def network_depth_of(sd: SD) -> int:
    d = 0
    while f"brch_1_graph_conv_layers.{d}.nnConv.root" in sd:
        d += 1
    return d


def init_node_feature_trans(x: Tensor, sd: SD, update_running: bool = False) -> Tensor:
    """K10 (TilinGNN.py:31,54): MLP[Fx -> C -> C], LeakyReLU, BN on both layers."""
    return mlp(x, sd, "init_node_feature_trans", 2, leaky_relu, bn=True, update_running=update_running)


def final_mlp(cat: Tensor, sd: SD, update_running: bool = False) -> Tensor:
    """K11 (TilinGNN.py:45-48,76): MLP(672 -> 256 -> 128 -> 64 -> 32, LeakyReLU, BN) then
    Linear_trans(32 -> out, Sigmoid, no BN)."""
    v = mlp(cat, sd, "final_mlp.0", 4, leaky_relu, bn=True, update_running=update_running)
    return linear_trans(v, sd, "final_mlp.1", sigmoid, bn=False)


def tilingnn_forward(sd: SD, x: Tensor, adj_e_index: Tensor, adj_e_features: Tensor, col_e_idx: Tensor,
                     col_e_features: Optional[Tensor] = None, update_running: bool = False,
                     capture: Optional[dict] = None) -> Tuple[Tensor, Tensor]:
    """TilinGNN.forward (TilinGNN.py:51-78).  Returns (probs [N, out_dim], adj_e_features).

    `capture`, when a dict, receives the intermediate tensors used for teacher-forced
    per-op parity tests: 'init', then per layer i 'h1_in.i', 'h2_in.i', 'nnconv.i' (pre-activation),
    'gconv.i' (post-BN), 'gin.i' (pre-activation), 'cconv.i' (post-BN), 'mid.i+1', then 'cat', 'probs'.
    `col_e_features` is accepted and ignored exactly as the reference does (TilinGNN.py:51)."""
    depth = network_depth_of(sd)
    residual_skip_num = 2                                   # TilinGNN.py:25
    h1 = init_node_feature_trans(x, sd, update_running)     # TilinGNN.py:54
    h2 = h1                                                 # :55
    middle: List[Tensor] = [h1]                             # :58
    if capture is not None:
        capture["init"] = h1
    for i in range(depth):                                  # :59-71
        p1 = f"brch_1_graph_conv_layers.{i}"
        p2 = f"brch_2_coll_conv_layers.{i}"
        pre1 = pre2 = None
        if capture is not None:                             # (the two convolutions once: their outputs are captured AND passed on)
            capture[f"h1_in.{i}"], capture[f"h2_in.{i}"] = h1, h2
            pre1 = capture[f"nnconv.{i}"] = nnconv_mean(h1, adj_e_index, adj_e_features, sd, p1)
            pre2 = capture[f"gin.{i}"] = gin_conv(h2, col_e_idx, sd, p2)
        g1 = graph_conv(h1, adj_e_index, adj_e_features, sd, p1, update_running, pre1)    # :62
        h2 = coll_conv(h2, col_e_idx, sd, p2, update_running, pre2)                       # :63
        h1 = g1 * h2                                                                # :64
        if i - residual_skip_num >= 0:                                              # :67-69
            h1 = h1 + middle[i - residual_skip_num]
        middle.append(h1)                                                           # :71
        if capture is not None:
            capture[f"gconv.{i}"], capture[f"cconv.{i}"], capture[f"mid.{i + 1}"] = g1, h2, h1
    cat = torch.cat(middle, dim=1)                                                  # :74
    probs = final_mlp(cat, sd, update_running)                                      # :76
    if capture is not None:
        capture["cat"], capture["probs"] = cat, probs
    return probs, adj_e_features                                                    # :78


# --------------------------------------------------------------------------------------
# the caller's loss (ml_solver.py:46,133-136 -> losses.py:48-116); used by predict's best-map pick
# --------------------------------------------------------------------------------------
LOSS_EPS = 1e-7                                            # losses.py:10


def unsupervised_losses(probs: Tensor, node_feature: Tensor, collide_edge_index: Tensor,
                        adj_edges_index: Tensor, adj_edge_features: Tensor,
                        collision_weight: float = 1.0 / math.log(1.0 + 1e-1),
                        align_length_weight: float = 0.02, avg_area_weight: float = 1.0) -> Tensor:
    """Losses.calculate_unsupervised_loss (losses.py:48-116), per probability map, with the
    weights of inputs/config.py:49-51.  Returns the loss vector [M]; predict picks
    argsort(losses)[0] (ml_solver.py:133-136)."""
    m = probs.shape[1]
    e_col = collide_edge_index.shape[1] if collide_edge_index.numel() > 0 else 0
    e_adj = adj_edges_index.shape[1] if adj_edges_index.numel() > 0 else 0
    zero = torch.zeros((), dtype=probs.dtype)
    out = []
    for k in range(m):
        p = probs[:, k]
        # average node area (losses.py:65-67): log(clamp(mean(area_ratio * p), eps))
        loss_area = torch.log(torch.clamp(torch.mean(node_feature[:, -1] * p), min=LOSS_EPS))
        # collision feasibility (losses.py:69-81)
        if e_col > 0:
            pp = torch.clamp(p[collide_edge_index[0]] * p[collide_edge_index[1]], min=LOSS_EPS, max=1 - LOSS_EPS)
            loss_feas = torch.log(1 - pp).sum() / e_col
        else:
            loss_feas = zero
        # alignment length (losses.py:83-98)
        if e_adj > 0:
            pp = torch.clamp(p[adj_edges_index[0]] * p[adj_edges_index[1]] * adj_edge_features[:, 1], min=LOSS_EPS)
            loss_align = (torch.log(pp) / math.log(10)).sum() / e_adj
        else:
            loss_align = zero
        out.append((1 - avg_area_weight * loss_area) * (1 - collision_weight * loss_feas)
                   * (1 - align_length_weight * loss_align))                        # losses.py:104-106
    return torch.stack(out)


def training_step_grads(sd: SD, x: Tensor, adj_e_index: Tensor, adj_e_features: Tensor, col_e_idx: Tensor):
    """One training step of Trainer.train (trainer.py:68-84) without the optimizer: forward in train mode, the
    unsupervised loss (min over the probability maps, losses.py:108-109), backward -- by autograd over the
    restatement above.  Returns (probs, loss, d loss / d probs, {key: gradient}) for every floating-point entry of
    `sd` that is a parameter (BatchNorm running statistics and GIN's eps buffer are not).  Pinned against
    tests/golden/ref_grads.npz."""
    leaf = {}
    for k, v in sd.items():
        is_param = v.is_floating_point() and not k.endswith(("running_mean", "running_var", ".eps"))
        leaf[k] = v.detach().clone().requires_grad_(True) if is_param else v.detach().clone()
    probs, _ = tilingnn_forward(leaf, x, adj_e_index, adj_e_features, col_e_idx)
    probs.retain_grad()
    loss = unsupervised_losses(probs, x, col_e_idx, adj_e_index, adj_e_features).min()
    loss.backward()
    grads = {}
    for k, v in leaf.items():
        if v.requires_grad and ".nnConv.nn." not in k:            # aliases of '<prefix>.mlp.*' (same tensors in the reference)
            grads[k] = v.grad.detach() if v.grad is not None else torch.zeros_like(v)
    return probs.detach(), loss.detach(), probs.grad.detach(), grads


# --------------------------------------------------------------------------------------
# helpers for tests / bench
# --------------------------------------------------------------------------------------
def cast_sd(sd: SD, dtype: torch.dtype) -> SD:
    """Copy a state dict to `dtype` (integer entries such as num_batches_tracked are kept)."""
    return {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.detach().clone()) for k, v in sd.items()}


def rel_max_err(a: Tensor, b: Tensor) -> float:
    """max-norm relative error  max|a-b| / max|b|  (the parity metric of SURVEY.md section 8c)."""
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


# --------------------------------------------------------------------------------------
# NNConv at sizes where the reference's materialised [Ea, C, C] tensor does not fit the test box (fp64 at 1M edges:
# 8 GB): the same arithmetic with the edge MLP evaluated once per DISTINCT attribute row and the messages formed in
# chunks of edges.  Algebraically identical (W_e depends on the attribute row only); pinned equal to `nnconv_mean`
# on the labyrinth graph by tests/test_oracle_vs_reference_golden.py::test_dedup_nnconv_equals_the_port.
# --------------------------------------------------------------------------------------
def nnconv_mean_dedup(x: Tensor, edge_index: Tensor, edge_attr: Tensor, sd: SD, prefix: str, chunk: int = 1 << 18) -> Tensor:
    root = sd[prefix + ".nnConv.root"]
    c_in, c_out = root.shape
    n = x.shape[0]
    src, dst = edge_index[0], edge_index[1]
    uniq, inv = torch.unique(edge_attr, dim=0, return_inverse=True)
    w = edge_weight_matrices(uniq.to(x.dtype), sd, prefix + ".mlp", c_in, c_out)     # [T, C, C]
    agg = torch.zeros(n, c_out, dtype=x.dtype)
    for e0 in range(0, int(src.shape[0]), chunk):
        e1 = min(e0 + chunk, int(src.shape[0]))
        xs = x.index_select(0, src[e0:e1])
        msg = torch.zeros(e1 - e0, c_out, dtype=x.dtype)
        for t in range(w.shape[0]):                                                  # grouped by type: [m, C] @ [C, C]
            sel = (inv[e0:e1] == t).nonzero().squeeze(1)
            if sel.numel():
                msg.index_copy_(0, sel, xs.index_select(0, sel) @ w[t])
        agg.index_add_(0, dst[e0:e1], msg)
    cnt = torch.zeros(n, dtype=x.dtype).index_add_(0, dst, torch.ones_like(dst, dtype=x.dtype))
    agg = agg / cnt.clamp(min=1).unsqueeze(1)
    return agg + x @ root + sd[prefix + ".nnConv.bias"]
