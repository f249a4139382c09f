"""The training step on the GPU: every adjoint kernel against float64 torch on the same inputs,
the composed NNConv / GIN adjoints against autograd over the oracle, and the whole step against the gradients of the
REFERENCE's own network + loss (tests/golden/ref_grads.npz) with the reference's own float32 run as the yardstick."""
import numpy as np
import pytest
import torch

from oracle import tilingnn_oracle as orc
from tests.golden_util import graph_tensors, load_labyrinth_graph, load_npz

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return float((got - want).abs().max() / want.abs().max().clamp(min=1e-30))


def _rng(seed):
    g = torch.Generator().manual_seed(seed)
    return lambda *shape: torch.randn(*shape, generator=g)


# ---------------------------------------------------------------------------------------------- single kernels
@pytest.mark.parametrize("n,c", [(1, 5), (1000, 32), (4097, 256), (13, 1024)])
def test_colsum(n, c):
    from tilingnn_amd import train
    x = _rng(n)(n, c).to(DEV)
    assert _rel(train.colsum(x), x.double().sum(0)) < 1e-6
    wide = _rng(n + 1)(n, c + 8).to(DEV)
    assert _rel(train.colsum(wide[:, :c]), wide[:, :c].double().sum(0)) < 1e-6           # row stride > width


@pytest.mark.parametrize("n,cout,cin", [(1000, 32, 32), (13, 1024, 64), (777, 1, 32), (300, 32, 3), (4097, 448, 32),
                                        (2500, 256, 128), (5, 64, 64)])
def test_wgrad(n, cout, cin):
    from tilingnn_amd import train
    r = _rng(n + cout)
    dz, x = r(n, cout).to(DEV), r(n, cin).to(DEV)
    assert _rel(train.wgrad(dz, x), dz.double().t() @ x.double()) < 2e-6
    w, b = train.wgrad(dz, x, with_bias=True)                             # the bias gradient rides on the same pass
    assert _rel(w, dz.double().t() @ x.double()) < 2e-6 and _rel(b, dz.double().sum(0)) < 2e-6


def test_wgrad_slot_major_is_the_concatenation():
    from tilingnn_amd import train
    r = _rng(3)
    skip, dz = r(21, 3001, 32).to(DEV), r(3001, 256).to(DEV)
    cat = torch.cat(list(skip), dim=1)                                     # TilinGNN.py:74
    assert _rel(train.wgrad(dz, skip, slot_major=True), dz.double().t() @ cat.double()) < 2e-6


def _bn_stat(a, bn):
    from tilingnn_amd import ops
    sums = torch.stack([a.double().sum(0), (a.double() ** 2).sum(0)]).contiguous()
    return ops.bn_stat_from_sums(sums, int(a.shape[0]), bn, update_running=False)


def _bn(f, seed):
    bn = torch.nn.BatchNorm1d(f)
    r = _rng(seed)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * r(f))
        bn.bias.copy_(0.2 * r(f))
    return bn.to(DEV)


def _bn_ref(z, bn, leaky):
    a = torch.where(z > 0, z, 0.01 * z) if leaky else z
    mean, var = a.mean(0), a.var(0, unbiased=False)
    return (a - mean) / torch.sqrt(var + bn.eps) * bn.weight.detach().double() + bn.bias.detach().double()


@pytest.mark.parametrize("n,f,leaky", [(1000, 32, True), (5000, 256, True), (333, 64, False), (2, 32, True)])
def test_batchnorm_backward(n, f, leaky):
    from tilingnn_amd import ops, train
    r = _rng(n + f)
    z = (r(n, f) + 0.3).to(DEV)
    a = torch.where(z > 0, z, 0.01 * z) if leaky else z
    dy, scale = r(n, f).to(DEV), (r(n).abs() + 0.5).to(DEV)
    bn = _bn(f, 7)
    zz = z.double().requires_grad_(True)
    gam, bet = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    aa = torch.where(zz > 0, zz, 0.01 * zz) if leaky else zz
    y = (aa - aa.mean(0)) / torch.sqrt(aa.var(0, unbiased=False) + bn.eps) * gam + bet
    (y * dy.double()).sum().backward()
    dz, scaled, dgamma, dbeta = train.bn_bwd(dy, a, _bn_stat(a, bn), bn.eps, ops.ACT_LEAKY_RELU if leaky else ops.ACT_NONE,
                                             row_scale=scale)
    assert _rel(dz, zz.grad) < 1e-5 and _rel(dgamma, gam.grad) < 1e-5 and _rel(dbeta, bet.grad) < 1e-5
    assert _rel(scaled, zz.grad * scale.double()[:, None]) < 1e-5


@pytest.mark.parametrize("n,with_resid,with_carry", [(1000, True, True), (4099, False, False), (37, True, False)])
def test_merge_backward(n, with_resid, with_carry):
    from tilingnn_amd import ops, train
    from tilingnn_amd._lib import check, lib, ptr
    r = _rng(n)
    z1, z2 = (r(n, 32) + 0.2).to(DEV), (r(n, 32) + 0.2).to(DEV)
    a1, a2 = torch.where(z1 > 0, z1, 0.01 * z1), torch.where(z2 > 0, z2, 0.01 * z2)
    bn1, bn2 = _bn(32, 1), _bn(32, 2)
    st1, st2 = _bn_stat(a1, bn1), _bn_stat(a2, bn2)
    dcat = r(n, 96).to(DEV)                                               # dh lives in slot 2, the residual in slot 0
    before = dcat.clone()
    carry = r(n, 32).to(DEV) if with_carry else None
    q1, q2 = z1.double().requires_grad_(True), z2.double().requires_grad_(True)
    y1, y2 = _bn_ref(q1, bn1, True), _bn_ref(q2, bn2, True)
    obj = (y1 * y2 * before[:, 64:].double()).sum()
    if with_carry:
        obj = obj + (y2 * carry.double()).sum()
    obj.backward()
    dy1, dy2 = torch.empty(n, 32, device=DEV), torch.empty(n, 32, device=DEV)
    coef, dgb = torch.empty(2, 2, 32, device=DEV), torch.empty(4, 32, device=DEV)
    nb = lib.tgnn_reduce_workspace_bytes(32)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    check(lib.tgnn_merge_bwd_reduce(ptr(dcat[:, 64:]), 96, ptr(a1), ptr(st1), ptr(a2), ptr(st2), ptr(carry), n, 32,
                                    bn1.eps, bn2.eps, ptr(dy1), ptr(dy2), ptr(dcat) if with_resid else None, 96,
                                    ptr(coef[0]), ptr(dgb[0]), ptr(dgb[1]), ptr(coef[1]), ptr(dgb[2]), ptr(dgb[3]), ptr(ws),
                                    nb, train._s(dy1)))
    dz1, _ = train.bn_bwd_apply(dy1, a1, st1, coef[0], ops.ACT_LEAKY_RELU)
    dz2, _ = train.bn_bwd_apply(dy2, a2, st2, coef[1], ops.ACT_LEAKY_RELU)
    assert _rel(dz1, q1.grad) < 1e-5 and _rel(dz2, q2.grad) < 1e-5
    want_slot0 = before[:, :32] + before[:, 64:] if with_resid else before[:, :32]
    assert torch.equal(dcat[:, :32], want_slot0) and torch.equal(dcat[:, 32:], before[:, 32:])


def _small():
    z = load_npz("ref_ops_small.npz")
    return dict(x=z["x"], adj=z["adj"].astype(np.int64), adj_attr=z["adj_attr"], col=z["col"].astype(np.int64),
                col_attr=z["col_attr"])


def _tiny():
    z = load_npz("tiny_graph.npz")
    return dict(x=z["x"], adj=z["adj"], adj_attr=z["adj_attr"], col=z["col"], col_attr=z["col_attr"])


def _net(fe, depth, seed):
    from tilingnn_amd.graph_networks.networks.TilinGNN import TilinGNN
    from tilingnn_amd.weights import make_state_dict
    net = TilinGNN(adj_edge_features_dim=fe, network_depth=depth, network_width=32, node_features_dim=3)
    sd = make_state_dict(fe, depth, 32, 1, 3, seed=seed)
    net.load_state_dict(sd)
    return net.to(DEV).train(), sd


def test_type_sum_and_transposed_graph():
    from tilingnn_amd import ops, train
    g = _small()
    x, adj, attr, col, _ = graph_tensors(g, torch.float32, DEV)
    n = int(x.shape[0])
    graph = ops.prepare_graph(n, adj, attr, col)
    tg = train.TrainGraph(graph, adj, col)
    T = graph.n_types
    h = _rng(5)(n, 32).to(DEV)
    et = graph.edge_type[:adj.shape[1]].long()
    for rowptr, src, typ, gather_from, scatter_to in ((graph.adj_rowptr, graph.adj_src, graph.adj_type, adj[0], adj[1]),
                                                      (tg.adjT_rowptr, tg.adjT_src, tg.adjT_type, adj[1], adj[0])):
        got = train.type_sum(h, h, tg.deg, rowptr, src, typ, n, T).view(n, T + 1, 32)
        want = torch.zeros(n, T + 1, 32, dtype=torch.float64, device=DEV)
        want.view(n * (T + 1), 32).index_add_(0, scatter_to * (T + 1) + et, h.double()[gather_from])
        deg = torch.bincount(adj[1], minlength=n).clamp(min=1).double()
        want[:, T] = h.double() * deg[:, None]
        assert _rel(got, want) < 1e-6
    assert torch.equal(tg.deg.cpu(), torch.bincount(adj[1].cpu(), minlength=n).clamp(min=1).float())


@pytest.mark.parametrize("case", ["small", "tiny"])
def test_nnconv_and_gin_adjoints_teacher_forced(case):
    """Same inputs, same upstream gradient: the composed adjoints against float64 autograd over the oracle's ops."""
    from tilingnn_amd import ops, train
    g, fe, seed = (_small(), 15, 5) if case == "small" else (_tiny(), 6, 3)
    net, sd = _net(fe, 3, seed)
    sd64 = orc.cast_sd(sd, torch.float64)
    x, adj, attr, col, _ = graph_tensors(g, torch.float32, DEV)
    n = int(x.shape[0])
    tg = train.TrainGraph(ops.prepare_graph(n, adj, attr, col), adj, col)
    T = tg.g.n_types
    r = _rng(11)
    h, dz = r(n, 32).to(DEV), r(n, 32).to(DEV)
    l1, l2 = net.brch_1_graph_conv_layers[1], net.brch_2_coll_conv_layers[1]
    p1, p2 = "brch_1_graph_conv_layers.1", "brch_2_coll_conv_layers.1"

    # ---- NNConv
    wtab = ops.edge_weight_table(attr, tg.g, *l1.nnConv._edge_mlp_params(), 32).contiguous()
    grads = {}
    dh = train.nnconv_backward(l1.nnConv, p1, tg, wtab, h, dz, dz * tg.inv_deg[:, None], attr, grads)
    leaf = {k: (v.clone().requires_grad_(True) if k.startswith(p1) and v.is_floating_point() else v) for k, v in sd64.items()}
    hh = h.double().cpu().requires_grad_(True)
    out = orc.nnconv_mean(hh, adj.cpu(), attr.double().cpu(), leaf, p1)
    (out * dz.double().cpu()).sum().backward()
    assert _rel(dh, hh.grad) < 1e-5
    for k, v in grads.items():
        assert _rel(v.reshape(leaf[k].shape), leaf[k].grad) < 1e-5, k
    assert {k for k in grads} == {k for k in leaf if k.startswith(p1) and leaf[k].requires_grad and leaf[k].grad is not None
                                  and ".nnConv.nn." not in k}

    # ---- GIN
    leaf = {k: (v.clone().requires_grad_(True) if k.startswith(p2) and v.is_floating_point() and not k.endswith(".eps")
                else v) for k, v in sd64.items()}
    out = orc.gin_conv(hh := h.double().cpu().requires_grad_(True), col.cpu(), leaf, p2)
    (out * dz.double().cpu()).sum().backward()
    u = train.gin_aggregate(h, tg.g.col_rowptr, tg.g.col_src, l2.ginConv.eps, n)
    gw = l2.ginConv._mlp_params()
    t1 = ops.dense_act(u, gw[0], gw[1], ops.ACT_SIGMOID)[0]
    t3 = ops.dense_act(ops.dense_act(t1, gw[2], gw[3], ops.ACT_SIGMOID)[0], gw[4], gw[5], ops.ACT_SIGMOID)[0]
    grads = {}
    dh2 = train.gin_backward(l2.ginConv, p2, tg, u, t3, dz, grads)
    assert _rel(dh2, hh.grad) < 1e-5
    for k, v in grads.items():
        assert _rel(v.reshape(leaf[k].shape), leaf[k].grad) < 1e-5, k


@pytest.mark.parametrize("maps", [1, 3])
def test_loss_backward(maps):
    from tilingnn_amd.solver.ml_solver.losses import Losses
    g = load_labyrinth_graph()
    x, adj, attr, col, _ = graph_tensors(g, torch.float32, DEV)
    n = int(x.shape[0])
    p = (0.02 + 0.96 * torch.rand(n, maps, generator=torch.Generator().manual_seed(maps))).to(DEV).requires_grad_(True)
    loss, min_index, losses = Losses.calculate_unsupervised_loss(p, x, col, adj, attr)
    (3.0 * loss).backward()
    q = p.detach().double().cpu().requires_grad_(True)
    ref = orc.unsupervised_losses(q, x.double().cpu(), col.cpu(), adj.cpu(), attr.double().cpu())
    (3.0 * ref.min()).backward()
    assert int(min_index) == int(ref.argmin()) and abs(float(loss.detach()) - float(ref.min())) < 1e-5 * float(ref.min())
    assert _rel(p.grad, q.grad) < 1e-5
    if maps > 1:
        others = [k for k in range(maps) if k != int(min_index)]
        assert float(p.grad[:, others].abs().max()) == 0.0              # torch.min routes the gradient to the arg-min map


def test_loss_backward_respects_the_clamps():
    from tilingnn_amd.solver.ml_solver.losses import Losses
    g = _tiny()
    x, adj, attr, col, _ = graph_tensors(g, torch.float32, DEV)
    p = torch.tensor([[1.0], [1.0], [1e-5], [0.9], [0.0], [0.5]], device=DEV).requires_grad_(True)   # pp = 1, tiny and 0
    loss, _, _ = Losses.calculate_unsupervised_loss(p, x, col, adj, attr)
    loss.backward()
    q = p.detach().double().cpu().requires_grad_(True)
    orc.unsupervised_losses(q, x.double().cpu(), col.cpu(), adj.cpu(), attr.double().cpu()).min().backward()
    assert torch.isfinite(p.grad).all() and _rel(p.grad, q.grad) < 1e-5


# ---------------------------------------------------------------------------------------------- the whole step
def _hip_step(g, fe, depth, seed):
    from tilingnn_amd.solver.ml_solver.losses import Losses
    net, _ = _net(fe, depth, seed)
    net.autograd = True
    x, adj, attr, col, _ = graph_tensors(g, torch.float32, DEV)
    probs, _ = net(x, adj, attr, col)
    probs.retain_grad()
    loss, _, _ = Losses.calculate_unsupervised_loss(probs, x, col, adj, attr)
    loss.backward()
    return net, probs, loss


def _stats(name, v):
    seed = int.from_bytes(name.encode(), "little") % (2 ** 31)
    proj = np.random.default_rng(seed).standard_normal(v.shape)
    return np.array([v.sum(), np.sqrt((v ** 2).sum()), (v * proj).sum()])


@pytest.mark.parametrize("case,fe,depth,seed,slack", [("small", 15, 3, 5, 4), ("tiny", 6, 3, 3, 10), ("laby", 15, 20, 0, 4)])
def test_training_step_against_the_reference(case, fe, depth, seed, slack):
    """Every parameter gradient of one training step against the reference's float64 step.  Gradients through 3..20
    train-mode BatchNorms amplify rounding (the reference's own float32 step is off by up to 2 % at depth 3 and by more
    than 50 % at depth 20, ref_grads.npz: err32): the bound per tensor is `slack` x that yardstick, not a fixed epsilon
    (the yardstick is ONE float32 sample per tensor; the 6-node graph, BatchNorm over 6 rows, gets the widest slack)."""
    ref = load_npz("ref_grads.npz")
    g = {"small": _small, "tiny": _tiny, "laby": load_labyrinth_graph}[case]()
    net, probs, loss = _hip_step(g, fe, depth, seed)
    assert abs(float(loss.detach()) - float(ref[f"{case}.loss"])) <= 3 * abs(float(ref[f"{case}.loss32"]) - float(ref[f"{case}.loss"])) + 1e-5
    names = [k for k, _ in net.named_parameters()]
    assert sorted(names) == sorted(k[len(case) + 7:] for k in ref.files if k.startswith(f"{case}.err32."))
    err32 = {k: float(ref[f"{case}.err32.{k}"]) for k in names}
    floor = float(np.median(list(err32.values())))
    # stats = (sum, L2 norm, projection on N(0,1)): natural scales (norm sqrt(size), norm, norm)
    scale = {k: ref[f"{case}.stat.{k}"][1] * np.array([np.sqrt(ref[f"{case}.err32stat.{k}"].size and p.numel()), 1.0, 1.0])
             for k, p in net.named_parameters() if f"{case}.stat.{k}" in ref.files}
    rel_spread = {k: float((ref[f"{case}.err32stat.{k}"] / np.maximum(scale[k], 1e-300)).max()) for k in scale}
    stat_floor = float(np.median(list(rel_spread.values()))) if rel_spread else 0.0
    hip_err = []
    for k, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        got = p.grad.double().cpu().numpy()
        if f"{case}.grad.{k}" in ref.files:
            want = ref[f"{case}.grad.{k}"].astype(np.float64)
            e = np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)
            hip_err.append(e)
            assert e <= slack * max(err32[k], floor) + 1e-5, (k, e, err32[k])
        want_stat = ref[f"{case}.stat.{k}"] if f"{case}.stat.{k}" in ref.files else None
        if want_stat is not None:
            # the float32 noise on this tensor's sums: its own sample, or the median over all tensors if that is larger
            bound = (slack * max(rel_spread[k], stat_floor) + 1e-4) * scale[k]
            assert np.all(np.abs(_stats(k, got) - want_stat) <= bound), (k, np.abs(_stats(k, got) - want_stat), bound)
    assert np.median(hip_err) <= slack / 2 * floor + 1e-6, (np.median(hip_err), floor)


def test_autograd_switch_and_optimizer_step():
    """`network.autograd` off: the inference forward, no graph.  On: every parameter receives a gradient, BatchNorm
    statistics advance once per step, and the caller's optimizer (network_train.py hands Adam to Trainer.train) lowers
    the loss."""
    from tilingnn_amd.solver.ml_solver.losses import Losses
    net, _ = _net(15, 3, 5)
    x, adj, attr, col, _ = graph_tensors(_small(), torch.float32, DEV)
    probs, _ = net(x, adj, attr, col)
    assert not probs.requires_grad
    net.autograd = True
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        probs, _ = net(x, adj, attr, col)
        assert probs.requires_grad
        loss, _, _ = Losses.calculate_unsupervised_loss(probs, x, col, adj, attr)
        loss.backward()
        assert all(p.grad is not None for p in net.parameters())
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]
    assert int(net.brch_1_graph_conv_layers[0].batch_norm.num_batches_tracked) == 7        # 1 inference + 6 steps
    net.eval()
    with torch.no_grad():
        probs, _ = net(x, adj, attr, col)                                   # eval mode never takes the training path
    assert not probs.requires_grad


def test_trainer_loop_on_layout_files(tmp_path):
    """Trainer.train over layout files on disk: losses fall, checkpoints are written and load back."""
    import os
    from tests.golden_util import GOLDEN
    from tilingnn_amd.solver.ml_solver.ml_solver import ML_Solver
    from tilingnn_amd.solver.ml_solver.trainer import Trainer
    from tilingnn_amd.tiling.tile_graph import TileGraph
    from tilingnn_amd.util import data_util as du
    graph = TileGraph(2)
    graph.load_graph_state(os.path.join(GOLDEN, "complete_graph_small.pkl"), sidecar=False)
    rng = np.random.default_rng(0)
    for split, count in (("train", 4), ("test", 2)):
        os.makedirs(tmp_path / split, exist_ok=True)
        for i in range(count):
            tiles = sorted(int(v) for v in rng.choice(150, size=int(rng.integers(60, 140)), replace=False))
            x, ci, cf, ai, af, re_index = du.create_brick_layout_from_super_set(graph, tiles)
            du.write_brick_layout_data(f"layout_{i}.pkl", re_index, node_features=x, collide_edge_index=ci,
                                       collide_edge_features=cf, align_edge_index=ai, align_edge_features=af,
                                       prefix=str(tmp_path / split / "raw"))
    net, _ = _net(15, 3, 5)
    solver = ML_Solver(None, DEV, graph, net, num_prob_maps=1)
    opt = torch.optim.Adam(net.parameters(), lr=2e-3)
    trainer = Trainer(None, None, DEV, net, str(tmp_path))
    history = trainer.train(solver, opt, batch_size=1, training_epoch=4, save_model_per_epoch=2, shuffle_seed=1,
                            log=lambda *_: None)
    assert len(history) == 4 and history[-1][0] < history[0][0] and all(np.isfinite(h).all() for h in history)
    saved = sorted(os.listdir(tmp_path / "model"))
    assert any(f.startswith("model_0_") for f in saved) and any(f.startswith("optimizer_0_") for f in saved)
    assert not net.autograd                                              # switched off again after the steps
    with pytest.raises(NotImplementedError):
        trainer.train(solver, opt, batch_size=4)


# Which parameters a float32 run of this ill-conditioned step gets wrong by how much is a lottery of the rounding realisation
# (scratch/grad_ratio.py, ten seeds, two builds that differ in the summation order of the first layer: the worst per-parameter
# ratio against the float32 oracle is 0.1 .. 1.3 on seven seeds and 13 .. 114 on the other three -- and WHICH three changes with the
# build; the float32 oracle's own worst parameter spans 3e-4 .. 8.5e-2 over the seeds).  A seed draws a bad realisation with
# probability ~0.3, independently of the others and anew with every build.  So the gate is over NINE seeds and asks for what a
# correct adjoint delivers with probability 0.996 whatever the draw: at least THREE seeds whose worst parameter is within 4 x of
# the float32 oracle's own error (a rule on the median would fail one build in ten by the draw alone), and no seed off by more
# than the float32 oracle itself can be.  A wrong adjoint -- a systematic relative error of 1e-3 on one parameter is a ratio of
# ~1 000 -- is off on EVERY seed and fails both.
GRAD_SEEDS = (1, 2, 3, 4, 5, 6, 7, 8, 9)


def test_training_step_with_many_edge_types():
    """25 distinct edge-attribute rows: more than the matrix-core NNConv kernel's weight image holds, so the forward of the
    step runs on the CSR / LDS-table kernel; the adjoints (type sums with 26 slots) must not care.  Depth 2, so that float32
    rounding stays small against the float64 autograd of the oracle.

    The yardstick is the same step by the oracle in float32 -- but the collision branch's BatchNorm (columns that vary by
    ~1 % of their value) amplifies rounding by ~1e4 and LeakyReLU kinks turn last-bit differences into discrete ones: see the
    note above GRAD_SEEDS for what that does to any single seed and for the gate."""
    from tilingnn_amd.graph_networks.networks.TilinGNN import TilinGNN
    from tilingnn_amd.solver.ml_solver.losses import Losses
    from tilingnn_amd.synth import make_super_graph
    from tilingnn_amd.weights import make_state_dict
    sg = make_super_graph(600, 6000, 7500, tile_count=2, n_edge_types=25, seed=9)
    fe = 2 + 25
    x, adj, attr, col, _ = sg.to_torch(DEV)
    torch.set_num_threads(8)
    worst = []
    for seed in GRAD_SEEDS:
        net = TilinGNN(adj_edge_features_dim=fe, network_depth=2, network_width=32, node_features_dim=3)
        sd = make_state_dict(fe, 2, 32, 1, 3, seed=seed)
        net.load_state_dict(sd)
        net = net.to(DEV).train()
        net.autograd = True
        probs, _ = net(x, adj, attr, col)
        loss, _, _ = Losses.calculate_unsupervised_loss(probs, x, col, adj, attr)
        loss.backward()
        _, ref_loss, _, ref_grads = orc.training_step_grads(orc.cast_sd(sd, torch.float64), x.double().cpu(), adj.cpu(),
                                                            attr.double().cpu(), col.cpu())
        assert abs(float(loss.detach()) - float(ref_loss)) < 1e-4 * float(ref_loss)
        _, _, _, f32_grads = orc.training_step_grads(orc.cast_sd(sd, torch.float32), x.cpu(), adj.cpu(), attr.cpu(), col.cpu())
        err32 = {k: _rel(f32_grads[k], ref_grads[k]) for k in ref_grads}
        floor = float(np.median(list(err32.values())))
        errs = {k: _rel(p.grad, ref_grads[k]) for k, p in net.named_parameters()}
        assert set(errs) == set(err32)
        print(f"seed {seed}: worst parameter ours {max(errs.values()):.2e}, float32 oracle's worst {max(err32.values()):.2e}, "
              f"float32 oracle's median {floor:.2e}")
        assert max(errs.values()) < max(0.06, 2.0 * max(err32.values())), max(errs.items(), key=lambda kv: kv[1])
        worst.append(max(e / (max(err32[k], floor) + 2.5e-6) for k, e in errs.items()))
    print("worst parameter, ours / float32 oracle, per seed:", [f"{w:.1f}" for w in worst])
    print(f"median {float(np.median(worst)):.1f}, third best {sorted(worst)[2]:.1f}")
    assert sorted(worst)[2] <= 4.0, worst


@pytest.mark.parametrize("case,fe,depth,seed", [("small", 15, 3, 5), ("tiny", 6, 3, 3), ("laby", 15, 20, 0)])
def test_library_backward_equals_the_spelled_out_schedule(case, fe, depth, seed):
    """tgnn_backward (csrc/train.hip) enqueues the kernels backward_train calls one by one: same inputs, same kernels, same
    order -- the same bits in every gradient."""
    from tilingnn_amd import train
    g = {"small": _small, "tiny": _tiny, "laby": load_labyrinth_graph}[case]()
    net, _ = _net(fe, depth, seed)
    x, adj, attr, col, _ = graph_tensors(g, torch.float32, DEV)
    probs, sv = train.forward_train(net, x, adj, attr, col)
    dprobs = _rng(1)(int(x.shape[0]), 1).to(DEV) * 1e-2
    a = train.backward_library(net, sv, dprobs)
    b = train.backward_train(net, sv, dprobs)
    assert sorted(a) == sorted(b) == sorted(k for k, _ in net.named_parameters())
    for k in a:
        assert torch.equal(a[k].reshape(-1), b[k].reshape(-1)), k
