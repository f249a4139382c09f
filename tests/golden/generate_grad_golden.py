"""Golden gradients for the training step, produced by the REFERENCE itself.

Run in the build container only (needs /root/reference):   python tests/golden/generate_grad_golden.py

One training step of `Trainer.train` (solver/ml_solver/trainer.py:68-84) without the optimizer: the reference's network
(graph_networks/*, imported unchanged, PyG ops stood in as in generate_golden.py) in train mode, its own
`Losses.calculate_unsupervised_loss` (solver/ml_solver/losses.py:48-116, imported unchanged), `loss.backward()`.
Everything in float64, single thread.

Stored (ref_grads.npz)
  small.*   256-node induced sub-graph of the labyrinth graph, depth 3 (the residual skip is live at layer 2), width 32:
            probs, loss, d loss / d probs, and the gradient of EVERY parameter (rounded once to float32).
  laby.*    the full labyrinth graph, depth 20: probs, loss, d loss / d probs; for every parameter the triple
            (sum, L2 norm, dot with a seeded N(0,1) vector) of its gradient, and six gradients in full.
  <case>.err32.* / err32stat.* / loss32
            the reference's OWN float32 step against its float64 one, per parameter: the yardstick for a float32
            implementation (gradients through 20 train-mode BatchNorms amplify rounding by orders of magnitude).
  tiny.*    the 6-node graph (self loops, zero-in-degree node), depth 3: the triples, and every gradient in full except
            the two largest tensors.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import generate_golden as gg                                   # noqa: E402

FULL_KEYS = ("brch_1_graph_conv_layers.0.nnConv.root", "brch_1_graph_conv_layers.19.mlp.mlp.2.linear.weight",
             "brch_2_coll_conv_layers.7.ginConv.nn.mlp.1.linear.weight", "init_node_feature_trans.mlp.0.linear.weight",
             "final_mlp.1.linear.weight", "brch_1_graph_conv_layers.10.batch_norm.weight")


def projection(name, shape):
    seed = int.from_bytes(name.encode(), "little") % (2 ** 31)
    return np.random.default_rng(seed).standard_normal(shape)


def step(net, Losses, g, dtype=torch.float64):
    x, adj, adj_attr, col, col_attr = gg.to_t(g, dtype)
    net.zero_grad()
    probs, _ = net(x=x, adj_e_index=adj, adj_e_features=adj_attr, col_e_idx=col, col_e_features=None)
    probs.retain_grad()
    loss, _, _ = Losses.calculate_unsupervised_loss(probs, x, col, adj_edges_index=adj, adj_edge_features=adj_attr)
    loss.backward()
    grads = {k: p.grad.detach().numpy().copy() for k, p in net.named_parameters()}
    return probs.detach().numpy().copy(), float(loss.item()), probs.grad.detach().numpy().copy(), grads


def stats(k, v):
    v = v.astype(np.float64)
    return np.array([v.sum(), np.sqrt((v ** 2).sum()), (v * projection(k, v.shape)).sum()])


def fp32_spread(out, case, TilinGNN, Losses, g, fe, fx, seed, depth, grads64):
    """The reference's OWN float32 training step against its float64 one: what single precision costs on this problem.
    err32.<name> = max |g32 - g64| / max |g64|;  err32stat.<name> = |stats(g32) - stats(g64)|."""
    net32 = gg.build_reference_net(TilinGNN, fe, fx, torch.float32, seed=seed, depth=depth, width=32)
    _, loss32, dprobs32, grads32 = step(net32, Losses, g, torch.float32)
    out[f"{case}.loss32"] = np.float64(loss32)
    for k, v64 in grads64.items():
        v32 = grads32[k].astype(np.float64)
        out[f"{case}.err32.{k}"] = np.float64(np.abs(v32 - v64).max() / max(np.abs(v64).max(), 1e-300))
        out[f"{case}.err32stat.{k}"] = np.abs(stats(k, v32) - stats(k, v64))


def main():
    torch.set_num_threads(1)
    g = gg.load_labyrinth()
    fe, fx = g["adj_attr"].shape[1], g["x"].shape[1]
    TilinGNN = gg.import_reference(g["tile_count"])
    cfg = sys.modules["inputs.config"]
    cfg.COLLISION_WEIGHT, cfg.ALIGN_LENGTH_WEIGHT, cfg.AVG_AREA_WEIGHT = 1 / math.log(1 + 1e-1), 0.02, 1   # config.py:49-51
    from solver.ml_solver.losses import Losses

    out = {}
    gs = gg.induced_subgraph(g, 256)
    net = gg.build_reference_net(TilinGNN, fe, fx, torch.float64, seed=5, depth=3, width=32)
    probs, loss, dprobs, grads = step(net, Losses, gs)
    out["small.probs"], out["small.loss"], out["small.dprobs"] = probs, np.float64(loss), dprobs
    for k, v in grads.items():
        out[f"small.grad.{k}"] = v.astype(np.float32)
    fp32_spread(out, "small", TilinGNN, Losses, gs, fe, fx, 5, 3, grads)
    print("small: loss", loss, "params", len(grads), "max |grad|", max(np.abs(v).max() for v in grads.values()))

    net = gg.build_reference_net(TilinGNN, fe, fx, torch.float64, seed=0, depth=20, width=32)
    probs, loss, dprobs, grads = step(net, Losses, g)
    out["laby.probs"], out["laby.loss"], out["laby.dprobs"] = probs, np.float64(loss), dprobs
    for k, v in grads.items():
        out[f"laby.stat.{k}"] = np.array([v.sum(), np.sqrt((v ** 2).sum()), (v * projection(k, v.shape)).sum()])
    for k in FULL_KEYS:
        out[f"laby.grad.{k}"] = grads[k]
    fp32_spread(out, "laby", TilinGNN, Losses, g, fe, fx, 0, 20, grads)
    print("laby: loss", loss, "params", len(grads), "max |grad|", max(np.abs(v).max() for v in grads.values()))
    e = np.array([out[f"laby.err32.{k}"] for k in grads])
    print("laby: reference fp32 vs fp64 gradient error: median %.2e  max %.2e" % (np.median(e), e.max()))

    tiny = dict(np.load(os.path.join(HERE, "tiny_graph.npz")))
    TilinGNN2 = gg.import_reference(2)
    sys.modules["inputs.config"].COLLISION_WEIGHT = cfg.COLLISION_WEIGHT
    sys.modules["inputs.config"].ALIGN_LENGTH_WEIGHT, sys.modules["inputs.config"].AVG_AREA_WEIGHT = 0.02, 1
    nett = gg.build_reference_net(TilinGNN2, tiny["adj_attr"].shape[1], 3, torch.float64, seed=3, depth=3, width=32)
    tg = dict(x=tiny["x"], adj=tiny["adj"], adj_attr=tiny["adj_attr"], col=tiny["col"], col_attr=tiny["col_attr"])
    probs, loss, dprobs, grads = step(nett, Losses, tg)
    out["tiny.probs"], out["tiny.loss"], out["tiny.dprobs"] = probs, np.float64(loss), dprobs
    for k, v in grads.items():
        out[f"tiny.stat.{k}"] = np.array([v.sum(), np.sqrt((v ** 2).sum()), (v * projection(k, v.shape)).sum()])
        if "mlp.mlp.2" not in k and "final_mlp.0.mlp.0.linear.weight" not in k:      # the two big ones: stats only
            out[f"tiny.grad.{k}"] = v.astype(np.float32)
    fp32_spread(out, "tiny", TilinGNN2, Losses, tg, tiny["adj_attr"].shape[1], 3, 3, 3, grads)
    print("tiny: loss", loss)
    path = os.path.join(HERE, "ref_grads.npz")
    np.savez_compressed(path, **out)
    print(f"ref_grads.npz: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
