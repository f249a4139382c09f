"""End-to-end gradient check of the HIP training step against tests/golden/ref_grads.npz (GPU box)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import load_npz, load_labyrinth_graph, graph_tensors
from tilingnn_amd.weights import make_state_dict
from tilingnn_amd.graph_networks.networks.TilinGNN import TilinGNN
from tilingnn_amd.solver.ml_solver.losses import Losses

ref = load_npz("ref_grads.npz")
def run(case, g, fe, depth, seed):
    net = TilinGNN(adj_edge_features_dim=fe, network_depth=depth, network_width=32, node_features_dim=3)
    net.load_state_dict(make_state_dict(fe, depth, 32, 1, 3, seed=seed)); net = net.cuda().train(); net.autograd = True
    x, adj, attr, col, _ = graph_tensors(g, torch.float32, "cuda")
    probs, _ = net(x, adj, attr, col)
    probs.retain_grad()
    loss, mi, _ = Losses.calculate_unsupervised_loss(probs, x, col, adj, attr)
    loss.backward()
    torch.cuda.synchronize()
    print(case, "loss", float(loss), "ref", float(ref[case + ".loss"]))
    print("  probs err", np.abs(probs.detach().cpu().numpy() - ref[case + ".probs"]).max())
    dp = probs.grad.cpu().numpy(); rdp = ref[case + ".dprobs"]
    print("  dprobs rel err", np.abs(dp - rdp).max() / np.abs(rdp).max())
    worst = []
    for name, p in net.named_parameters():
        key = f"{case}.grad.{name}"
        if key in ref.files:
            want = ref[key].astype(np.float64); got = p.grad.cpu().numpy().astype(np.float64)
            worst.append((np.abs(got - want).max() / max(np.abs(want).max(), 1e-30), name, np.abs(want).max()))
        key = f"{case}.stat.{name}"
        if key in ref.files and f"{case}.grad.{name}" not in ref.files:
            got = p.grad.cpu().numpy().astype(np.float64); w = ref[key]
            worst.append((abs(np.sqrt((got ** 2).sum()) - w[1]) / max(w[1], 1e-30), name + " (norm)", w[1]))
    worst.sort(reverse=True)
    for w in worst[:12]: print("   %.3e  %s  (max |ref| %.3e)" % (w[0], w[1], w[2]))
    print("   median rel err %.3e over %d tensors" % (np.median([w[0] for w in worst]), len(worst)))

z = load_npz("ref_ops_small.npz")
small = dict(x=z["x"], adj=z["adj"].astype(np.int64), adj_attr=z["adj_attr"], col=z["col"].astype(np.int64), col_attr=z["col_attr"])
run("small", small, 15, 3, 5)
t = load_npz("tiny_graph.npz")
run("tiny", dict(x=t["x"], adj=t["adj"], adj_attr=t["adj_attr"], col=t["col"], col_attr=t["col_attr"]), 6, 3, 3)
run("laby", load_labyrinth_graph(), 15, 20, 0)
