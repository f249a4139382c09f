"""The gradient yardstick of tests/test_training_hip.py::test_training_step_with_many_edge_types over more weight seeds: per seed
our worst parameter error against the float64 oracle, the float32 oracle's own, and the worst per-parameter ratio."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import tilingnn_oracle as orc
from tests.test_training_hip import _rel, DEV
from tilingnn_amd.graph_networks.networks.TilinGNN import TilinGNN
from tilingnn_amd.solver.ml_solver.losses import Losses
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
sg = make_super_graph(600, 6000, 7500, tile_count=2, n_edge_types=25, seed=9)
fe = 2 + 25
x, adj, attr, col, _ = sg.to_torch(DEV)
torch.set_num_threads(8)
rows = []
for seed in [int(a) for a in sys.argv[1:]] or list(range(1, 11)):
    net = TilinGNN(adj_edge_features_dim=fe, network_depth=2, network_width=32, node_features_dim=3)
    sd = make_state_dict(fe, 2, 32, 1, 3, seed=seed)
    net.load_state_dict(sd); net = net.to(DEV).train(); net.autograd = True
    probs, _ = net(x, adj, attr, col)
    loss, _, _ = Losses.calculate_unsupervised_loss(probs, x, col, adj, attr)
    loss.backward()
    _, ref_loss, _, ref = orc.training_step_grads(orc.cast_sd(sd, torch.float64), x.double().cpu(), adj.cpu(), attr.double().cpu(), col.cpu())
    _, _, _, f32 = orc.training_step_grads(orc.cast_sd(sd, torch.float32), x.cpu(), adj.cpu(), attr.cpu(), col.cpu())
    err32 = {k: _rel(f32[k], ref[k]) for k in ref}
    floor = float(np.median(list(err32.values())))
    errs = {k: _rel(p.grad, ref[k]) for k, p in net.named_parameters()}
    worst = max(e / (max(err32[k], floor) + 2.5e-6) for k, e in errs.items())
    print(f"seed {seed}: ours worst {max(errs.values()):.2e} median {np.median(list(errs.values())):.2e} | f32 oracle worst {max(err32.values()):.2e} median {floor:.2e} | worst per-parameter ratio {worst:.1f}", flush=True)
