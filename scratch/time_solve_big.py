"""BASELINE configs 4 / 5 on ONE GPU: a whole greedy solve of a 500 000- and a 2 000 000-node layout with the device acceptance step
(the reference's host sweep would take ~1 000+ forwards of that size)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN
from tilingnn_amd.solver.ml_solver.ml_solver import ML_Solver
from tilingnn_amd.synth import make_super_graph_on_device
from tilingnn_amd.util import algorithms as alg
from tilingnn_amd.weights import make_state_dict
dev = torch.device("cuda:0")
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0), strict=True)
net = net.to(dev).train()
ms = ML_Solver(None, dev, None, net, num_prob_maps=1)
for n, ea, ec in ((500_000, 6_000_000, 7_500_000), (2_000_000, 20_000_000, 25_000_000)):
    x, adj, attr, col, _ = make_super_graph_on_device(n, ea, ec, dev, tile_count=2, n_edge_types=13, seed=4)
    layout = alg.DeviceLayout(x, adj, attr, col)
    sizes = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sel, _, order = alg.solve_by_device_greedy(ms, layout, seed=1, on_round=lambda l: sizes.append(int(l.node_feature.shape[0])))
    torch.cuda.synchronize(); t_dev = time.perf_counter() - t0
    sel_t = torch.as_tensor(sel, device=dev).bool()
    both = sel_t[col[0]] & sel_t[col[1]] & (col[0] != col[1])
    print(f"n {n} ({ea} + {ec} edges): solved in {t_dev * 1e3:.0f} ms, {alg.solve_by_device_greedy.last_rounds} rounds, {int(sel_t.sum())} tiles "
          f"selected, colliding pairs among them {int(both.sum())}; sub-layout sizes {sizes[:4]} ... {sizes[-3:]}", flush=True)
    del x, adj, attr, col, layout
    torch.cuda.empty_cache()
