"""A whole greedy solve at n nodes: the host sweep (reference semantics, numpy RNG) against the batched device acceptance."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN
from tilingnn_amd.solver.ml_solver.ml_solver import ML_Solver
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.util import algorithms as alg
from tilingnn_amd.weights import make_state_dict
dev = torch.device("cuda:0")
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0), strict=True)
net = net.to(dev).train()
ms = ML_Solver(None, dev, None, net, num_prob_maps=1)
for n in [int(a) for a in sys.argv[1:]] or [10000, 100000]:
    sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
    x, adj, attr, col, _ = sg.to_torch(dev)
    layout = alg.DeviceLayout(x, adj, attr, col)
    alg.solve_by_device_greedy(ms, layout, seed=1)                     # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sel, _, order = alg.solve_by_device_greedy(ms, layout, seed=1)
    torch.cuda.synchronize(); t_dev = time.perf_counter() - t0
    r_dev = alg.solve_by_device_greedy.last_rounds
    sizes = []
    np.random.seed(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sel_h, _, order_h = alg.solve_by_probablistic_greedy(ms, layout, on_round=lambda l: sizes.append(int(l.node_feature.shape[0])))
    torch.cuda.synchronize(); t_host = time.perf_counter() - t0
    print(f"n {n}: device-batched acceptance {t_dev * 1e3:.1f} ms in {r_dev} rounds ({int(sel.sum())} tiles) | host sweep (reference "
          f"semantics) {t_host * 1e3:.1f} ms in {len(sizes)} rounds ({int(sel_h.sum())} tiles); sub-layout sizes of the host loop: "
          f"first {sizes[:3]}, median {int(np.median(sizes))}, last {sizes[-3:]}", flush=True)
