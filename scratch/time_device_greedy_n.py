"""solve_by_device_greedy at the benchmark layout (100 000 nodes): wall time per round of {sub-layout build, predict, acceptance},
measured with a synchronise around each piece (the pieces' own costs, not their overlap), then the un-instrumented solve."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
from tilingnn_amd.solver.ml_solver.ml_solver import ML_Solver
from tilingnn_amd.util import algorithms as alg
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
sg = make_super_graph(n, int(6.8 * n), int(8.35 * n), tile_count=2, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
net.cache_graph = False
solver = ML_Solver(None, dev, None, net, num_prob_maps=1)
lay = alg.DeviceLayout(x, adj, attr, col)
for _ in range(2): alg.solve_by_device_greedy(solver, lay, seed=1)
torch.cuda.synchronize(); t = time.perf_counter()
alg.solve_by_device_greedy(solver, lay, seed=1); torch.cuda.synchronize()
print("solve %.2f ms, %d rounds" % ((time.perf_counter() - t) * 1e3, alg.solve_by_device_greedy.last_rounds))
rows = []
ob, op = alg.SubLayoutBuilder.build, solver.predict_on_device
def build(self, alive):
    torch.cuda.synchronize(); t = time.perf_counter(); r = ob(self, alive); torch.cuda.synchronize()
    rows.append([int(r.node_feature.shape[0]), (time.perf_counter() - t) * 1e3, 0.0]); return r
def pred(sub):
    torch.cuda.synchronize(); t = time.perf_counter(); r = op(sub); torch.cuda.synchronize(); rows[-1][2] = (time.perf_counter() - t) * 1e3; return r
alg.SubLayoutBuilder.build = build; solver.predict_on_device = pred
torch.cuda.synchronize(); t = time.perf_counter()
alg.solve_by_device_greedy(solver, lay, seed=1); torch.cuda.synchronize()
tot = (time.perf_counter() - t) * 1e3
for r in rows: print("n %6d  build %.3f ms  predict %.3f ms" % tuple(r))
print("instrumented solve %.2f ms: build %.2f, predict %.2f, rest %.2f" % (tot, sum(r[1] for r in rows), sum(r[2] for r in rows), tot - sum(r[1] + r[2] for r in rows)))
