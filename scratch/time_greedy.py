"""Where a greedy solve spends its time (labyrinth graph, 1254 nodes).  GPU box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import load_labyrinth_graph
from tilingnn_amd import TilinGNN
from tilingnn_amd.weights import make_state_dict
from tilingnn_amd.solver.ml_solver.ml_solver import ML_Solver, LayoutArrays
from tilingnn_amd.util import algorithms as alg

g = load_labyrinth_graph()
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0)); net = net.cuda().train()
layout = LayoutArrays(g["x"].astype(np.float32), g["adj"], g["adj_attr"].astype(np.float32), g["col"], g["col_attr"].astype(np.float32))
solver = ML_Solver(None, "cuda:0", None, net, num_prob_maps=1)
T = {"build": 0.0, "predict": 0.0, "rounds": 0}
orig_build, orig_predict = alg.SubLayoutBuilder.build, solver.predict
def build(self, alive):
    torch.cuda.synchronize(); t = time.perf_counter(); r = orig_build(self, alive); torch.cuda.synchronize(); T["build"] += time.perf_counter() - t; return r
def predict(lay):
    torch.cuda.synchronize(); t = time.perf_counter(); r = orig_predict(lay); torch.cuda.synchronize(); T["predict"] += time.perf_counter() - t; T["rounds"] += 1; return r
alg.SubLayoutBuilder.build = build; solver.predict = predict
for rep in range(3):
    T.update(build=0.0, predict=0.0, rounds=0)
    np.random.seed(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sel, _, order = alg.solve_by_probablistic_greedy(solver, layout)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print(f"solve {tot*1e3:.1f} ms, rounds {T['rounds']}, selected {int(sel.sum())}: sub-layout build {T['build']*1e3:.1f} ms, predict {T['predict']*1e3:.1f} ms, "
          f"rest (host sweep, copies) {(tot-T['build']-T['predict'])*1e3:.1f} ms")
# split of one predict at full size: prep vs forward
from tilingnn_amd import ops
x, adj, attr, col, _ = layout.get_data_as_torch_tensor("cuda:0")
def tm(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("prepare_graph %.3f ms" % tm(lambda: ops.prepare_graph(1254, adj, attr, col)))
with torch.no_grad():
    net.cache_graph = True
    print("forward (cached layout) %.3f ms" % tm(lambda: net(x, adj, attr, col)))
    net.cache_graph = False
    print("forward (prep every call) %.3f ms" % tm(lambda: net(x, adj, attr, col)))
    print("predict (incl. .cpu()) %.3f ms" % tm(lambda: orig_predict(layout)))
