import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import load_labyrinth_graph, graph_tensors
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0)); net = net.cuda().train()
def bench(f, n=60):
    for _ in range(8): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
out = []
with torch.no_grad():
    x, adj, attr, col, _ = graph_tensors(load_labyrinth_graph(), torch.float32, "cuda:0")
    out.append("laby %.3f" % bench(lambda: net(x, adj, attr, col)))
    for n in (300, 2500, 5000, 10000, 20000, 50000, 100000):
        sg = make_super_graph(n, 10 * n, int(12.5 * n), tile_count=2, n_edge_types=13, seed=2)
        x, adj, attr, col, _ = sg.to_torch("cuda:0")
        out.append("%d %.3f" % (n, bench(lambda: net(x, adj, attr, col))))
print(os.environ.get("TGNN_COLS_TILES_PER_BLOCK"), os.environ.get("TGNN_COLS_MIN_NODES"), " | ".join(out))
