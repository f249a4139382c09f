import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import load_labyrinth_graph, graph_tensors
from tilingnn_amd import TilinGNN
from tilingnn_amd.weights import make_state_dict
g = load_labyrinth_graph()
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0)); net = net.cuda().train()
x, adj, attr, col, _ = graph_tensors(g, torch.float32, "cuda:0")
with torch.no_grad():
    for _ in range(10): net(x, adj, attr, col)
    torch.cuda.synchronize()
    time.sleep(0.05)
    for _ in range(3):
        net(x, adj, attr, col); torch.cuda.synchronize(); time.sleep(0.02)
