import sys, time, os, torch
sys.path.insert(0, '.')
from oracle import tilingnn_oracle as orc
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
sg = make_super_graph(5000, 50000, 62500, tile_count=2, n_edge_types=13, seed=11)
sd = make_state_dict(15, 20, 32, 1, 3, seed=0)
x, adj, adj_attr, col, _ = sg.to_torch("cpu")
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        orc.tilingnn_forward(sd, x, adj, adj_attr, col)
        t0 = time.perf_counter(); orc.tilingnn_forward(sd, x, adj, adj_attr, col); dt = time.perf_counter() - t0
    print(th, f"{5000/dt:.0f} nodes/s", flush=True)
