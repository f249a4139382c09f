"""Cached-layout forward: eager launches vs hipGraph replay (captured through torch.cuda.graph).  GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict

def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0)); net = net.cuda().train()
for n in (300, 1254, 5000, 20000, 100000):
    sg = make_super_graph(n, 10 * n, int(12.5 * n), tile_count=2, n_edge_types=13, seed=2)
    x, adj, attr, col, _ = sg.to_torch("cuda:0")
    with torch.no_grad():
        ref = net(x, adj, attr, col)[0].clone()
        eager = bench(lambda: net(x, adj, attr, col))
        try:
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                net(x, adj, attr, col)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                out = net(x, adj, attr, col)[0]
            g.replay(); torch.cuda.synchronize()
            same = torch.equal(out, ref)
            replay = bench(g.replay)
            print(f"N={n}: eager {eager:.3f} ms, graph replay {replay:.3f} ms, identical={same}")
        except Exception as e:
            print(f"N={n}: eager {eager:.3f} ms, capture failed: {type(e).__name__}: {str(e)[:300]}")
