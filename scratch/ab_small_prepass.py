"""Small layouts (the one-launch preparation + the persistent forward): result words polled in pinned memory (1) against the copy +
stream synchronise (0); forward with preparation, HIP events over 40 forwards, alternating."""
import sys, torch, statistics
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN
from tilingnn_amd._lib import lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
for n, ea, ec in ((1254, 8502, 10472), (4000, 32000, 40000)):
    sg = make_super_graph(n, ea, ec, tile_count=2, n_edge_types=13, seed=1)
    x, adj, attr, col, _ = sg.to_torch(dev)
    net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
    net.cache_graph = False
    out = {}
    for m in (0, 1):
        lib.tgnn_set_lean_head(11 if m == 0 else 3)
        out[m] = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].clone()
    acc = {0: [], 1: []}
    for rep in range(8):
        for m in (0, 1):
            lib.tgnn_set_lean_head(11 if m == 0 else 3)
            for _ in range(5): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
            e1.record(); torch.cuda.synchronize()
            acc[m].append(e0.elapsed_time(e1) / 40)
    print(f"n {n}: identical {bool((out[0] == out[1]).all())}; " + "; ".join(f"prepass {m}: {statistics.mean(acc[m]):.4f} +- {statistics.stdev(acc[m]) / 8 ** 0.5:.4f} ms" for m in (0, 1)), flush=True)
lib.tgnn_set_lean_head(3)
