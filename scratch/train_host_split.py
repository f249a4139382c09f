"""Host-side cost of the backward at a small layout: cProfile of train.backward_train called directly."""
import cProfile, pstats, io, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN, train, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
n = int(sys.argv[1]) if len(sys.argv) > 1 else 999
sg = make_super_graph(n, 10 * n, int(12.5 * n), tile_count=2, n_edge_types=13, seed=2)
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0)); net = net.cuda().train()
x, adj, attr, col, _ = sg.to_torch("cuda:0")
dp = torch.randn(n, 1, device="cuda") * 1e-3
for _ in range(3):
    probs, sv = train.forward_train(net, x, adj, attr, col); train.backward_train(net, sv, dp)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    probs, sv = train.forward_train(net, x, adj, attr, col)
torch.cuda.synchronize(); tf = (time.perf_counter() - t) / 10
t = time.perf_counter()
for _ in range(10):
    with _lib.pinned_stream("cuda:0"):
        train.backward_train(net, sv, dp)
t_issue = (time.perf_counter() - t) / 10
torch.cuda.synchronize(); tb = (time.perf_counter() - t) / 10
print(f"N={n}: forward_train {tf*1e3:.2f} ms; backward_train host issue {t_issue*1e3:.2f} ms, done {tb*1e3:.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    with _lib.pinned_stream("cuda:0"):
        train.backward_train(net, sv, dp)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
