"""Same-box A/B of tgnn_set_dense_rows_mode (0 = round-5 head, 1 = no memsets / early edge-weight event, 3 = + fused init MLP):
probabilities against each other, cached-layout forward and full step (graph preparation inside), HIP events, alternating."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN
from tilingnn_amd._lib import lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
modes = [int(m) for m in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 2]
sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=1)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0), strict=True)
net = net.to(dev).train()
probs = {}
for m in modes:
    lib.tgnn_set_dense_rows_mode(m)
    probs[m] = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].clone()
for m in modes[1:]:
    print(f"mode {m} vs {modes[0]}: max |dp| = {float((probs[m] - probs[modes[0]]).abs().max()):.3e}, identical {bool((probs[m] == probs[modes[0]]).all())}")
def timed(cache):
    net.cache_graph = cache
    for _ in range(3): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
import statistics
acc = {m: ([], []) for m in modes}
for rep in range(int(os.environ.get("AB_REPS", "10"))):
    for m in modes:
        lib.tgnn_set_dense_rows_mode(m)
        acc[m][0].append(timed(True)); acc[m][1].append(timed(False))
for m in modes:
    c, p = acc[m]
    print(f"mode {m}: cached {statistics.mean(c):.4f} +- {statistics.stdev(c) / len(c) ** 0.5:.4f} ms (min {min(c):.4f})   with preparation "
          f"{statistics.mean(p):.4f} +- {statistics.stdev(p) / len(p) ** 0.5:.4f} ms (min {min(p):.4f})")
lib.tgnn_set_dense_rows_mode(2)
