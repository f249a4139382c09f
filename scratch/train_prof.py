"""Five training steps at the benchmark shape, for rocprofv3 --kernel-trace --stats (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
from tilingnn_amd.solver.ml_solver.losses import Losses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
sg = make_super_graph(n, 10 * n, int(12.5 * n), tile_count=2, n_edge_types=13, seed=2)
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0)); net = net.cuda().train(); net.autograd = True
x, adj, attr, col, _ = sg.to_torch("cuda:0")
opt = torch.optim.Adam(net.parameters(), lr=1e-4, **({"fused": True} if os.environ.get("ADAM_FUSED") == "1" else {}))
def step():
    probs, _ = net(x, adj, attr, col)
    opt.zero_grad()
    loss, _, _ = Losses.calculate_unsupervised_loss(probs, x, col, adj, attr)
    loss.backward()
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t) / 5 * 1e3)
# host-side cost of issuing the backward alone (no sync inside): time until the calls return
probs, _ = net(x, adj, attr, col); loss, _, _ = Losses.calculate_unsupervised_loss(probs, x, col, adj, attr)
torch.cuda.synchronize(); t = time.perf_counter(); loss.backward(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("backward: host issue %.1f ms, until done %.1f ms" % ((t1 - t) * 1e3, (t2 - t) * 1e3))
# steady state: host time to ISSUE five steps (no sync inside) vs wall time until the GPU is done
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): step()
t_issue = time.perf_counter() - t
torch.cuda.synchronize(); t_done = time.perf_counter() - t
print("5 steps: host issue %.1f ms, until done %.1f ms" % (t_issue * 1e3, t_done * 1e3))
