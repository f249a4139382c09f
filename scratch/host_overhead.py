"""Host time of one TilinGNN.forward call (no synchronise inside the timed region) at 10 000 nodes, cached layout, and where it goes."""
import os, sys, time, torch, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
n = 10_000
sg = make_super_graph(n, 8 * n, 10 * n, tile_count=2, n_edge_types=13, seed=1)
x, adj, attr, col, _ = sg.to_torch(dev)
for cached in (True, False):
    net.cache_graph = cached
    for _ in range(10): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
    torch.cuda.synchronize()
    ts = []
    for _ in range(50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
        ts.append((time.perf_counter() - t0) * 1e6)
    print(f"cached={cached}: host time of the call {sorted(ts)[25]:.0f} us (median of 50)")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
    pr.disable(); torch.cuda.synchronize()
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(12); print("\n".join(st.getvalue().splitlines()[:24]))

# ---- where the host time goes: python before the library call | the call | python after
import tilingnn_amd.graph_networks.networks.TilinGNN as mod
real = mod.lib.tgnn_forward
marks = []
class Wrap:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = real(*a); marks.append((t0, time.perf_counter())); return r
mod.lib.tgnn_forward = Wrap()
net.cache_graph = True
for _ in range(5): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
torch.cuda.synchronize()
pre, call, post, e2e = [], [], [], []
for _ in range(50):
    torch.cuda.synchronize(); marks.clear(); t0 = time.perf_counter()
    net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    a, b = marks[-1]
    pre.append((a - t0) * 1e6); call.append((b - a) * 1e6); post.append((t1 - b) * 1e6); e2e.append((t2 - t0) * 1e6)
med = lambda v: sorted(v)[len(v) // 2]
print(f"cached forward at {n} nodes: python before the library call {med(pre):.0f} us | the call {med(call):.0f} us | python after {med(post):.0f} us | "
      f"call entry -> results on the device {med(e2e):.0f} us")
