"""Where prepare_graph's wall time goes at the benchmark shape: Python before the library call, the call itself (host side of
~20 launches), the wait for the result words, Python after."""
import sys, time, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops, _lib
from tilingnn_amd.synth import make_super_graph
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
real = _lib.lib.tgnn_graph_prep
marks = {}
def timed(*a):
    marks['call0'] = time.perf_counter(); r = real(*a); marks['call1'] = time.perf_counter(); return r
ops.lib.tgnn_graph_prep = timed
orig_cpu = torch.Tensor.cpu
def cpu(self, *a, **k):
    marks.setdefault('cpu0', time.perf_counter()); r = orig_cpu(self, *a, **k); marks.setdefault('cpu1', time.perf_counter()); return r
for _ in range(5): ops.prepare_graph(n, adj, attr, col)
torch.cuda.synchronize()
acc = [0.0] * 5
N = 30
for _ in range(N):
    marks.clear(); torch.Tensor.cpu = cpu
    t0 = time.perf_counter(); ops.prepare_graph(n, adj, attr, col); t1 = time.perf_counter()
    torch.Tensor.cpu = orig_cpu; torch.cuda.synchronize()
    for k, v in enumerate((marks['call0'] - t0, marks['call1'] - marks['call0'], marks['cpu0'] - marks['call1'], marks['cpu1'] - marks['cpu0'], t1 - marks['cpu1'])):
        acc[k] += v * 1e6 / N
print(f"n {n}: python before {acc[0]:.0f} us | library call {acc[1]:.0f} us | between {acc[2]:.0f} us | .cpu() wait {acc[3]:.0f} us | python after {acc[4]:.0f} us | total {sum(acc):.0f} us")
