"""Times the isolated NNConv op (column kernel) with HIP events: N=100k/Ea=1M/T=13 (or argv: n ea), warm L2.
TGNN_LIB_PATH selects the library build (ablations)."""
import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
ea = int(sys.argv[2]) if len(sys.argv) > 2 else 10 * n
dev = torch.device('cuda:0')
sg = make_super_graph(n, ea, ea // 4 * 5, tile_count=2, n_edge_types=13, seed=2)
x, adj, adj_attr, col, _ = sg.to_torch(dev)
g = ops.prepare_graph(n, adj, adj_attr, col)
torch.manual_seed(0)
h = torch.randn(n, 32, device=dev)
wtab = torch.randn(g.n_types, 32, 32, device=dev) * 0.2
root = torch.randn(32, 32, device=dev) * 0.2
bias = torch.randn(32, device=dev)
part = ops.new_partials(32, dev)
for _ in range(5):
    out, npart = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, part)
torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out, npart = ops.nnconv_mean(h, g, wtab, root, bias, ops.ACT_LEAKY_RELU, part)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
import os
print(f"{os.environ.get('TGNN_LIB_PATH', 'default'):40s} n={n} ea={ea} us/launch (incl. weight image, ~5 us): min {min(ts):.1f} median {sorted(ts)[2]:.1f}  checksum {float(out.double().sum()):.6e}")
