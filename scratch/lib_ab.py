"""argv: out file, sizes...: cached-layout forward times at the given sizes; probabilities and BatchNorm running buffers saved to the
out file.  Run once per library (TGNN_LIB_PATH=scratch/libs/...) and compare with scratch/lib_cmp.py: builds that must agree bit
for bit."""
import sys, time, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
out = {}
for n in [int(a) for a in sys.argv[2:]]:
    net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
    sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
    x, adj, attr, col, _ = sg.to_torch(dev)
    p = net(x, adj, attr, col)[0]
    p2 = net(x, adj, attr, col)[0]
    out[n] = (p.cpu(), p2.cpu(), {k: v.cpu() for k, v in net.state_dict().items() if 'running' in k or 'tracked' in k})
    for _ in range(5): net(x, adj, attr, col)
    torch.cuda.synchronize(); ts = []
    for _ in range(40):
        t = time.perf_counter(); net(x, adj, attr, col); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"n {n}: cached-layout forward median {sorted(ts)[20]:.3f} ms  min {min(ts):.3f}", flush=True)
torch.save(out, sys.argv[1])
