"""The final MLP as one persistent kernel behind the mid-size layer loop (csrc/forward_tail.hip) against the general final MLP:
probabilities, the final BatchNorms' running statistics, run-to-run bits, cached-layout forward time."""
import sys, time, torch
sys.path.insert(0, ".")
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [4100, 5000, 10000, 20000, 32768, 50000]
_lib.lib.tgnn_set_mid_layout_limit(65536)
for n in sizes:
    ea, ec = (8 * n, 10 * n) if n == 10000 else (10 * n, 12 * n + n // 2)
    sg = make_super_graph(n, ea, ec, tile_count=2, n_edge_types=13, seed=1)
    x, adj, attr, col, _ = sg.to_torch(dev)
    res = {}
    for tail in (0, 3):
        _lib.lib.tgnn_set_mid_tail(tail)
        net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
        net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0), strict=True)
        net = net.to(dev).train()
        p0 = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].clone()
        rm = [net.final_mlp[0].mlp[l].batch_norm.running_mean.clone() for l in range(4)]
        rv = [net.final_mlp[0].mlp[l].batch_norm.running_var.clone() for l in range(4)]
        same = all(torch.equal(net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0], p0) for _ in range(5))
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(20):
                net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 20 * 1e3)
        res[tail] = (p0, rm, rv, same, min(ts))
    d = float((res[0][0].double() - res[3][0].double()).abs().max())
    drm = max(float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)) for a, b in zip(res[0][1], res[3][1]))
    drv = max(float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)) for a, b in zip(res[0][2], res[3][2]))
    print(f"n {n:6d}: max |probs diff| {d:.2e}  running mean / var rel diff {drm:.1e} / {drv:.1e}  reproducible {res[0][3]} / {res[3][3]}  "
          f"cached forward general tail {res[0][4]:.3f} ms, persistent tail {res[3][4]:.3f} ms", flush=True)
