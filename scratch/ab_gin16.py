"""A/B inside one process: cached-layout forward at n nodes with the inference GIN MLP on fp16 pairs (gin32_mlp16_kernel) vs the
bf16 x 3 kernel.  argv: sizes"""
import sys, time, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
for n in [int(a) for a in sys.argv[1:]] or [100_000]:
    sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
    x, adj, attr, col, _ = sg.to_torch(dev)
    net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
    res = {}
    for rnd in range(3):
        for mode in (0, 1):
            _lib.lib.tgnn_set_gin_mlp_f16(mode)
            for _ in range(5): net(x, adj, attr, col)
            torch.cuda.synchronize(); ts = []
            for _ in range(30):
                t = time.perf_counter(); net(x, adj, attr, col); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
            res.setdefault(mode, []).append(sorted(ts)[15])
    _lib.lib.tgnn_set_gin_mlp_f16(1)
    print(f"n {n}: cached-layout forward median ms  bf16x3 MLP {['%.3f' % v for v in res[0]]}   fp16-pair MLP {['%.3f' % v for v in res[1]]}", flush=True)
