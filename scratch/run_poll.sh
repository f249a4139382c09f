#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/ab_poll.py <<PY
import os, sys, torch, statistics
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from tilingnn_amd import TilinGNN
from tilingnn_amd._lib import lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
n = 100_000
sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=1)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
net.cache_graph = False
out = {}
for m in (0, 1):
    lib.tgnn_set_prep_words_poll(m)
    out[m] = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].clone()
print("identical", bool((out[0] == out[1]).all()))
acc = {0: [], 1: []}
for rep in range(10):
    for m in (0, 1):
        lib.tgnn_set_prep_words_poll(m)
        for _ in range(3): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
        e1.record(); torch.cuda.synchronize()
        acc[m].append(e0.elapsed_time(e1) / 20)
for m in (0, 1):
    print(f"poll {m}: with preparation {statistics.mean(acc[m]):.4f} +- {statistics.stdev(acc[m]) / 10 ** 0.5:.4f} ms (min {min(acc[m]):.4f})")
PY
timeout 300 python /tmp/ab_poll.py 2>&1 | tail -3
timeout 300 python -m pytest tests/test_forward_begin.py tests/test_running_stats.py tests/test_hip_parity.py -m gpu -q -x -k "begin or running or prep or csr or index or forward" 2>&1 | tail -3
