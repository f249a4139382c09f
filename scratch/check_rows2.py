"""dense_f16_rows2_kernel (tgnn_set_dense_rows_mode(1)) against dense_f16_rows_kernel (mode 0): bit-identity of the slot-major
672 -> 256 Linear and of the whole forward's probabilities (its layers 256 -> 128 -> 64 take the BatchNorm-on-load variants), and
the cached-layout forward's time in both modes (HIP events, alternating)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import ops, TilinGNN
from tilingnn_amd._lib import lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
torch.manual_seed(0)
for N in (49152 + 37, 100_000, 300_001):
    mid = torch.randn(21, N, 32, device=dev); w = torch.randn(256, 672, device=dev) * 0.05; b = torch.randn(256, device=dev)
    parts = ops.new_partials(256, dev)
    outs = []
    for mode in (0, 1, 1):
        lib.tgnn_set_dense_rows_mode(mode)
        o, npart = ops.dense_act(mid, w, b, 1, slot_major=True, f16_split=True, partials=parts)
        sums = parts[: npart * 512].view(npart, 512).sum(0).clone()
        outs.append((o.clone(), sums, npart))
    ref = (mid.permute(1, 0, 2).reshape(N, 672)[:2000].double() @ w.double().t() + b.double())
    ref = torch.where(ref >= 0, ref, ref * 0.01)
    print(N, "bit-identical to mode 0:", bool((outs[0][0] == outs[1][0]).all()), "repeat:", bool((outs[1][0] == outs[2][0]).all()),
          "partials", outs[0][2], outs[1][2], "col sums rel diff", float(((outs[0][1] - outs[1][1]).abs() / outs[0][1].abs().clamp(min=1)).max()),
          "vs fp64 (2000 rows)", float((outs[1][0][:2000].double() - ref).abs().max() / ref.abs().max()))
n = 100_000
sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=1)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0), strict=True)
net = net.to(dev).train()
probs = {}
for mode in (0, 1):
    lib.tgnn_set_dense_rows_mode(mode)
    probs[mode] = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].clone()
print("forward probs bit-identical:", bool((probs[0] == probs[1]).all()), float((probs[0] - probs[1]).abs().max()))
for rep in range(3):
    for mode in (0, 1):
        lib.tgnn_set_dense_rows_mode(mode)
        for _ in range(3): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
        e1.record(); torch.cuda.synchronize()
        print("mode", mode, "cached forward ms", e0.elapsed_time(e1) / 20)
lib.tgnn_set_dense_rows_mode(1)
