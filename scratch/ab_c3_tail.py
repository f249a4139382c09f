"""Config 3 (width 64, bf16 storage, 100 000 nodes): the final MLP's Linears 1 .. 3 on the fp16-pair resident kernels (split precision 1)
against the bf16 x 3 block-tile kernels (0), same box, alternating; cached layout, HIP events over 20 forwards."""
import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN
from tilingnn_amd._lib import lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
n = 100_000
sg = make_super_graph(n, 10 * n, 10 * n // 4 * 5, tile_count=4, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=64, node_features_dim=5)
net.load_state_dict(make_state_dict(15, 20, 64, 1, 5, seed=0))
net = net.to(dev).train()
net.activation_dtype = torch.bfloat16
out = {}
for mode in (0, 1):
    lib.tgnn_set_split_precision(mode)
    out[mode] = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].clone()
print("max |dp| between the two tails:", float((out[0] - out[1]).abs().max()))
for rep in range(4):
    for mode in (0, 1):
        lib.tgnn_set_split_precision(mode)
        for _ in range(3): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
        e1.record(); torch.cuda.synchronize()
        print(f"mode {mode}: {e0.elapsed_time(e1) / 20:.4f} ms", flush=True)
