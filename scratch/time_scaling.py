import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, ops
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev)
l1 = net.brch_1_graph_conv_layers[0]; l2 = net.brch_2_coll_conv_layers[0]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for N in (1024, 16384, 32768, 65536, 100000, 200000, 400000):
    sg = make_super_graph(N, 10*N, int(12.5*N), tile_count=2, n_edge_types=13, seed=2)
    x, adj, adj_attr, col, _ = sg.to_torch(dev)
    h = torch.randn(N, 32, device=dev)
    g = ops.prepare_graph(N, adj, adj_attr, col)
    wtab = ops.edge_weight_table(adj_attr, g, *l1.nnConv._edge_mlp_params(), 32)
    parts = ops.new_partials(32, dev)
    t1 = timeit(lambda: ops.nnconv_mean(h, g, wtab, l1.nnConv.root, l1.nnConv.bias, act=1, partials=parts))
    t2 = timeit(lambda: ops.gin(h, g, l2.ginConv.eps, *l2.ginConv._mlp_params(), act=1, partials=parts))
    print(f"N={N:7d} nnconv {t1:7.1f} us  gin {t2:7.1f} us", flush=True)
