"""Runs only NNConv / GIN launches (for PMC passes): N=100k/Ea=1M synthetic graph.  argv[1]: nnconv = the edge-group kernel as
tgnn_forward runs it at this size, nnconv_cols = the type-column kernel (fp16 x 2 split), nnconv_bf16x3 = its bf16 x 3 variant (the per-op entry point), gin = the collision branch."""
import sys, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, ops
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
sg = make_super_graph(100_000, 1_000_000, 1_250_000, tile_count=2, n_edge_types=13, seed=2)
x, adj, adj_attr, col, _ = sg.to_torch(dev)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev)
h = torch.randn(100_000, 32, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else 'nnconv'
g = ops.prepare_graph(100_000, adj, adj_attr, col)
conv = net.brch_1_graph_conv_layers[0]
wtab = ops.edge_weight_table(adj_attr, g, *conv.nnConv._edge_mlp_params(), 32)
for _ in range(5):
    if which == 'nnconv':
        ops.nnconv_mean(h, g, wtab, conv.nnConv.root, conv.nnConv.bias, ops.ACT_LEAKY_RELU, ops.new_partials(32, dev), kernel="eg")
    elif which == 'nnconv_cols':
        ops.nnconv_mean(h, g, wtab, conv.nnConv.root, conv.nnConv.bias, ops.ACT_LEAKY_RELU, ops.new_partials(32, dev), kernel="cols_f16")
    elif which == 'nnconv_bf16x3':
        ops.nnconv_mean(h, g, wtab, conv.nnConv.root, conv.nnConv.bias, ops.ACT_LEAKY_RELU, ops.new_partials(32, dev), kernel="cols")
    else:
        net.brch_2_coll_conv_layers[0](h, col)
torch.cuda.synchronize()
