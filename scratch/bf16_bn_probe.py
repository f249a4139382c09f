"""How much does bf16 storage of the PRE-BatchNorm branch outputs cost after the BatchNorm? (labyrinth graph, width 64)"""
import sys, torch
sys.path.insert(0, '.')
from oracle import tilingnn_oracle as orc
from tests.golden_util import graph_tensors, load_labyrinth_graph
from tests.test_bf16_path import make_net, bf, W
from tilingnn_amd import ops, ops_bf16
dev = torch.device('cuda:0')
g = graph_tensors(load_labyrinth_graph(), torch.float32, dev)
x, adj, attr, col, _ = g
n = 1254
for depth, i in ((3, 0), (3, 1), (3, 2)):
    net, sd = make_net(dev, depth=depth)
    sd64 = orc.cast_sd(sd, torch.float64)
    gen = torch.Generator().manual_seed(7 + i)
    h = bf(torch.randn(n, W, generator=gen))
    graph = ops.prepare_graph(n, adj, attr, col)
    p1, p2 = f"brch_1_graph_conv_layers.{i}", f"brch_2_coll_conv_layers.{i}"
    with torch.no_grad():
        want_g = orc.graph_conv(h.double(), adj.cpu(), attr.cpu().double(), sd64, p1)
        want_c = orc.coll_conv(h.double(), col.cpu(), sd64, p2)
        pre_c = torch.nn.functional.leaky_relu(orc.gin_conv(h.double(), col.cpu(), sd64, p2))
    l1, l2 = net.brch_1_graph_conv_layers[i], net.brch_2_coll_conv_layers[i]
    hb = h.to(dev).to(torch.bfloat16)
    wtab = ops.edge_weight_table(attr, graph, *l1.nnConv._edge_mlp_params(), W)
    pa, pb = ops.new_partials(W, dev), ops.new_partials(W, dev)
    a1, n1 = ops_bf16.nnconv64(hb, graph, wtab, l1.nnConv.root, l1.nnConv.bias, ops.ACT_LEAKY_RELU, pa)
    a2, n2 = ops_bf16.gin64(hb, graph, l2.ginConv.eps, *l2.ginConv._mlp_params(), act=ops.ACT_LEAKY_RELU, partials=pb)
    s1 = ops.bn_finalize(pa, n1, n, l1.batch_norm, True)
    s2 = ops.bn_finalize(pb, n2, n, l2.batch_norm, True)
    g1 = ops.bn_apply(a1.float(), s1).cpu()
    g2 = ops.bn_apply(a2.float(), s2).cpu()
    std = pre_c.std(0)
    print(f"layer {i}: GraphConv+BN err {orc.rel_max_err(g1, want_g):.2e}  CollConv+BN err {orc.rel_max_err(g2, want_c):.2e}  "
          f"pre-BN GIN column std: min {float(std.min()):.2e} median {float(std.median()):.2e}; mean |col| {float(pre_c.abs().mean()):.2f}")
