"""Probabilities of one forward at the benchmark shape into argv[1] (or compared with it, bit for bit, when it exists)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
for n in (100_000, 40_000):
    sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
    x, adj, attr, col, _ = sg.to_torch(dev)
    net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
    from tilingnn_amd._lib import lib
    lib.tgnn_set_mid_layout_limit(0)
    p = net(x, adj, attr, col)[0].cpu()
    rm = net.brch_2_coll_conv_layers[7].batch_norm.running_mean.cpu()
    f = f"{sys.argv[1]}_{n}.pt"
    if os.path.exists(f):
        q, rq = torch.load(f)
        print(n, os.path.basename(os.environ.get("TGNN_LIB_PATH", "default")), "identical to the first library:", bool(torch.equal(p, q)), bool(torch.equal(rm, rq)),
              float((p - q).abs().max()))
    else:
        torch.save((p, rm), f)
        print(n, "saved")
