"""Small layout (1254 nodes), ONE forward at a time with a host synchronise behind it (bench.py's config-0 measure): the pre-pass /
polling switches off (lean 11, poll 0) and on, alternating; median of 30 each."""
import sys, time, torch
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN
from tilingnn_amd._lib import lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
sg = make_super_graph(1254, 8502, 10472, tile_count=2, n_edge_types=13, seed=1)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
net.cache_graph = False
def med(lean, poll):
    lib.tgnn_set_lean_head(lean); lib.tgnn_set_prep_words_poll(poll)
    for _ in range(5): net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
    torch.cuda.synchronize(); ts = []
    for _ in range(30):
        t = time.perf_counter(); net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    return sorted(ts)[15]
for rep in range(4):
    print("old (no pre-pass, copy + sync) %.4f | poll only %.4f | pre-pass + poll %.4f" % (med(11, 0), med(11, 1), med(3, 1)), flush=True)
lib.tgnn_set_lean_head(3); lib.tgnn_set_prep_words_poll(1)
