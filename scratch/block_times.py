"""When do the column NNConv's blocks start and end inside the two-stream forward?  (TGNN_LIB_PATH=scratch/libs/libtgnn_BLOCKTIMES.so)
The kernel runs one block per CU on CUs - 32 CUs, every block a fixed share of the tiles: a block that has to wait for a CU the
collision chain holds finishes late and the launch with it."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilingnn_amd import TilinGNN, _lib
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
sg = make_super_graph(100_000, 1_000_000, 1_250_000, tile_count=2, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
raw = C.CDLL(os.environ["TGNN_LIB_PATH"])
buf = (C.c_ulonglong * 512)()
import numpy as np
for two in (1, 0):
    os.environ["TGNN_TWO_STREAMS"] = str(two)
    _lib._side_streams.clear()
    for _ in range(5):
        net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
    res = []
    for rep in range(10):
        net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
        torch.cuda.synchronize()
        assert raw.tgnn_debug_block_times(buf) == 0
        t = np.array(buf[:], dtype=np.float64).reshape(2, 256)[:, :224] / 100.0     # us (100 MHz clock); the LAST NNConv launch of the forward
        s0 = t[0].min()
        res.append((t[0].max() - s0, np.median(t[1] - t[0]), (t[1] - t[0]).max(), t[1].max() - s0, np.sort(t[0] - s0)[[112, 200, 216, 223]]))
    r = np.array([[a, b, c, d] for a, b, c, d, _ in res])
    print(f"two_streams={two}: last block starts {r[:,0].mean():.1f} us after the first (max {r[:,0].max():.1f}); a block runs {r[:,1].mean():.1f} us "
          f"(median), {r[:,2].mean():.1f} (slowest); first in -> last out {r[:,3].mean():.1f} us; start offsets of blocks #112/#200/#216/#223 (sorted): "
          f"{np.mean([q for *_, q in res], axis=0).round(1)}")
