"""One of two processes sharing a GPU (tests/test_two_processes.py): `reps` persistent forwards of an n-node layout; prints the
number of forwards whose health check passed and the number that fell back to the general schedule."""
import json
import os
import sys
import warnings

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def main():
    n, reps, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    from tilingnn_amd import TilinGNN, _lib
    from tilingnn_amd.synth import make_super_graph
    from tilingnn_amd.weights import make_state_dict
    dev = torch.device("cuda:0")
    sg = make_super_graph(n, 8 * n, 10 * n, tile_count=2, n_edge_types=13, seed=seed)
    x, adj, attr, col, _ = sg.to_torch(dev)
    net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
    net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0), strict=True)
    net = net.to(dev).train()
    _lib.lib.tgnn_set_spin_budget_us(100000)
    want = None
    fell_back = 0
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for _ in range(reps):
            p = net.forward_checked(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0]
            assert bool(torch.isfinite(p).all())
            if want is None and not caught:
                want = p.clone()
        fell_back = sum("gave up" in str(w.message) for w in caught)
    torch.cuda.synchronize()
    print("OK " + json.dumps({"forwards": reps, "fell_back": fell_back, "paths": _lib.forward_path_counts()}), flush=True)


if __name__ == "__main__":
    main()
