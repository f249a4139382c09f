"""Worker of tests/test_dist_gpu.py: one RCCL rank (backend "nccl" IS RCCL on ROCm).  Initialises the process group, runs
`ShardedTilinGNN.step()` -- tgnn_forward_sharded with its collectives going through torch.distributed from host
callbacks -- and compares with the unsharded forward of the same network.  Prints one line `OK {...}` on success."""
import json
import os
import sys


import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n, depth = int(sys.argv[1]), int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dev = torch.device("cuda", torch.cuda.current_device())
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)                                   # RCCL's banner goes to fd 1: keep stdout for the verdict
    dist.init_process_group("nccl", device_id=dev)
    from tilingnn_amd import TilinGNN
    from tilingnn_amd.dist import ShardedTilinGNN
    from tilingnn_amd.synth import make_super_graph
    from tilingnn_amd.weights import make_state_dict
    sg = make_super_graph(n, 10 * n, 12 * n, tile_count=2, n_edge_types=13, seed=5)
    net = TilinGNN(adj_edge_features_dim=15, network_depth=depth, network_width=32, node_features_dim=3)
    net.load_state_dict(make_state_dict(15, depth, 32, 1, 3, seed=0), strict=True)
    net = net.to(dev).train()
    runner = ShardedTilinGNN(net, sg, rank, world, dev)
    got = runner.step()
    torch.cuda.synchronize()
    dist.barrier()
    ref_net = TilinGNN(adj_edge_features_dim=15, network_depth=depth, network_width=32, node_features_dim=3)
    ref_net.load_state_dict(make_state_dict(15, depth, 32, 1, 3, seed=0), strict=True)
    ref_net = ref_net.to(dev).train()
    from tilingnn_amd import _lib
    _lib.lib.tgnn_set_small_layout_limit(0)      # the sharded step runs the general schedule's kernels: compare with those
    _lib.lib.tgnn_set_mid_layout_limit(0)
    x, adj, attr, col, _ = sg.to_torch(dev)
    want = ref_net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0]
    lo = runner.shard.lo
    want_own = want[lo:lo + runner.n_local]
    diff = float((got.double() - want_own.double()).abs().max())
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "max_abs_diff": diff,
           "bit_identical": bool(torch.equal(got, want_own)), "collectives": runner.collectives_per_forward,
           "running_mean_equal": bool(torch.equal(net.final_mlp[0].mlp[0].batch_norm.running_mean,
                                                  ref_net.final_mlp[0].mlp[0].batch_norm.running_mean))}
    dist.destroy_process_group()
    os.dup2(saved, 1)
    print("OK " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
