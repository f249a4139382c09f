"""Sharded schedule at world 1 (nccl): one side stream for the collision branch vs everything on one stream."""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
from tilingnn_amd import TilinGNN
from tilingnn_amd.dist import ShardedTilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device("cuda:0")
sg = make_super_graph(100_000, 1_000_000, 1_250_000, tile_count=2, n_edge_types=13, seed=2)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
runner = ShardedTilinGNN(net, sg, 0, 1, dev)
for two in (False, True, False, True):
    runner.fused.two_streams = two
    for _ in range(5): runner.step()
    torch.cuda.synchronize(); ts = []
    for _ in range(20):
        t = time.perf_counter(); runner.step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"two_streams={two}: median {sorted(ts)[10]:.3f} ms min {min(ts):.3f}")
dist.destroy_process_group()
