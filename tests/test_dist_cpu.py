"""Multi-GPU path on CPU: partitioning, halo exchange and the BatchNorm all-reduce schedule of
tilingnn_amd.dist, driven (a) by LocalSimComm with 1/2/3/4 virtual ranks and (b) by two real processes over
gloo, with the oracle-backed CPU backend standing in for the HIP kernels.  The sharded result must equal the
single-process oracle forward (fp64: only the BN-sum association differs)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import tilingnn_oracle as orc
from tests.dist_cpu_backend import OracleBackend
from tests.golden_util import graph_tensors, load_labyrinth_graph
from tilingnn_amd import dist as tdist
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _param_holder(sd, depth, width, fe, fx):
    """A module tree with the reference's names holding fp64 tensors (no GPU library import)."""
    import types

    def lin(prefix):
        w = sd[prefix + ".weight"].double()
        return types.SimpleNamespace(weight=w, bias=sd[prefix + ".bias"].double(), out_features=w.shape[0])

    def bnm(prefix):
        return types.SimpleNamespace(weight=sd[prefix + ".weight"].double(), bias=sd[prefix + ".bias"].double(), eps=1e-5,
                                     num_features=sd[prefix + ".weight"].shape[0])

    def lt(prefix, bn=True):
        return types.SimpleNamespace(linear=lin(prefix + ".linear"), batch_norm=bnm(prefix + ".batch_norm") if bn else None)

    def mlp(prefix, n, bn):
        return types.SimpleNamespace(mlp=[lt(f"{prefix}.mlp.{i}", bn) for i in range(n)])
    net = types.SimpleNamespace(network_width=width, network_depth=depth)
    net.init_node_feature_trans = mlp("init_node_feature_trans", 2, True)
    net.brch_1_graph_conv_layers, net.brch_2_coll_conv_layers = [], []
    for i in range(depth):
        p1, p2 = f"brch_1_graph_conv_layers.{i}", f"brch_2_coll_conv_layers.{i}"
        conv = types.SimpleNamespace(nn=mlp(p1 + ".mlp", 3, False), root=sd[p1 + ".nnConv.root"].double(),
                                     bias=sd[p1 + ".nnConv.bias"].double(), in_channels=width, out_channels=width)
        net.brch_1_graph_conv_layers.append(types.SimpleNamespace(nnConv=conv, batch_norm=bnm(p1 + ".batch_norm")))
        gconv = types.SimpleNamespace(nn=mlp(p2 + ".ginConv.nn", 3, False), eps=sd[p2 + ".ginConv.eps"].double())
        net.brch_2_coll_conv_layers.append(types.SimpleNamespace(ginConv=gconv, batch_norm=bnm(p2 + ".batch_norm")))
    net.final_mlp = [mlp("final_mlp.0", 4, True), lt("final_mlp.1", bn=False)]
    return net


def _graph(kind):
    if kind == "labyrinth":
        g = load_labyrinth_graph()
        return g["x"].astype(np.float64), g["adj"], g["adj_attr"].astype(np.float64), g["col"]
    if kind == "config4_20k":      # BASELINE configs[3] (500k nodes, 6M + 7.5M edges, two tile classes) at 1 / 25 of its size
        sg = make_super_graph(20000, 240000, 300000, tile_count=2, n_edge_types=13, seed=7)
    else:
        sg = make_super_graph(3000, 24000, 30000, tile_count=2, n_edge_types=13, seed=5)
    return sg.node_feature, sg.align_edge_index, sg.align_edge_features, sg.collide_edge_index


@pytest.mark.parametrize("kind", ["labyrinth", "synthetic"])
def test_shards_partition_the_graph(kind):
    x, adj, attr, col = _graph(kind)
    n = x.shape[0]
    for world in (1, 2, 3, 8):
        shards = [tdist.make_shard(x, adj, attr, col, r, world) for r in range(world)]
        tdist.LocalSimComm.setup(shards)
        assert sum(s.n_own for s in shards) == n
        assert sum(s.adj.shape[1] for s in shards) == adj.shape[1]
        assert sum(s.col.shape[1] for s in shards) == col.shape[1]
        for s in shards:
            lo, hi = tdist.node_range(n, s.rank, world)
            assert (s.lo, s.n_own) == (lo, hi - lo)
            # local ids map back to the global edges, in global order
            glob = np.concatenate([np.arange(lo, hi), s.halo_ids])
            keep = (adj[1] >= lo) & (adj[1] < hi)
            np.testing.assert_array_equal(glob[s.adj[0]], adj[0][keep])
            np.testing.assert_array_equal(s.adj[1] + lo, adj[1][keep])
            assert ((s.halo_ids < lo) | (s.halo_ids >= hi)).all() and sum(s.recv_counts) == s.halo_ids.shape[0]
            owners = tdist.owner_of(s.halo_ids, n, world)
            assert (np.diff(owners) >= 0).all()                      # grouped by owner rank
            for p in range(world):                                   # what p sends me == my halo rows owned by p
                got = shards[p].send_ids[s.rank] + shards[p].lo
                np.testing.assert_array_equal(got, s.halo_ids[owners == p])


@pytest.mark.parametrize("world", [1, 2, 4])
def test_sharded_forward_equals_oracle_forward(world):
    x, adj, attr, col = _graph("synthetic")
    depth, width, fe, fx = 4, 32, attr.shape[1], x.shape[1]
    sd = make_state_dict(fe, depth, width, 1, fx, seed=2)
    net = _param_holder(sd, depth, width, fe, fx)
    shards = [tdist.make_shard(x, adj, attr, col, r, world) for r in range(world)]
    tdist.LocalSimComm.setup(shards)
    be = OracleBackend()
    progs = [tdist.ShardProgram(net, s, be, update_running=False) for s in shards]
    parts = tdist.LocalSimComm.run(progs)
    got = torch.cat(parts)
    with torch.no_grad():
        want, _ = orc.tilingnn_forward(orc.cast_sd(sd, torch.float64), torch.from_numpy(x), torch.from_numpy(adj),
                                       torch.from_numpy(attr), torch.from_numpy(col))
    assert got.shape == want.shape
    assert float((got - want).abs().max()) < 1e-9


@pytest.mark.parametrize("world", [2, 4])
def test_rounds_of_local_compaction_equal_the_oracle_sub_layouts(world):
    """The greedy loop scores, every round, the sub-layout of the still unlabelled nodes (reference: util/algorithms.py:18-62,
    tiling/brick_layout.py:248-286).  Sharded, every rank cuts its shard of that sub-layout out of its shard of the round
    before (dist.compact_shard: mask -> local compact -> halo-list rebuild).  Three rounds with shrinking alive sets: the
    sharded forward over the compacted shards equals the oracle's forward over oracle.greedy_oracle.compute_sub_layout."""
    from oracle import greedy_oracle as go
    x, adj, attr, col = _graph("synthetic")
    n = x.shape[0]
    depth, width, fe, fx = 3, 32, attr.shape[1], x.shape[1]
    sd = make_state_dict(fe, depth, width, 1, fx, seed=4)
    net = _param_holder(sd, depth, width, fe, fx)
    shards = [tdist.make_shard(x, adj, attr, col, r, world) for r in range(world)]
    rng = np.random.default_rng(9)
    alive_orig = np.ones(n, dtype=bool)                     # over the ORIGINAL numbering (what the oracle cuts from)
    dummy = np.zeros((col.shape[1], 1))
    for keep in (0.7, 0.6, 0.3):
        ids_before = np.flatnonzero(alive_orig)
        alive_orig &= rng.uniform(size=n) < keep
        if keep == 0.6:
            alive_orig[: n // world] = False                # a rank left without a single node
        alive_rel = alive_orig[ids_before]                  # over the numbering of the round before (what the ranks hold)
        shards = [tdist.compact_shard(s, alive_rel) for s in shards]
        tdist.LocalSimComm.setup(shards)
        assert sum(s.n_own for s in shards) == int(alive_orig.sum()) and all(s.n_total == int(alive_orig.sum()) for s in shards)
        sub = go.compute_sub_layout(x, adj, attr, col, dummy, np.flatnonzero(alive_orig))
        for s in shards:                                    # the shard's rows ARE the sub-layout's rows of its range
            np.testing.assert_array_equal(s.x, sub[0][s.lo:s.lo + s.n_own])
        be = OracleBackend()
        got = torch.cat(tdist.LocalSimComm.run([tdist.ShardProgram(net, s, be, update_running=False) for s in shards]))
        with torch.no_grad():
            want, _ = orc.tilingnn_forward(orc.cast_sd(sd, torch.float64), torch.from_numpy(sub[0]), torch.from_numpy(sub[1]),
                                           torch.from_numpy(sub[2]), torch.from_numpy(sub[3]))
        assert got.shape == want.shape and float((got - want).abs().max()) < 1e-9


@pytest.mark.parametrize("kind,dedup", [("labyrinth", False), ("config4_20k", True)])
def test_two_process_gloo_run_matches_oracle(tmp_path, kind, dedup):
    """Real torch.distributed ranks (gloo, world_size 2) through TorchDistComm: setup all-to-all of the halo id
    lists, 21 halo exchanges and 26 BN all-reduces of a depth-20 forward -- on the labyrinth layout and on a layout of
    BASELINE configs[3]'s shape (20 000 nodes, 12 + 15 edges per node; there the edge MLP runs on the distinct attribute rows
    only, on both sides: oracle.nnconv_mean_dedup, pinned to the port)."""
    script = tmp_path / "rank.py"
    script.write_text(f'''
KIND, DEDUP = {kind!r}, {dedup!r}
import sys, os, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {REPO!r})
from oracle import tilingnn_oracle as orc
from tests.dist_cpu_backend import OracleBackend
from tests.test_dist_cpu import _param_holder, _graph
from tilingnn_amd import dist as tdist
from tilingnn_amd.weights import make_state_dict
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
x, adj, attr, col = _graph(KIND)
if DEDUP:
    orc.nnconv_mean = orc.nnconv_mean_dedup
sd = make_state_dict(attr.shape[1], 20, 32, 1, x.shape[1], seed=0)
net = _param_holder(sd, 20, 32, attr.shape[1], x.shape[1])
shard = tdist.make_shard(x, adj, attr, col, rank, world)
comm = tdist.TorchDistComm()
comm.setup(shard)
probs = comm.run(tdist.ShardProgram(net, shard, OracleBackend(dedup_rows=DEDUP), update_running=False))
gathered = [torch.empty(tdist.node_range(x.shape[0], r, world)[1] - tdist.node_range(x.shape[0], r, world)[0], 1,
                        dtype=torch.float64) for r in range(world)]
dist.all_gather(gathered, probs.contiguous())
if rank == 0:
    with torch.no_grad():
        want, _ = orc.tilingnn_forward(orc.cast_sd(sd, torch.float64), torch.from_numpy(x), torch.from_numpy(adj),
                                       torch.from_numpy(attr), torch.from_numpy(col))
    err = float((torch.cat(gathered) - want).abs().max())
    print("MAXERR", err)
    assert err < 1e-6, err
dist.destroy_process_group()
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "MAXERR" in out.stdout
