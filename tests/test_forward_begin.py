"""[r6] The general schedule's head: `tgnn_set_lean_head` (no memsets, early edge-weight event, the init MLP as three recomputing
launches -- csrc/init_mlp.hip) and `tgnn_forward_begin` / `tgnn_forward_resume` (the head queued beside a NEW layout's preparation)
must give the plain forward's probabilities BIT FOR BIT (reference: graph_networks/networks/TilinGNN.py:51-78; the head is :54)."""
import pytest
import torch

from tests.test_hip_parity import make_net

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _layout(dev, n, n_types=13, seed=5):
    from tilingnn_amd.synth import make_super_graph
    sg = make_super_graph(n, 8 * n, 10 * n, tile_count=2, n_edge_types=n_types, seed=seed)
    return sg.to_torch(dev)[:4]


@pytest.mark.parametrize("n", [40_000, 70_000])
def test_lean_head_and_fused_init_are_the_launch_per_op_head_bit_for_bit(dev, n):
    """40 000 rows: every kernel behind the head is the same in both settings -> identical probabilities.  70 000 rows: the lean head also
    puts the final MLP's fourth Linear on fp16 pairs (the resident kernel) -- same to 2e-6, and slot 0 (the init MLP) still bit for bit."""
    from tilingnn_amd._lib import lib
    x, adj, attr, col = _layout(dev, n)
    net, _ = make_net(dev)
    outs = {}
    prev = lib.tgnn_set_lean_head(-1)
    try:
        for mode in (0, 1, 3):
            lib.tgnn_set_lean_head(mode)
            outs[mode] = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].clone()
    finally:
        lib.tgnn_set_lean_head(prev)
    assert torch.equal(outs[1], outs[3])                       # the fused init MLP against the five launches, same tail
    if n < 49152:
        assert torch.equal(outs[0], outs[3])
    else:
        assert float((outs[0] - outs[3]).abs().max()) < 2e-6


@pytest.mark.parametrize("n,n_types,hub", [(40_000, 13, 0), (70_000, 13, 0), (40_000, 20, 0), (40_000, 13, 3000)])
def test_begin_resume_is_the_plain_forward(dev, n, n_types, hub):
    """A new layout (cache off / a cache miss) takes tgnn_forward_begin + tgnn_forward_resume, a cached one the plain tgnn_forward: the
    same bits.  20 edge types (a larger workspace than begin carved: the whole forward again, bit 1 of update_running) and an
    in-degree above 2 048 (resume without the edge groups): begin's work is not picked up, and the init MLP's running statistics still
    get ONE update."""
    x, adj, attr, col = _layout(dev, n, n_types=n_types)
    if hub:                                          # (one node with an in-degree above 2 048: no edge groups either)
        adj = adj.clone()
        adj[1, :hub] = 5
    fe = attr.shape[1]
    ref, _ = make_net(dev, fe=fe)
    ref(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)       # cache miss: begin / resume
    want = ref(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0].clone()   # cache hit: plain
    net, _ = make_net(dev, fe=fe)
    net.cache_graph = False
    got = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0]
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert int(net.init_node_feature_trans.mlp[1].batch_norm.num_batches_tracked) == 1
    assert int(net.final_mlp[0].mlp[3].batch_norm.num_batches_tracked) == 1
    fresh, _ = make_net(dev, fe=fe)
    fresh(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
    assert torch.allclose(net.init_node_feature_trans.mlp[0].batch_norm.running_mean, fresh.init_node_feature_trans.mlp[0].batch_norm.running_mean,
                          rtol=1e-6, atol=1e-8)


def test_an_index_error_behind_begin_leaves_nothing_in_flight(dev):
    """prepare_graph raises (an edge index out of range) AFTER tgnn_forward_begin has queued its launches on the side stream: the
    error comes through, and the next forward is the clean one."""
    x, adj, attr, col = _layout(dev, 40_000)
    net, _ = make_net(dev)
    net.cache_graph = False
    bad = adj.clone()
    bad[0, 7] = 40_000
    with pytest.raises(IndexError):
        net(x=x, adj_e_index=bad, adj_e_features=attr, col_e_idx=col)
    ref, _ = make_net(dev)
    want = ref(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0]
    got = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0]
    assert torch.equal(got, want)
