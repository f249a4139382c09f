"""ONE running-statistics update per forward(), on every path that may launch a forward twice.
The reference runs inference in train mode (/root/reference/solver/ml_solver/ml_solver.py:129-131): every predict() applies one
momentum update to the 46 BatchNorms' running buffers and bumps num_batches_tracked by one -- they are state a checkpoint carries."""
import pytest
import torch

from tests.test_hip_parity import make_net

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _layout(dev, n, seed=3):
    from tilingnn_amd.synth import make_super_graph
    sg = make_super_graph(n, 8 * n, 10 * n, tile_count=2, n_edge_types=13, seed=seed)
    return sg.to_torch(dev)[:4]


def _buffers(net):
    return {k: v.clone() for k, v in net.state_dict().items() if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}


def _assert_one_update(got, want):
    for k, v in want.items():
        if k.endswith("num_batches_tracked"):
            assert int(got[k]) == 1 == int(v), k
        else:
            assert torch.allclose(got[k], v, rtol=1e-5, atol=1e-7), (k, float((got[k] - v).abs().max()))


@pytest.mark.parametrize("n", [10_000, 40_000])
def test_uncached_layout_updates_once(dev, n):
    """A just-prepared layout (mid-size: optimistic batches; general schedule: tgnn_forward_begin / resume) against the cached one."""
    x, adj, attr, col = _layout(dev, n)
    ref, _ = make_net(dev)
    ref(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)        # cached path (first call prepares, nothing is repeated)
    want = _buffers(ref)
    net, _ = make_net(dev)
    net.cache_graph = False
    net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
    torch.cuda.synchronize()
    _assert_one_update(_buffers(net), want)


def test_swapped_parameter_storage_updates_once(dev):
    """`p.data = t` under a live Parameter: the cached pointer table is stale, the forward is repeated -- without a second update."""
    x, adj, attr, col = _layout(dev, 10_000)
    ref, _ = make_net(dev)
    probs_ref = ref(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0]
    want = _buffers(ref)
    net, _ = make_net(dev)
    net._param_table()                                                    # (the table is cached ...)
    w = net.final_mlp[0].mlp[0].linear.weight
    w.data = w.data.clone()                                               # (... and now points at a storage the Parameter left)
    probs = net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)[0]
    torch.cuda.synchronize()
    assert torch.equal(probs, probs_ref)
    _assert_one_update(_buffers(net), want)
