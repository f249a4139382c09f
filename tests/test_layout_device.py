"""The layout producer on the GPU: a layout cut out of the device-resident complete graph is,
bit for bit, what the REFERENCE's host producer + `to_torch_tensor` upload would have put in HBM."""
import os

import numpy as np
import pytest
import torch

from tests.golden_util import GOLDEN, load_npz

pytestmark = pytest.mark.gpu
SMALL = os.path.join(GOLDEN, "complete_graph_small.pkl")
DEV = "cuda:0"


@pytest.fixture(scope="module")
def graph():
    from tilingnn_amd.tiling.tile_graph import TileGraph
    g = TileGraph(2)
    g.load_graph_state(SMALL, sidecar=False)
    return g


@pytest.fixture(scope="module")
def on_device(graph):
    from tilingnn_amd.util.data_util import CompleteGraphOnDevice
    return CompleteGraphOnDevice(graph, DEV)


def _expected(ref, case):
    """What data_util.py:110-117 makes of the reference producer's arrays."""
    f = lambda k, dt: torch.from_numpy(np.ascontiguousarray(ref[f"{case}.{k}"])).to(dt)
    x = f("x", torch.float32)
    adj = f("adj", torch.int64).reshape(2, -1)
    col = f("col", torch.int64).reshape(2, -1)
    attr = f("adj_attr", torch.float32).reshape(adj.shape[1], -1) if adj.shape[1] else None
    return x, adj, attr, col


@pytest.mark.parametrize("case", ["small.first80", "small.random60", "small.all", "small.isolated"])
def test_device_producer_matches_reference(on_device, case):
    ref = load_npz("ref_layouts.npz")
    tiles = ref[f"{case}.super_tiles"]
    lay = on_device.layout(tiles)
    x, adj, attr, col = _expected(ref, case)
    assert torch.equal(lay.node_feature.cpu(), x)
    assert torch.equal(lay.align_edge_index.cpu(), adj) and torch.equal(lay.collide_edge_index.cpu(), col)
    if adj.shape[1]:
        assert torch.equal(lay.align_edge_features.cpu(), attr)
    assert lay.inverse_index.cpu().tolist() == tiles.tolist()


def test_device_producer_wants_ascending_tiles(on_device):
    with pytest.raises(ValueError):
        on_device.layout([3, 1, 2])
    with pytest.raises(ValueError):
        on_device.layout([0, 150])


def test_predict_on_a_device_produced_layout(graph, on_device):
    """ML_Solver.predict reads the device layout directly; same bits as the host-produced arrays uploaded the
    reference's way."""
    from tilingnn_amd.graph_networks.networks.TilinGNN import TilinGNN
    from tilingnn_amd.tiling.brick_layout import BrickLayout
    from tilingnn_amd.util import data_util as du
    torch.manual_seed(3)
    net = TilinGNN(adj_edge_features_dim=graph.total_feature_dim, network_depth=4, network_width=32,
                   node_features_dim=graph.tile_type_count + 1).to(DEV)
    tiles = list(range(10, 130))
    lay = on_device.layout(tiles)
    with torch.no_grad():
        got, *_ = net(lay.node_feature, lay.align_edge_index, lay.align_edge_features, lay.collide_edge_index)
        got = got.clone()
        x, ci, cf, ai, af, re_index = du.create_brick_layout_from_super_set(graph, tiles)
        host = BrickLayout(graph, x, ci, cf, ai, af, re_index)
        hx, hadj, hattr, hcol, _ = host.get_data_as_torch_tensor(DEV)
        want, *_ = net(hx, hadj, hattr, hcol)
    assert torch.equal(got, want)
