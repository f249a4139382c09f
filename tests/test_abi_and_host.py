"""CPU-side checks: the C-ABI library loads and exports every symbol include/tgnn.h declares (no compute
call without a GPU), the host-side mirror keeps the reference's module/state-dict contract, and the
product path refuses to run without a GPU instead of falling back."""
import copy
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(REPO, "include", "tgnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef TGNN_DEBUG.*?#endif", "", text, flags=re.S)       # the test hooks of libtgnn_debug.so
    return sorted(set(re.findall(r"\b(tgnn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tilingnn_amd import _lib
    declared = header_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(_lib.lib, name), f"{name} is declared in include/tgnn.h but not exported by libtgnn.so"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared, "ctypes binding table and header disagree"
    assert _lib.lib.tgnn_version() == 100
    # the production library carries no test hook (they live in libtgnn_debug.so: make -C tilingnn_amd/csrc debug)
    if not os.environ.get("TGNN_LIB_PATH"):
        assert not any(hasattr(_lib.lib, n) for n in ("tgnn_debug_spin_fault", "tgnn_debug_set_csr_bucket_cap", "tgnn_debug_set_block_caps"))


def test_param_table_names_are_the_reference_state_dict_keys():
    from tilingnn_amd import _lib
    from tilingnn_amd.weights import state_dict_spec
    dims = _lib.ModelDims(3, 15, 32, 20, 1)
    names = _lib.param_names(dims)
    spec = state_dict_spec(15, 20, 32, 1, 3)
    assert len(names) == 544 and len(set(names)) == 544
    assert set(names) == {k for k in spec if ".nnConv.nn." not in k}     # aliases of <p>.mlp.* are passed once


def test_workspace_size_queries_and_argument_validation_without_gpu():
    import ctypes as C
    from tilingnn_amd import _lib
    lib = _lib.lib
    dims = _lib.ModelDims(3, 15, 32, 20, 1)
    small = lib.tgnn_forward_workspace_bytes(C.byref(dims), 1000, 13)
    big = lib.tgnn_forward_workspace_bytes(C.byref(dims), 100000, 13)
    assert 0 < small < big < 2 ** 33
    assert lib.tgnn_csr_workspace_bytes(1000, 10000) > 0 and lib.tgnn_edge_dedup_workspace_bytes(10000, 15) > 0
    assert lib.tgnn_nnconv_cols_max_columns(1000, 10000) >= 10000 + 63
    assert lib.tgnn_nnconv_cols_max_types() >= 13
    bad = _lib.ModelDims(3, 15, 30, 20, 1)                               # width must be a multiple of 4
    assert lib.tgnn_param_count(C.byref(bad)) == -1
    # invalid arguments are rejected before any launch (no GPU needed) with a message
    rc = lib.tgnn_bn_finalize(7, None, 0, None, 32, 10, None, None, 1e-5, 0.1, None, None, None, None, None)
    assert rc == -1 and b"mode" in lib.tgnn_last_error()


def test_module_contract_matches_reference():
    from tilingnn_amd import TilinGNN
    from tilingnn_amd.weights import make_state_dict, state_dict_spec
    net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
    sd = net.state_dict()
    spec = state_dict_spec(15, 20, 32, 1, 3)
    assert list(sd) == list(spec) and all(tuple(sd[k].shape) == tuple(v) for k, v in spec.items())
    assert sum(p.numel() for p in net.parameters()) == 1_730_145
    net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0), strict=True)
    # the edge MLP is one module registered twice (edge_conv.py:17-18)
    l0 = net.brch_1_graph_conv_layers[0]
    assert l0.mlp is l0.nnConv.nn
    assert (net.network_depth, net.network_width, net.residual_skip_num) == (20, 32, 2)
    clone = copy.deepcopy(net)                                            # ml_solver.py:26
    assert clone is not net and torch.equal(clone.state_dict()["final_mlp.1.linear.weight"],
                                            sd["final_mlp.1.linear.weight"])
    assert net.training and isinstance(net.train(), TilinGNN) and not net.eval().training


def test_no_cpu_fallback():
    from tilingnn_amd import TilinGNN
    from tilingnn_amd.graph_networks.layers.util import MLP
    net = TilinGNN(adj_edge_features_dim=15, network_depth=2, network_width=32, node_features_dim=3)
    x = torch.zeros(8, 3); ei = torch.zeros(2, 4, dtype=torch.long); ea = torch.zeros(4, 15)
    with pytest.raises(RuntimeError, match="no CPU"):
        net(x=x, adj_e_index=ei, adj_e_features=ea, col_e_idx=ei)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        MLP(3, 8, [8], torch.nn.LeakyReLU())(x)


def test_package_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "tilingnn_amd")):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"


def test_synthetic_generator_matches_spec():
    from tilingnn_amd.synth import make_super_graph
    sg = make_super_graph(2000, 16000, 20000, tile_count=2, n_edge_types=13, seed=1)
    assert sg.node_feature.shape == (2000, 3) and sg.align_edge_features.shape == (16000, 15)
    for ei in (sg.align_edge_index, sg.collide_edge_index):
        assert ei.dtype == np.int64 and ei.min() >= 0 and ei.max() < 2000
        np.testing.assert_array_equal(ei[:, 0::2], ei[::-1, 1::2])       # (u,v),(v,u) consecutive
        assert (np.abs(ei[0] - ei[1]) <= int(np.ceil(8 * np.sqrt(2000)))).all() and (ei[0] != ei[1]).all()
    key = lambda ei: set(map(tuple, ei.T.tolist()))
    assert not (key(sg.align_edge_index) & key(sg.collide_edge_index))    # adj and collision sets are disjoint
    np.testing.assert_array_equal(sg.align_edge_features[0::2], sg.align_edge_features[1::2])   # symmetric attrs
    assert len(np.unique(sg.align_edge_features, axis=0)) == 13
    sg2 = make_super_graph(2000, 16000, 20000, tile_count=2, n_edge_types=13, seed=1)
    np.testing.assert_array_equal(sg.align_edge_index, sg2.align_edge_index)
