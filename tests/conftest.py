import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


@pytest.fixture
def general_schedule():
    """Layouts of up to 4 096 nodes normally run the whole forward as one persistent kernel (csrc/forward_small.hip), layouts
    of up to 65 536 nodes their layer loop (csrc/forward_mid.hip); both associate the BatchNorm / tile sums differently than the
    general launch schedule.  Tests that compare the general schedule's variants bit for bit (two streams vs one, sharded vs
    unsharded) switch both off."""
    from tilingnn_amd import _lib
    before = _lib.lib.tgnn_get_small_layout_limit(), _lib.lib.tgnn_get_mid_layout_limit()
    _lib.lib.tgnn_set_small_layout_limit(0)
    _lib.lib.tgnn_set_mid_layout_limit(0)
    try:
        yield
    finally:
        _lib.lib.tgnn_set_small_layout_limit(before[0])
        _lib.lib.tgnn_set_mid_layout_limit(before[1])


@pytest.fixture
def bf16x3_split():
    """The single-device general schedule and the fused sharded schedule (one all-to-all per layer) run their matrix-core kernels
    with the fp16 x 2 split (tgnn_set_split_precision); the per-op entry points, eval mode and the all-reduce + all-to-all sharded
    scheme run bf16 x 3: tests that compare the two at rounding level put everything on bf16 x 3."""
    from tilingnn_amd import _lib
    before = _lib.lib.tgnn_set_split_precision(0)
    try:
        yield
    finally:
        _lib.lib.tgnn_set_split_precision(before)
