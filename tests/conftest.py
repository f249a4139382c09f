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


@pytest.fixture
def debug_hooks(request):
    """Tests that need the tgnn_debug_* hooks: the production library does not export them.  Inside the debug build
    (TGNN_LIB_PATH = tilingnn_amd/libtgnn_debug.so) the fixture is False and the test runs; inside the production library the SAME
    test is run in a subprocess against the debug build, its verdict is asserted, and the fixture is True (the test returns)."""
    import os
    import subprocess
    import sys
    from tilingnn_amd import _lib
    if _lib.has_debug_hooks():
        return False
    assert os.path.exists(_lib.DEBUG_LIB_PATH), f"{_lib.DEBUG_LIB_PATH} is missing: make -C tilingnn_amd/csrc debug"
    env = dict(os.environ, TGNN_LIB_PATH=_lib.DEBUG_LIB_PATH)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", request.node.nodeid, "-x", "-q", "-p", "no:cacheprovider"], cwd=repo, env=env,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, "under libtgnn_debug.so:\n" + r.stdout[-3000:] + r.stderr[-1000:]
    return True
