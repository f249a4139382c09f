"""RCCL under the driver's eyes: the sharded forward with backend "nccl" (= RCCL on ROCm) on the one
GPU of the test box, world size 1 -- communicator creation, all_to_all_single / all_reduce on device buffers and the
host-callback path of tgnn_forward_sharded all execute; the result must equal the unsharded forward bit for bit (one
shard = every sum in the same order).  The world-2 schedule is covered on CPU over gloo (tests/test_dist_cpu.py) and with
thread-simulated ranks on one GPU (tests/test_hip_parity.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("library_rccl", [True, False])
@pytest.mark.parametrize("n,depth", [(3000, 3), (20000, 20)])
def test_nccl_world1_sharded_step_is_the_unsharded_forward(n, depth, library_rccl):
    """library_rccl: the collectives are RCCL calls of the library's own communicators (csrc/rccl_comm.hip: ncclSend / ncclRecv
    groups, ncclAllReduce) with the split exchange -- one all-to-all per branch and layer, the collision branch's on the side
    stream; else host callbacks into torch.distributed, one all-to-all per layer."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29631",
               HSA_ENABLE_IPC_MODE_LEGACY="0", TGNN_LIBRARY_RCCL="1" if library_rccl else "0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "dist_gpu_worker.py"), str(n), str(depth)],
                       env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    ok = [l for l in r.stdout.splitlines() if l.startswith("OK ")]
    assert r.returncode == 0 and ok, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(ok[-1][3:])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["bit_identical"], out
    assert out["running_mean_equal"], out
    c = out["collectives"]
    # one all-to-all per message-passing layer but the last (halo rows of both branches + the BatchNorm sums; split exchange: one
    # per branch) plus the exchange of middle[0]; all-reduces for the BatchNorms that have no halo exchange to ride on: 2 init +
    # 4 final MLP + the last layer's pair (one message)
    assert c["all_to_all_single"] == (2 * depth - 1 if library_rccl else depth) and c["all_reduce"] == 7, c
    assert c["issued_by"].startswith("library" if library_rccl else "torch.distributed"), c
