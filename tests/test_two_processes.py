"""Two PROCESSES on one GPU, each running the persistent small-layout kernel (csrc/forward_small.hip: its blocks wait for each
other).  The launch gate that keeps such kernels of ONE process from starving each other cannot see the other process; what makes
this safe is that every wait is bounded (csrc/forward_persist.h): a starved kernel gives up, the module repeats the forward on the
general schedule (TilinGNN.forward_checked, the call ML_Solver.predict makes).  Both workers must FINISH, with finite results."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [2000, 3000])
def test_two_processes_share_the_gpu_and_finish(n):
    """2 000 nodes: 125 blocks each, both kernels fit the 256 CUs side by side; 3 000 nodes: 188 blocks each -- they do NOT fit
    together, the very case the bounded spins exist for."""
    worker = os.path.join(REPO, "tests", "two_process_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(n), "50", str(k)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for k in (1, 2)]
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a worker did not finish: a persistent kernel is waiting without bound")
        assert p.returncode == 0, err[-2000:]
        line = [l for l in out.splitlines() if l.startswith("OK ")]
        assert line, out[-2000:] + err[-2000:]
        outs.append(json.loads(line[-1][3:]))
    print(outs)
    for o in outs:
        assert o["forwards"] == 50
