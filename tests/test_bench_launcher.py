"""bench.py --gpus N starts its own N ranks when no launcher is around it.  The launcher alone runs on the CPU
(gloo); the full path with one rank on the GPU."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, cwd=REPO, env=env, capture_output=True, text=True,
                          timeout=timeout)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2])
def test_launcher_starts_n_ranks_and_prints_one_line(n):
    r = _run(["--gpus", str(n), "--spawn", "--launcher-selftest"], 300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line == {"launcher_selftest": True, "ranks_seen": n, "n_gpus": n}


def test_more_ranks_than_gpus_fails_loudly_before_anything_starts():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    want = max(have + 1, 2)
    r = _run(["--gpus", str(want)], 120)
    assert r.returncode != 0
    assert f"needs {want} visible GPUs" in r.stderr and not r.stdout.strip()


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--launcher-selftest"], cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_one_rank_through_the_spawn_path_on_the_gpu():
    """The driver's N = 8 command is this one with another number: `python bench.py --gpus N ...` without a launcher."""
    r = _run(["--gpus", "1", "--spawn", "--force-sharded", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra-sizes",
              "--no-train-step"], 900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["collectives"]["ranks_seen"] == 1 and line["value"] > 0
