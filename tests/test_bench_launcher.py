"""`python bench.py --gpus N` must start its own rank processes (the driver calls it without a launcher) and relay
rank 0's single JSON line; a failing rank must end the run with a non-zero status instead of a hang."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "_launch_child.py")


def _run(code):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("n", [2, 3])
def test_launcher_relays_rank0_line(n):
    r = _run(f"import bench; bench.launch_ranks({n}, [], script={CHILD!r})")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out == {"n_gpus": n, "sum": n * (n + 1) / 2, "local": "0", "addr": "127.0.0.1"}


def test_launcher_fails_when_a_rank_fails():
    r = _run(f"import bench; bench.launch_ranks(2, ['--fail'], script={CHILD!r})")
    assert r.returncode != 0
    assert "rank exit codes" in r.stderr


def test_bench_decides_to_launch_only_without_a_launcher():
    src = open(os.path.join(ROOT, "bench.py")).read()
    # the parent must decide before importing torch / touching the GPU
    assert src.index("launch_ranks(args.gpus") < src.index("import torch\n    import torch.distributed as dist")


def test_rccl_run_refuses_more_ranks_than_devices():
    """--backend nccl with WORLD_SIZE > visible GPUs must end at once with a non-zero status and say why (two ranks on one
    device fail or hang in RCCL's communicator set-up); here no GPU is visible at all."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "nccl"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "one GPU per rank" in r.stderr and not r.stdout.strip()


def test_launcher_stops_hung_ranks():
    r = _run(f"import bench; bench.launch_ranks(2, ['--hang'], script={CHILD!r}, timeout_s=3.0)")
    assert r.returncode != 0 and "still running" in r.stderr
