"""The multi-rank path of bench.py, rehearsed on the one card a test box has: two rank processes share GPU 0
(`--share-device`).  What the driver's N = 2, 4, 8 runs do differently is one card per rank and RCCL instead of gloo for
the script's own barrier / reductions; the sharding of pairs and matrix tiles, the rank-0 line and the exit codes are
the ones exercised here.  bench.py is started as a child process (it starts its ranks itself, before it touches the
GPU)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, env=None, timeout=420):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline", *flags]
    e = dict(os.environ, **(env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    return subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_two_ranks_on_one_card():
    run = _bench("--backend", "gloo")
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                           # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["scaling"] == "weak"
    # whole-job accounting: both ranks' pairs over the slowest rank's time
    per_gpu = out["config"]["pairs_per_gpu"]
    assert out["pairs_per_s"] == pytest.approx(2 * per_gpu / (out["ms_per_step"] * 1e-3), rel=1e-3)
    assert "cpu_baseline" not in out
    matrix = out["extra"]["c4_relation_matrix"]
    assert "error" not in matrix and matrix["n_gpus"] == 2
    gold = load_golden("c4_matrix.json")
    assert matrix["variants"] == gold["n"]
    assert matrix["upper_histogram"] == gold["upper_histogram"]     # the two ranks' tiles, gathered, are the reference's matrix
    assert out["relations_per_s"] == matrix["relations_per_s"] > 0


def test_a_failing_collective_is_a_failing_run():
    """The library's RCCL communicator cannot come up (its librccl is named wrongly): the headline is still printed, the
    matrix's error is in the line, and the job's exit code says FAILED on every rank (3 = the relation matrix raised)."""
    run = _bench("--backend", "gloo", "--matrix-rccl", env={"MALCS_RCCL_LIB": "/nonexistent/librccl.so", "BENCH_MATRIX_TIMEOUT": "60"})
    assert run.returncode == 3, (run.returncode, run.stderr[-2000:])
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert "error" in out["extra"]["c4_relation_matrix"]
    assert "relation matrix failed" in run.stderr
