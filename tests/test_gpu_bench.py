"""bench.py as the driver runs it, on small batches: the multi-rank launch path (two ranks started by bench.py itself,
rehearsed with gloo on the one GPU of the box) and the 4K / 200-tag workload whose pose-graph window the configs[4] line
reports (bench.py exits non-zero when that solve is unhealthy: profiles/r02_bench_configs4_n1.json had cost 1.1e16)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=900):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), cwd=ROOT, capture_output=True, timeout=timeout)
    assert p.returncode == 0, (p.returncode, p.stderr.decode(errors="replace")[-2000:])
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_two_ranks_started_by_bench_itself():
    line = _bench("--gpus", "2", "--rehearse", "--batch", "16", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    mg = line["multi_gpu"]
    assert mg["backend"].startswith("gloo") and mg["graph_nodes"] >= 20 and mg["world_tag"] == 0
    assert mg["frames_through_sequential_update"] <= 2  # only the start-up frame(s) before a world tag exists


def test_configs4_window_solve_is_healthy():
    line = _bench("--workload", "configs4", "--batch", "16", "--steps", "2", "--warmup", "1", "--timed-only")
    lm = line["multi_gpu"]["lm_last"]
    assert lm is not None and lm["seeded"] and lm["tags"] >= 190 and lm["observations"] > 1000
    assert lm["cost"] <= lm["cost0"] and lm["cost"] / lm["observations"] < 5.0, lm
    assert line["multi_gpu"]["graph_nodes"] == 200
