"""Bounded runs of the randomised GPU-vs-oracle sweeps (tools/fuzz_parity.py, tools/fuzz_clutter.py) inside the GPU suite:
random and word / tile-boundary frame sizes, 1-9 tags, noise, decimate 1-3, ragged batches; and tag scenes under clutter
(rectangles, blocky noise, stripes, checkerboards, gradients) that fill the cluster table, the point pool and the dense
launches.  ids, hamming and margin exact, corners to 1e-9 px, PnP to 1e-6; the long runs are the tools themselves."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,count,seed", [("fuzz_parity.py", 300, 2025), ("fuzz_clutter.py", 300, 2025)])
def test_randomised_sweep(tool, count, seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(count), str(seed)], cwd=ROOT, capture_output=True, timeout=900)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and out.strip().splitlines()[-1].startswith("OK:"), out[-1500:] + p.stderr.decode(errors="replace")[-500:]
