"""-m gpu: a 45-second slice of tools/soak.py — random pitch / slip / seed / lane count / lane offset / auto-reset / max_steps,
single steps (lean, full, the gym outputs aligned and not), masked resets, fused rollouts (streams, sampled, mixed policies,
single-agent), every lane of every step against the oracle.  The long soaks quoted in profiles/ are builder-run; this slice is
what the driver's own GPU test run observes.  The seed is fixed: a failure prints a configuration that can be re-run."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_soak_slice():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "45", "20261005"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    m = re.search(r"soak ok: (\d+) random configurations, ([0-9.e+]+) lane-steps", r.stdout)
    assert m, r.stdout[-2000:]
    assert int(m.group(1)) >= 50 and float(m.group(2)) >= 5e6, r.stdout[-500:]


def test_one_handle_above_the_2_30_lane_launch_limit_for_real():
    """tools/huge_handle_check.py: 2^30 + 2^20 + 4 lanes on one handle (every batched_* call is two launches; ~25 GB of HBM, ~2 s):
    reset, steps, a fused rollout and the trajectory reduction equal small handles that own the first lanes, the lanes around the
    2^30 boundary and the last lanes.  (tests/test_gpu_swar.py exercises the same split at 4 096 lanes per launch through
    SOCCER_SWAR_LAUNCH_LANES; this is the real size.)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "huge_handle_check.py"), "0.2"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "huge handle ok: 1074790404 lanes" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
