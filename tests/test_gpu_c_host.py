"""-m gpu: the C ABI driven from plain C (examples/host.c, built with gcc against include/soccer_hip.h): the same run through the
Python layer must give the same counts to the unit — the boundary is the C ABI, the host language is the caller's choice."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from gym_soccer_littman94_amd import SoccerBatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("slip,seed,T", [(0.2, 7, 100), (0.0, 12345, 130)])
def test_c_host_matches_the_python_layer(tmp_path, slip, seed, T):
    exe = str(tmp_path / "host_c")
    lib_dir = os.path.join(ROOT, "gym_soccer_littman94_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "host.c"),
                           "-o", exe, "-L", lib_dir, "-lsoccer_hip", "-Wl,-rpath," + lib_dir])
    r = subprocess.run([exe, repr(slip), str(seed), str(T)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    m = re.search(r"lanes (\d+) steps (\d+) slip \S+ seed (\d+) tick (\d+) episodes (\d+) hist (\d+) (\d+) (\d+) last_sum (-?\d+) misuse 0", r.stdout)
    assert m, r.stdout
    n = int(m.group(1))
    b = SoccerBatch(n, 5, 4, slip, seed=seed, autoreset=True)
    rew = b.alloc((T, n), np.int8); term = b.alloc((T, n), np.uint8); trunc = b.alloc((T, n), np.uint8)
    last = b.alloc(n, np.int8); cnt = b.alloc(n, np.int32)
    b.reset()
    b.rollout(T, sample_actions=True, reward=rew, terminated=term, truncated=trunc, out_stride=n)
    hist = b.trajectory_returns(T, rew, term, trunc, n, last_return=last, episode_count=cnt)
    assert int(m.group(4)) == b.tick == 1 + T
    assert [int(m.group(k)) for k in (6, 7, 8)] == [int(x) for x in hist]
    assert int(m.group(5)) == int(cnt.download().sum()) and int(m.group(9)) == int(last.download().astype(np.int64).sum())
    assert int(hist.sum()) > n          # every lane finished at least one episode
    b.close()
