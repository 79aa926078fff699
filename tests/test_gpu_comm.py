"""-m gpu: the job's only exchange (BASELINE configs[3]) — per-lane episode returns from the result
trajectories (soccer_trajectory_returns), the RCCL communicator behind the C ABI (soccer_comm_*), and a two-rank rehearsal
of the sharded HIP path on ONE GPU through the host-file communicator (RCCL refuses two ranks on one device; the 8-GPU
curve is the driver's to measure)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from gym_soccer_littman94_amd import SoccerBatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _returns_numpy(rew, term, trunc):
    fin = (term | trunc) != 0
    T, n = rew.shape
    last = np.zeros(n, np.int8)
    for k in range(T):
        last[fin[k]] = rew[k][fin[k]]
    hist = np.array([int(((rew == v) & fin).sum()) for v in (-1, 0, 1)], np.uint64)
    return last, fin.sum(0).astype(np.int32), hist


@pytest.mark.parametrize("n,T,pad", [(4096, 230, 0), (4099, 300, 0), (1021, 77, 3), (4, 1, 0)])
def test_trajectory_returns_matches_numpy(n, T, pad):
    """real trajectories of a slip-0.2 rollout (goals, truncations, auto-resets); ragged lane counts and a row stride > n"""
    b = SoccerBatch(n, 5, 4, 0.2, seed=3, autoreset=True)
    b.reset()
    stride = n + pad
    rew = b.alloc((T, stride), np.int8).fill(0); term = b.alloc((T, stride), np.uint8).fill(0); trunc = b.alloc((T, stride), np.uint8).fill(0)
    b.rollout(T, sample_actions=True, reward=rew, terminated=term, truncated=trunc, out_stride=stride)
    last = b.alloc(n, np.int8).fill(0x55); cnt = b.alloc(n, np.int32).fill(0x55)
    hist = b.trajectory_returns(T, rew, term, trunc, stride, last_return=last, episode_count=cnt)
    R, TE, TR = (x.download()[:, :n] for x in (rew, term, trunc))
    e_last, e_cnt, e_hist = _returns_numpy(R, TE, TR)
    np.testing.assert_array_equal(last.download(), e_last)
    np.testing.assert_array_equal(cnt.download(), e_cnt)
    np.testing.assert_array_equal(hist, e_hist)
    np.testing.assert_array_equal(hist, b.stats()[0])            # the rollout's own episode histogram
    # outputs are optional; a second call starts from zero (the histogram is per call)
    np.testing.assert_array_equal(b.trajectory_returns(T, rew, term, trunc, stride), e_hist)
    with pytest.raises(AssertionError, match="stride"):
        b.trajectory_returns(T, rew, term, trunc, n - 1)
    b.close()


_RCCL_ONE_RANK = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np
from gym_soccer_littman94_amd import SoccerBatch
from gym_soccer_littman94_amd.comm import RcclComm
assert "torch" not in sys.modules
n = 1 << 20
b = SoccerBatch(n, 5, 4, 0.0, seed=1, autoreset=True)
c = RcclComm(b, 0, 1, timeout=60.0, directory=%(dir)r)
x = np.random.default_rng(0).integers(-1, 2, size=n).astype(np.int8)
send = b.alloc(n, np.int8).upload(x); recv = b.alloc(n, np.int8).fill(7)
c.all_gather_lanes(b, send, recv, n)
b.sync()
assert np.array_equal(recv.download(), x)
assert c.sum_u64([1, 2, 3]).tolist() == [1, 2, 3]
assert c.max_f64([0.25, -4.0]).tolist() == [0.25, -4.0]
assert c.gather_f64([1.5, 2.5]).tolist() == [[1.5, 2.5]]
c.barrier()
maps = open("/proc/self/maps").read()
rccl = sorted({l.split()[-1] for l in maps.splitlines() if "librccl" in l})
hip = sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l})
c.close(); b.close()
print(json.dumps({"rccl": rccl, "hip": hip}))
'''


def test_rccl_communicator_one_rank_through_the_c_abi(tmp_path):
    """soccer_comm_unique_id / _init / _all_gather / _sum_u64 / _max_f64 / _barrier with world = 1, in a torch-free process on the
    image's own ROCm runtime (what every bench.py rank is): librccl gets loaded, the calls go through, the bytes arrive."""
    env = dict(os.environ, SOCCER_HIP_RUNTIME="system")
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK % {"root": ROOT, "dir": str(tmp_path)}],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["rccl"], "librccl was not loaded"
    assert all("/torch/" not in p for p in info["hip"] + info["rccl"]), info


def test_torch_distributed_nccl_one_rank():
    """the torch.distributed helpers of distributed.py on backend "nccl" (= RCCL): 1-rank init + all_gather_into_tensor +
    all_reduce (tools/rccl_smoke.py, promoted to a test)"""
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_smoke.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "all_gather + all_reduce ok" in r.stdout


_SHARD_RANK = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np
from gym_soccer_littman94_amd import SoccerBatch
from gym_soccer_littman94_amd.comm import HostComm
from gym_soccer_littman94_amd.distributed import shard_range
rank, world, total, T = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), %(total)d, %(T)d
lo, hi = shard_range(total, rank, world); n = hi - lo
acts = np.random.default_rng(77).integers(0, 5, size=(T, 2, total), dtype=np.int8)[:, :, lo:hi]
b = SoccerBatch(n, 5, 4, 0.2, seed=9, autoreset=True, lane_offset=lo)
c = HostComm(rank, world, timeout=60.0, directory=%(dir)r)
A = b.alloc((T, 2, n), np.int8).upload(acts)
rew = b.alloc((T, n), np.int8); term = b.alloc((T, n), np.uint8); trunc = b.alloc((T, n), np.uint8); obs = b.alloc(n, np.uint16)
b.reset()
for k in range(T):
    b.step_plain(A.ptr + 2 * k * n, A.ptr + (2 * k + 1) * n, obs, rew.row(k), term.row(k), trunc.row(k))
last = b.alloc(n, np.int8)
hist = b.trajectory_returns(T, rew, term, trunc, n, last_return=last)
gathered = b.alloc(total, np.int8)
c.barrier()
c.all_gather_lanes(b, last, gathered, n)
hist = c.sum_u64(hist)
if rank == 0:
    np.savez(%(out)r, last=gathered.download(), hist=hist)
c.close(); b.close()
'''


def test_two_ranks_on_one_gpu_shard_the_hip_path(tmp_path):
    """Sharding invariance ON THE DEVICE: two processes, each with a handle over its half of the global lanes (lane_offset = its
    first global id), step the HIP kernels, reduce their trajectories and exchange through the host-file communicator; the
    gathered per-lane returns and the summed histogram must be those of ONE handle over all the lanes."""
    total, T, world = 8192, 130, 2
    out = str(tmp_path / "gathered.npz")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), SOCCER_HIP_RUNTIME="system")
        procs.append(subprocess.Popen([sys.executable, "-c", _SHARD_RANK % {"root": ROOT, "dir": str(tmp_path / "rdv"), "total": total, "T": T, "out": out}],
                                      env=env, stderr=subprocess.PIPE, text=True))
    os.makedirs(str(tmp_path / "rdv"), exist_ok=True)
    for p in procs:
        _, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-3000:]
    g = np.load(out)
    acts = np.random.default_rng(77).integers(0, 5, size=(T, 2, total), dtype=np.int8)
    b = SoccerBatch(total, 5, 4, 0.2, seed=9, autoreset=True)
    A = b.alloc((T, 2, total), np.int8).upload(acts)
    rew = b.alloc((T, total), np.int8); term = b.alloc((T, total), np.uint8); trunc = b.alloc((T, total), np.uint8); obs = b.alloc(total, np.uint16)
    b.reset()
    for k in range(T):
        b.step_plain(A.ptr + 2 * k * total, A.ptr + (2 * k + 1) * total, obs, rew.row(k), term.row(k), trunc.row(k))
    last = b.alloc(total, np.int8)
    hist = b.trajectory_returns(T, rew, term, trunc, total, last_return=last)
    np.testing.assert_array_equal(g["last"], last.download())
    np.testing.assert_array_equal(g["hist"], hist)
    assert hist.sum() > total
    b.close()
