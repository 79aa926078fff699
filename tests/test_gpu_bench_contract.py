"""-m gpu: bench.py's one JSON line keeps the driver's contract (fields, units, self-consistency) and the two-rank rehearsal
path (self-spawned ranks, host-file communicator, ranks sharing the one GPU in turn) produces a coherent line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*flags):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, *flags], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_keeps_the_contract():
    K, W, N = 6, 2, 1 << 18
    d = _run("--steps", str(K), "--warmup", str(W), "--lanes", str(N), "--cpu-seconds", "2", "--rollout", "10", "--no-vector-env")
    assert d["metric"].startswith("env-steps/sec") and d["unit"] == "env-steps/s" and d["higher_is_better"] is True
    assert (d["n_gpus"], d["steps"], d["warmup"], d["scaling"], d["vs_baseline"], d["dtype"], d["data"]) == (1, K, W, "weak", None, "int8", "synthetic")
    assert d["config"]["lanes_per_gpu"] == N and "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["host"]["torch_in_process"] is False and d["config"]["host"]["hip_runtime"] == "system"
    assert abs(d["value"] - N * K / (d["ms_per_step"] * 1e-3 * K)) / d["value"] < 1e-9          # value and ms_per_step are one clock
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - 19 * N / (d["ms_per_step"] * 1e-3) / 1e9) / r["achieved"] < 1e-9   # 19 algorithmic bytes per env-step
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1 and 0 < r["frac_device"] < 1
    assert r["regime"] in ("infinity-cache-resident", "hbm-streaming") and r["working_set_bytes"] == 12 * N + 7 * N * K
    assert "traffic" in r and r["algorithmic_bytes_per_launch"] == 19 * N
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert sum(d["episodes"]["hist_minus1_0_plus1"]) >= 0 and d["episodes"]["gathered_last_returns"] == N
    assert d["fused_rollout"]["steps_fused"] == 10 and d["selfplay_rollout_config5"]["env_steps_per_s"] > 0


def test_odd_step_count_is_one_captured_sequence():
    d = _run("--steps", "5", "--warmup", "1", "--lanes", str(1 << 16), "--no-cpu-baseline", "--rollout", "0", "--no-vector-env")
    assert d["steps"] == 5 and d["untimed"]["steps_per_replay"] == 5 and "cpu_baseline" not in d


def test_two_rank_rehearsal_on_one_gpu():
    N, K = 1 << 16, 8
    d = _run("--gpus", "2", "--comm", "host", "--steps", str(K), "--warmup", "2", "--lanes", str(N), "--no-cpu-baseline", "--rollout", "0")
    assert d["n_gpus"] == 2 and d["config"]["global_lanes"] == 2 * N and d["config"]["host"]["comm"] == "host-files"
    assert len(d["per_rank"]) == 2 and {x["rank"] for x in d["per_rank"]} == {0, 1}
    assert abs(d["ms_per_step"] * 1e3 * K - max(x["wall_us"] for x in d["per_rank"])) < 1e-6       # the job's time is the slowest rank's
    assert d["episodes"]["gathered_last_returns"] == 2 * N and "REHEARSAL" in d["timed_region"]


def test_two_real_ranks_when_the_box_has_two_gpus_else_a_loud_refusal():
    """`python bench.py --gpus 2` with the real communicator (RCCL through soccer_comm_*).  On a box with >= 2 GPUs this IS the
    two-rank RCCL run (shards in global lane order, per-rank clocks, no collective in the region); on a one-GPU box rank 1 must
    refuse before any collective can hang, and the parent must stop rank 0 (which is waiting for it in the RCCL bring-up)."""
    import ctypes
    import time
    from gym_soccer_littman94_amd import _lib
    n = ctypes.c_int()
    assert _lib.load().soccer_device_count(ctypes.byref(n)) == 0
    N, K = 1 << 16, 8
    flags = ["--gpus", "2", "--steps", str(K), "--warmup", "2", "--lanes", str(N), "--no-cpu-baseline", "--rollout", "0", "--rank-deadline", "120"]
    if n.value >= 2:
        d = _run(*flags)
        assert d["n_gpus"] == 2 and d["config"]["host"]["comm"] == "rccl" and len(d["per_rank"]) == 2
        assert d["episodes"]["gathered_last_returns"] == 2 * N and "REHEARSAL" not in d["timed_region"]
        return
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, *flags], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "only 1 are visible" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert time.monotonic() - t0 < 100
