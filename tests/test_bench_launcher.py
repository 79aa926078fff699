"""bench.py must be able to bring up its own ranks: the driver starts it as `python bench.py --gpus N`.

CPU-only rehearsal of the launcher (no GPU work, no measurement): the parent spawns N fresh children with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set before anything could touch a GPU, the ranks rendezvous over
gloo on 127.0.0.1, rank 0 prints the single JSON line, and a failing rank makes the whole run fail."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*flags, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *flags], capture_output=True, text=True, timeout=240, env=e)


def test_self_spawned_ranks_print_one_json_line():
    r = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--launcher-selftest")
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["launcher_selftest"] is True and d["n_gpus"] == 2 and d["rank_sum"] == 3.0
    assert d["env"] == {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1"}


def test_a_failing_rank_fails_the_run():
    r = _run("--gpus", "2", "--launcher-selftest", "--selftest-fail-rank", "1")
    assert r.returncode != 0
    assert "rank 1 exited with code 3" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


def test_a_rank_that_hangs_is_stopped_at_the_deadline():
    """One rank never reaches the rendezvous: the others block in it.  The parent's overall deadline stops every child
    it started and the run fails — within the deadline plus the termination grace, not after the driver's whole budget."""
    import time
    t0 = time.monotonic()
    r = _run("--gpus", "2", "--launcher-selftest", "--selftest-hang-rank", "1", "--rank-deadline", "8", "--collective-timeout", "60")
    took = time.monotonic() - t0
    assert r.returncode == 124, (r.returncode, r.stderr)
    assert "still running after the 8 s deadline" in r.stderr
    assert took < 40, took
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


def test_a_stuck_peer_fails_the_collective_timeout_of_the_others():
    """Without the parent (ranks started by a launcher): the process group's own timeout turns a missing peer into an
    error of the waiting rank instead of an endless rendezvous."""
    r = _run("--gpus", "2", "--launcher-selftest", "--collective-timeout", "5",
             env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29513"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


def test_under_a_launcher_it_does_not_spawn_again():
    # what torch.distributed.run sets for a 1-rank job: the process is the rank, no children
    r = _run("--gpus", "1", "--launcher-selftest",
             env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29511"})
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_world_size_mismatch_is_refused_before_any_gpu_call():
    r = _run("--gpus", "4", "--launcher-selftest",
             env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29512"})
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr


def test_parent_does_not_import_torch_or_load_hip():
    # the spawning parent must stay GPU-clean: its code path may not import torch or the HIP library
    src = open(BENCH).read()
    body = src[src.index("def spawn_ranks"):src.index("def launcher_selftest")]
    assert "import torch" not in body and "_lib" not in body and "SoccerBatch" not in body
    head = src[:src.index("def cpu_baseline")]
    assert "import torch" not in head and "gym_soccer_littman94_amd" not in head
