"""bench.py must be able to bring up its own ranks: the driver starts it as `python bench.py --gpus N`.

CPU-only rehearsal of the launcher (no GPU work, no measurement): the parent spawns N fresh children with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set before anything could touch a GPU, the ranks rendezvous through the
host-file communicator (gym_soccer_littman94_amd/comm.py), rank 0 prints the single JSON line, and a failing rank makes
the whole run fail.  Also: the shape of the timed region (no collective between the clock's start and stop) and the
torch-freeness of the rank processes."""
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


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_no_collective_inside_the_timed_region():
    """the N > 1 region used to close on a dist.barrier().  The region is ONE function used at every N; with spies
    for everything it calls, the job barrier must come before the clock starts and nothing but this rank's own launch and
    device synchronisation may happen between t0 and the clock's stop."""
    bench = _load_bench()
    log = []
    clock = iter(range(100))

    def now():
        log.append("clock"); return float(next(clock))
    t0, t1, t2 = bench.timed_region(lambda: log.append("job_barrier"), lambda: log.append("device_sync"),
                                    lambda: log.append("launch"), now)
    assert log == ["job_barrier", "device_sync", "clock", "launch", "clock", "device_sync", "clock"]
    assert (t0, t1, t2) == (0.0, 1.0, 2.0)
    log.clear()
    bench.timed_region(None, lambda: log.append("device_sync"), lambda: log.append("launch"), now)      # N = 1: the same region
    assert log == ["device_sync", "clock", "launch", "clock", "device_sync", "clock"]
    # and main() hands it the communicator's barrier and nothing else: no comm call appears between the call and the clocks' use
    src = open(BENCH).read()
    body = src[src.index("def measure(job_barrier):"):src.index("per_rank = None")]
    assert body.count("timed_region(") == 1 and "timed_region(job_barrier, device_sync, launch, time.perf_counter)" in body
    inside = body[:body.index("# A rehearsal with more ranks than GPUs")]
    assert "comm" not in inside, "a communicator call inside measure(): between the clock's start and the per-rank clocks"
    launch_body = src[src.index("def launch():"):src.index("def measure(job_barrier):")]
    assert "comm" not in launch_body and "barrier" not in launch_body


def test_rank_processes_are_torch_free():
    """torch (and with it PyTorch's bundled, older HIP runtime) is imported in ONE function: the vector-env child leg."""
    src = open(BENCH).read()
    leg = src[src.index("def vector_env_leg"):src.index("def run_vector_env_child")]
    rest = src.replace(leg, "")
    assert "import torch" in leg and "import torch" not in rest and "init_process_group" not in rest and "dist." not in rest
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import gym_soccer_littman94_amd, gym_soccer_littman94_amd.comm; "
            "assert 'torch' not in sys.modules" % ROOT)
    subprocess.check_call([sys.executable, "-c", code])


def test_gpu_local_cpus_never_raises_and_stays_inside_the_affinity_mask():
    bench = _load_bench()
    cpus, note = bench.gpu_local_cpus(0, 1)
    assert isinstance(note, str) and (cpus is None or set(cpus) <= set(os.sched_getaffinity(0)))
    assert bench._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def _host_comm_worker(rank, world, directory, q):
    sys.path.insert(0, ROOT)
    import numpy as np
    from gym_soccer_littman94_amd.comm import HostComm
    c = HostComm(rank, world, timeout=30.0, directory=directory)
    c.barrier()
    s = c.sum_u64([rank + 1, 10 * (rank + 1), 7])
    m = c.max_f64([float(rank), -float(rank)])
    g = c.gather_f64([100.0 + rank, 0.5 * rank])
    for _ in range(20):                       # many exchanges: the files of finished exchanges are cleaned up as it goes
        c.barrier()
    left = len(os.listdir(directory))
    c.close()
    q.put((rank, s.tolist(), m.tolist(), g.tolist(), left))


def test_host_file_communicator_three_ranks(tmp_path):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    ps = [ctx.Process(target=_host_comm_worker, args=(r, world, str(tmp_path / "rdv"), q)) for r in range(world)]
    os.makedirs(str(tmp_path / "rdv"))
    for p in ps: p.start()
    got = sorted(q.get(timeout=60) for _ in range(world))
    for p in ps: p.join(30)
    for rank, s, m, g, left in got:
        assert s == [6, 60, 21] and m == [2.0, 0.0]
        assert g == [[100.0, 0.0], [101.0, 0.5], [102.0, 1.0]]
        assert left <= 2 * world


def test_under_torch_distributed_run_two_ranks_rendezvous():
    """The driver's N > 1 command line: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`.  The ranks are
    children of the elastic agent (same parent, same MASTER_PORT), which is all the host-file / RCCL-id rendezvous needs."""
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--launcher-selftest"],
                       capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["rank_sum"] == 3.0


class _FakeLib:
    """Stands in for libsoccer_hip's soccer_comm_unique_id in the CPU test of RcclComm's bring-up."""
    def soccer_comm_unique_id(self, buf):
        for i in range(128):
            buf[i] = (i * 7 + 3) & 0xff
        return 0

    def soccer_last_error(self, h):
        return b""


class _FakeBatch:
    def __init__(self, log, collective):
        self.lib, self.h, self.log, self.collective = _FakeLib(), None, log, collective

    def comm_init(self, world, rank, uid):
        self.log.append(("init", world, rank, bytes(uid)))
        self.collective.barrier()              # ncclCommInitRank is collective: it returns when every rank has joined

    def comm_barrier(self):
        self.log.append(("barrier",))

    def comm_destroy(self):
        self.log.append(("destroy",))


def _rccl_bringup_worker(rank, world, directory, q):
    sys.path.insert(0, ROOT)
    import time
    from gym_soccer_littman94_amd.comm import HostComm, RcclComm
    if rank == 0:
        time.sleep(0.5)                        # the other ranks are already polling for the id when rank 0 publishes it
    log = []
    c = RcclComm(_FakeBatch(log, HostComm(rank, world, timeout=30.0, directory=directory + "_collective")), rank, world,
                 timeout=30.0, directory=directory)
    c.close()
    q.put((rank, log))


def test_rccl_bring_up_hands_every_rank_the_same_unique_id(tmp_path):
    """The part of the N > 1 path no single GPU can run: ranks != 0 learn rank 0's 128-byte id through the rendezvous file,
    ignore a stale file of an earlier job, and every rank calls comm_init(world, rank, id) then the first barrier.  (RCCL itself is
    replaced by a recording fake; the real library runs in tests/test_gpu_comm.py with world = 1.)"""
    import multiprocessing as mp
    import time
    d = tmp_path / "rdv"; d.mkdir(); (tmp_path / "rdv_collective").mkdir()
    stale = d / "rccl_unique_id"; stale.write_bytes(b"\x00" * 128)
    old = time.time() - 3600
    os.utime(str(stale), (old, old))           # an hour old: a job that died before cleaning up
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    ps = [ctx.Process(target=_rccl_bringup_worker, args=(r, world, str(d), q)) for r in range(world)]
    for p in ps: p.start()
    got = dict(q.get(timeout=60) for _ in range(world))
    for p in ps: p.join(30)
    uid = bytes((i * 7 + 3) & 0xff for i in range(128))
    for r in range(world):
        assert got[r][0] == ("init", world, r, uid), got[r][:1]
        assert got[r][1] == ("barrier",) and got[r][-1] == ("destroy",)


def test_rank_placement_on_a_two_socket_eight_gpu_topology(tmp_path):
    """gpu_local_cpus against a made-up sysfs tree of the node the driver would use for the scaling run: two CPU nodes (KFD
    nodes 0-1, no SIMDs), eight GPUs (KFD nodes 2-9) of which 0-3 hang off NUMA node 0 and 4-7 off node 1, 32 CPUs per node with
    hyper-thread siblings listed as a second range.  Every rank must land on its GPU's node, in a slice no other rank has."""
    bench = _load_bench()
    root = tmp_path / "sys"
    for node in range(10):
        d = root / "class/kfd/kfd/topology/nodes" / str(node); d.mkdir(parents=True)
        gpu = node - 2
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\ndrm_render_minor %d\n"
                                      % ((16, 0, -1) if node < 2 else (0, 1024, 128 + gpu)))
        if node >= 2:
            r = root / ("class/drm/renderD%d/device" % (128 + gpu)); r.mkdir(parents=True)
            (r / "numa_node").write_text("%d\n" % (0 if gpu < 4 else 1))
    for nn, cl in ((0, "0-15,32-47"), (1, "16-31,48-63")):
        d = root / ("devices/system/node/node%d" % nn); d.mkdir(parents=True)
        (d / "cpulist").write_text(cl + "\n")
    everything = set(range(64))
    seen = {}
    for r in range(8):
        cpus, note = bench.gpu_local_cpus(r, 8, sysfs=str(root), allowed=everything)
        assert cpus and len(cpus) == 8, note
        node_cpus = set(range(0, 16)) | set(range(32, 48)) if r < 4 else set(range(16, 32)) | set(range(48, 64))
        assert set(cpus) <= node_cpus, (r, cpus, note)
        for c in cpus:
            assert c not in seen, "ranks %d and %d share CPU %d" % (seen[c], r, c)
            seen[c] = r
    assert len(seen) == 64
    # one rank on an eight-GPU node: the whole node of GPU 0; an affinity mask that leaves a rank one CPU: not pinned, and said so
    cpus, _ = bench.gpu_local_cpus(0, 1, sysfs=str(root), allowed=everything)
    assert sorted(cpus) == sorted(set(range(0, 16)) | set(range(32, 48)))
    cpus, note = bench.gpu_local_cpus(5, 8, sysfs=str(root), allowed={16, 17, 18})
    assert cpus is None and "not pinned" in note
    # a launcher that hides all GPUs but one from every rank: the rank's GPU is the one it sees (its node, unsliced)
    os.environ["HIP_VISIBLE_DEVICES"] = "6"
    try:
        cpus, note = bench.gpu_local_cpus(6, 8, sysfs=str(root), allowed=everything)
    finally:
        del os.environ["HIP_VISIBLE_DEVICES"]
    assert sorted(cpus) == sorted(set(range(16, 32)) | set(range(48, 64))) and "unsliced" in note
    # a GPU node this process may not read (another tenant's): skipped, the visible GPUs keep HIP's numbering
    os.chmod(str(root / "class/kfd/kfd/topology/nodes/2/properties"), 0)
    if os.geteuid() != 0:                        # (root reads anything)
        cpus, note = bench.gpu_local_cpus(3, 7, sysfs=str(root), allowed=everything)
        assert set(cpus) <= set(range(16, 32)) | set(range(48, 64)), note          # visible GPU 3 = the host's GPU 4: NUMA node 1
