#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched Littman-94 soccer step on MI355X.

Workload (BASELINE.json metric / configs[2]): 1 048 576 environments per
GPU, uniform-random joint actions resident in HBM, auto-reset on terminal/truncation, int8 SoA
state.  One "step" = one batched_step launch over the whole batch through the C ABI
(libsoccer_hip.so).  With --gpus N each rank owns its own contiguous shard of N x 2^20 global
lanes (weak scaling; no collective on the data path; one RCCL all-gather of the int8 per-lane
episode returns after the timed region — BASELINE configs[3]).

The rank processes are torch-free: actions come from numpy, buffers from soccer_malloc, the clock closes on
soccer_sync, the statistics come from soccer_trajectory_returns and the exchange goes through soccer_comm_* (RCCL
resolved by the library itself) — so every rank, at N = 1 and at N = 8 alike, runs on the image's own ROCm runtime.
PyTorch appears in one place only: the `vector_env_device` leg (VectorSoccerEnv(io="device") hands out torch tensors),
which runs in a child process of its own, started and finished before this process touches the GPU.

Prints ONE JSON line (rank 0).  `roofline.achieved` = 19 algorithmic bytes per env-step (SURVEY.md
§8(d): read 6 B state + 2 B actions, write 6 B state + 5 B obs/reward/terminated/truncated) x lanes
per launch / ms_per_step, the host wall clock `value` is computed from; `roofline.achieved_device` /
`frac_device` use device clock stamps captured around the K launches on the kernel's own stream instead.
`cpu_baseline` = the CPU oracle (oracle/soccer_oracle.c, a port pinned bit-for-bit to the reference
by tests/golden) timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 19
HBM_PEAK_GBPS = 8000.0                # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
INFINITY_CACHE_BYTES = 256 << 20      # same guide: 256 MB Infinity Cache in front of HBM


def cpu_baseline(lanes_total, slip, seconds):
    """The oracle on the host cores: one Oracle per thread, each over its own lane shard."""
    from oracle.oracle import Oracle
    cores = len(os.sched_getaffinity(0))
    threads = max(1, min(cores, 64))
    per = 16384
    rng = np.random.default_rng(2024)
    acts = rng.integers(0, 5, size=(64, 2, per), dtype=np.int8)
    oracles = [Oracle(5, 4, slip, n=per, seed=0, lane_offset=i * per, autoreset=True) for i in range(threads)]
    for o in oracles:
        o.reset()
    counts = [0] * threads
    deadline = [0.0]

    def work(i):
        o, k = oracles[i], 0
        while time.perf_counter() < deadline[0]:
            o.step(acts[k % 64, 0], acts[k % 64, 1]); k += 1
        counts[i] = k
    # one core first ("on 1 core and on all cores"), a quarter of the budget; then all threads
    deadline[0] = time.perf_counter() + seconds * 0.25
    t0 = time.perf_counter(); work(0); dt1 = time.perf_counter() - t0
    one_core = counts[0] * per / dt1
    deadline[0] = time.perf_counter() + seconds * 0.75
    t0 = time.perf_counter()
    ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    steps = sum(counts) * per
    return {"value": steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port", "value_1_core": one_core,
            "sample": "%d threads x %d lanes x ~%d steps of the same uniform-random workload (%.1f s), after 1 thread alone for %.1f s"
                      % (threads, per, counts[0], dt, dt1)}


REFERENCE_PYTHON = {"value": 5.3e4, "unit": "env-steps/s", "cores": 1, "where": "build container (Xeon 2.1 GHz)",
                    "what": "the reference's own SoccerSimultaneousEnv.step loop, BASELINE config 1 (slip 0; 4.7e4 at slip 0.2)",
                    "source": "BASELINE.md section 2 (the reference's Python cannot travel to the GPU box)"}


def csrc_sha256():
    """Fingerprint of the kernel sources: profiles/traffic.json records the one its PMC run was built from."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gym_soccer_littman94_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", "Makefile")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


# ---- rank placement: sysfs only, nothing here touches the GPU -----------------------------------------------------------
def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_local_cpus(local_rank, local_world, sysfs="/sys", allowed=None):
    """CPUs this rank should run on: those of the NUMA node its GPU hangs off, cut into disjoint slices for the ranks that
    share the node (a spinning host thread per rank: 15 % of a ~100 us timed region is host time).  Read from sysfs — the KFD
    topology lists the GPUs in the order HIP numbers them — so it can run before the process makes its first GPU call.
    Returns (cpus or None, note)."""
    try:
        topo = sysfs + "/class/kfd/kfd/topology/nodes"
        gpus = []
        for node in sorted(os.listdir(topo), key=int):
            try:
                text = open(os.path.join(topo, node, "properties")).read()
            except OSError:
                continue                       # a GPU of the host that this container was not given: HIP does not number it either
            props = dict(l.split(None, 1) for l in text.splitlines() if " " in l)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(int(props.get("drm_render_minor", "-1")))
        narrowed = False
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            v = os.environ.get(var)
            if v and all(x.strip().isdigit() for x in v.split(",")):
                gpus = [gpus[int(x)] for x in v.split(",") if int(x) < len(gpus)]; narrowed = True
        if not gpus:
            return None, "no GPU in the KFD topology"

        def numa_of(minor):
            n = int(open(sysfs + "/class/drm/renderD%d/device/numa_node" % minor).read())
            return max(n, 0)                   # -1: a single-node machine
        allowed = set(os.sched_getaffinity(0)) if allowed is None else set(allowed)
        if len(gpus) == 1 and local_world > 1 and narrowed:
            # every rank isolated behind its own *_VISIBLE_DEVICES: this rank's GPU is the one it sees; who else shares the
            # node is not knowable from here, so the whole node it is
            node = numa_of(gpus[0])
            cpus = [c for c in _cpulist(open(sysfs + "/devices/system/node/node%d/cpulist" % node).read()) if c in allowed]
            return (cpus or None), "NUMA node %d of the one visible GPU (render minor %d), unsliced" % (node, gpus[0])
        nodes = [numa_of(m) for m in gpus[:local_world]] if local_world <= len(gpus) else None
        if nodes is None or local_rank >= len(nodes):
            return None, "more local ranks than GPUs (rehearsal): not pinned"
        mine = nodes[local_rank]
        cpus = [c for c in _cpulist(open(sysfs + "/devices/system/node/node%d/cpulist" % mine).read()) if c in allowed]
        sharers = [r for r in range(len(nodes)) if nodes[r] == mine]
        per = len(cpus) // len(sharers)
        if per < 2:
            return None, "NUMA node %d has %d usable CPUs for %d ranks: not pinned" % (mine, len(cpus), len(sharers))
        k = sharers.index(local_rank)
        return cpus[k * per:(k + 1) * per], "NUMA node %d of GPU %d (render minor %d), slice %d of %d" % (mine, local_rank, gpus[local_rank], k, len(sharers))
    except Exception as e:                     # containers without the topology: run unpinned, and say so
        return None, "topology not readable (%s): not pinned" % type(e).__name__


def spawn_ranks(n, argv, deadline_s):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves.

    Runs in a parent process that has NOT touched the GPU (no torch import, no libsoccer_hip load): each rank
    is a fresh child `python bench.py <same flags>` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, exactly
    what torch.distributed.run would set (each child pins itself to its GPU's cores first thing, before any GPU call).
    Children inherit stdout, so rank 0's JSON line is the only one.
    Returns the exit code: 0 only if every rank exited 0; the first failure ends the others, and so does the overall
    deadline (a rank stuck in rendezvous or in a collective while the others wait for it: exit code 124).  Only the
    children started here are ever signalled, and nothing is re-executed."""
    import socket
    import subprocess
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    t_end = time.monotonic() + deadline_s
    rc, live = 0, set(range(n))

    def stop_others(grace=5.0):
        for q in live:
            procs[q].terminate()              # exactly the children started above
        t_kill = time.monotonic() + grace
        for q in live:
            try:
                procs[q].wait(timeout=max(0.0, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                procs[q].kill(); procs[q].wait()

    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print("bench.py: rank %d exited with code %d; stopping the other ranks" % (r, code), file=sys.stderr)
                stop_others(); live.clear()
                break
        if live and time.monotonic() > t_end:
            print("bench.py: ranks %s still running after the %.0f s deadline (--rank-deadline); stopping them"
                  % (sorted(live), deadline_s), file=sys.stderr)
            stop_others(); live.clear()
            rc = rc or 124
        time.sleep(0.05)
    return rc


def launcher_selftest(args, rank, world):
    """CPU-only rehearsal of the multi-rank wiring (tests/test_bench_launcher.py): rendezvous through the host-file
    communicator, one reduction, one JSON line from rank 0.  No GPU work and no measurement — never a bench result."""
    from gym_soccer_littman94_amd.comm import HostComm
    if args.selftest_fail_rank == rank:
        raise SystemExit(3)
    if args.selftest_hang_rank == rank:
        time.sleep(3600)                      # a rank that never reaches the rendezvous
    total = float(rank + 1)
    if world > 1:
        comm = HostComm(rank, world, timeout=args.collective_timeout)
        comm.barrier()
        total = float(comm.sum_u64([rank + 1])[0])
        clocks = comm.gather_f64([100.0 + rank])
        assert [float(x) for x in clocks[:, 0]] == [100.0 + r for r in range(world)]
        comm.close()
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "rank_sum": total,
                          "env": {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR")}}))


def timed_region(job_barrier, device_sync, launch, now):
    """THE timed region, the same code at every N.  The job's barrier and a device synchronisation come BEFORE the clock
    starts; the clock stops on this rank's OWN device synchronisation — no collective and no other rank inside the region
    (the job's time is the maximum over the ranks' clocks, gathered afterwards).  Returns (t0, t_launched, t_done).
    tests/test_bench_launcher.py holds this order."""
    if job_barrier is not None:
        job_barrier()
    device_sync()
    t0 = now()
    launch()
    t1 = now()
    device_sync()
    t2 = now()
    return t0, t1, t2


def vector_env_leg(args):
    """Child process (the only place PyTorch is imported): the gym-style surface north_star names,
    VectorSoccerEnv(io="device") — step() = one Python call = one ctypes call = one launch of the FULL kernel
    (final_obs + prob_code + both agents' float32 rewards + terminated|truncated + episode histogram on top of the four result
    streams: 31 B per env-step; lean: 23 B), and rollout(T) = the fused path behind the same API.  Prints one JSON object."""
    import torch
    from gym_soccer_littman94_amd import VectorSoccerEnv
    N, K = args.lanes, args.steps
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(2024)
    KV = max(200, min(K, 1000))
    KA = KV
    acts = torch.randint(0, 5, (KA, 2, N), dtype=torch.int8, device=dev, generator=g)
    stream_actions = KA * 2 * N > (96 << 20) if args.action_loads == "auto" else args.action_loads == "nt"
    out = {}
    for key, info, nbytes, note in (
            ("full", True, 31, "info=True: + final_observation, prob_code (info[agent]['p'] computed on access), int8 reward, episode histogram"),
            ("lean", False, 23, "info=False: observations, player_a's float32 reward (player_b's = its negation, on access), terminated, truncated, _final_observation")):
        v = VectorSoccerEnv(N, slip_prob=args.slip, seed=0, io="device", device=0, info=info, stream_actions=stream_actions)
        v.reset()
        pairs = [{"player_a": acts[k, 0], "player_b": acts[k, 1]} for k in range(KV)]
        for k in range(20):
            v.step(pairs[k])
        torch.cuda.synchronize()
        tv = time.perf_counter()
        for k in range(KV):
            v.step(pairs[k])
        torch.cuda.synchronize()
        dv = time.perf_counter() - tv
        o_, r_, te_, tr_, inf_ = v.step(pairs[0])
        assert r_["player_a"].dtype == torch.float32 and bool((r_["player_a"] == -r_["player_b"]).all())
        assert bool((inf_["_final_observation"] == (te_["player_a"] | tr_["player_a"])).all())
        assert v.batch.misuse() == 0
        out[key] = {"api": "VectorSoccerEnv(io='device', info=%s).step(dict of int8 CUDA tensors)" % info, "steps": KV,
                    "us_per_step": dv / KV * 1e6, "env_steps_per_s": N * KV / dv, "bytes_per_env_step": nbytes, "note": note}
        if key == "full" and args.rollout > 0:
            # the fused path behind the same API: rollout(T) == T step() calls (tests/test_gpu_vector_rollout.py) with the state in
            # registers: [T, N] trajectories of obs / int8 reward / terminated / truncated + the last step's infos (the default,
            # infos="last": T - 1 fused steps + one full step), or every step's final_obs / prob_code as well (infos="all": one launch)
            T = args.rollout
            ra = {"player_a": acts[:T, 0].contiguous(), "player_b": acts[:T, 1].contiguous()}
            v.rollout(T, ra); torch.cuda.synchronize()
            reps = []
            for _ in range(5):
                t0 = time.perf_counter(); O, R, TE, TR, I = v.rollout(T, ra); torch.cuda.synchronize(); reps.append(time.perf_counter() - t0)
            dt = sorted(reps)[2]
            assert O["player_a"].shape == (T, N) and int(I["reward_int8"].abs().max()) <= 1 and v.batch.misuse() == 0
            bpe = 7 + (12.0 + 24.0) / T                 # 2 B actions in, 5 B out per step; state + the last step's extras amortised over T
            out["rollout"] = {"api": "VectorSoccerEnv(io='device').rollout(T=%d, dict of [T, N] int8 CUDA tensors)" % T, "steps_fused": T,
                              "ms_per_rollout": dt * 1e3, "env_steps_per_s": N * T / dt, "bytes_per_env_step": bpe,
                              "frac_of_hbm_peak": bpe * N * T / dt / 1e9 / HBM_PEAK_GBPS,
                              "note": "wall clock around the call + torch.cuda.synchronize(), median of 5; infos='last' (the default): T - 1 steps by "
                                      "batched_rollout, the last by batched_step_ex (the T-th step's infos)"}
            for mode, key2, b2 in (("all", "rollout_infos_all", 10 + 12.0 / T), ("none", "rollout_infos_none", 7 + 12.0 / T)):
                v.rollout(T, ra, infos=mode); torch.cuda.synchronize()
                reps = []
                for _ in range(5):
                    t0 = time.perf_counter(); v.rollout(T, ra, infos=mode); torch.cuda.synchronize(); reps.append(time.perf_counter() - t0)
                out[key2] = {"api": "VectorSoccerEnv(io='device').rollout(T=%d, ..., infos='%s')" % (T, mode), "steps_fused": T,
                             "ms_per_rollout": sorted(reps)[2] * 1e3, "env_steps_per_s": N * T / sorted(reps)[2], "bytes_per_env_step": b2,
                             "note": "one batched_rollout_ex launch; 'all' = + every step's final_observation and prob_code trajectories"}
        v.close()
    print(json.dumps(out))


def run_vector_env_child(argv, timeout_s):
    """Start the vector-env leg as a child BEFORE this process touches the GPU and wait for it; returns its dict or a note."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    env.pop("SOCCER_HIP_RUNTIME", None)       # (the child shares torch's HIP runtime: libsoccer_hip's default when torch is installed)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", "vector-env"] + list(argv),
                           capture_output=True, text=True, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired:
        return {"skipped": "the vector-env child did not finish within %.0f s" % timeout_s}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"skipped": "the vector-env child failed (exit %d): %s" % (r.returncode, r.stderr.strip().splitlines()[-1:] or "")}
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--lanes", type=int, default=1 << 20, help="environments per GPU")
    ap.add_argument("--slip", type=float, default=0.0)
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph")
    ap.add_argument("--envs-per-thread", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rollout", type=int, default=100, help="also time a fused T-step rollout (0 = skip)")
    ap.add_argument("--comm", choices=["rccl", "host"], default="rccl",
                    help="rccl: the real path (RCCL over xGMI through soccer_comm_*); host: rehearse the multi-rank path on fewer GPUs "
                         "than ranks (ranks share devices, the exchange goes through files in host memory) — never a measured path")
    ap.add_argument("--no-vector-env", action="store_true", help="skip the VectorSoccerEnv(io='device') timing (a torch child process)")
    ap.add_argument("--action-loads", choices=["auto", "nt", "plain"], default="auto",
                    help="how batched_step reads its action streams: nt = SOCCER_F_STREAM_ACTIONS, plain = the library default, "
                         "auto = nt when the synthetic action trajectory exceeds 96 MB")
    ap.add_argument("--hip-runtime", choices=["system", "torch"], default="system",
                    help="system: the image's ROCm runtime (/opt/rocm); torch: the older runtime bundled with PyTorch (what a "
                         "process that imports torch gets; A/B runs)")
    ap.add_argument("--no-pin", action="store_true", help="do not pin the rank to its GPU's NUMA-local cores")
    ap.add_argument("--isolated-devices", action="store_true",
                    help="every rank has been given its own *_VISIBLE_DEVICES by the launcher and sees exactly one GPU: use device 0")
    ap.add_argument("--leg", choices=["main", "vector-env"], default="main", help=argparse.SUPPRESS)
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--selftest-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--selftest-hang-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--rank-deadline", type=float, default=300.0,
                    help="self-spawned ranks (python bench.py --gpus N): seconds after which ranks that are still running are "
                         "stopped and the run fails with exit code 124")
    ap.add_argument("--collective-timeout", type=float, default=120.0,
                    help="timeout of the rendezvous (and of every exchange of the host-file communicator), seconds")
    args = ap.parse_args()
    if args.leg == "vector-env":
        return vector_env_leg(args)

    # ---- rank bring-up: before anything touches the GPU -------------------------------------------
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], args.rank_deadline))         # parent: no torch, no HIP
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; start it as `python bench.py --gpus N` or with "
                         "torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    if args.launcher_selftest:
        return launcher_selftest(args, rank, world)
    pinned, pin_note = (None, "--no-pin") if args.no_pin else gpu_local_cpus(local_rank, local_world)
    if pinned:
        os.sched_setaffinity(0, pinned)      # this process, in place: nothing is re-executed

    # the gym-style surface (a torch process of its own), run to completion before this process makes its first GPU call
    vec_env = None
    if not args.no_vector_env and world == 1:
        child = ["--lanes", str(args.lanes), "--steps", str(args.steps), "--slip", str(args.slip), "--rollout", str(args.rollout),
                 "--action-loads", args.action_loads]
        vec_env = run_vector_env_child(child, 600.0)

    if args.hip_runtime == "system":
        os.environ["SOCCER_HIP_RUNTIME"] = "system"      # _lib.py: do not pre-load PyTorch's bundled HIP runtime
    from gym_soccer_littman94_amd import SoccerBatch, _lib
    from gym_soccer_littman94_amd.comm import HostComm, RcclComm
    from gym_soccer_littman94_amd.distributed import shard_range
    import ctypes
    lib = _lib.load()
    ndev = ctypes.c_int()
    if lib.soccer_device_count(ctypes.byref(ndev)) != 0 or ndev.value < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path to bench)")
    host_comm = args.comm == "host"
    # --isolated-devices: a launcher that hides all GPUs but one from every rank (its own *_VISIBLE_DEVICES per rank): each rank then
    # sees ONE device, number 0, whatever LOCAL_RANK says.  Never inferred: a box that shows the whole job one GPU looks the same.
    isolated = args.isolated_devices and ndev.value == 1
    if not host_comm and not isolated and local_rank >= ndev.value:
        raise SystemExit("bench.py: rank %d needs GPU %d but only %d are visible (one rank per GPU; --comm host "
                         "rehearses more ranks than GPUs)" % (rank, local_rank, ndev.value))
    dev_index = 0 if isolated else (local_rank % ndev.value if host_comm else local_rank)

    N, K, W = args.lanes, args.steps, args.warmup
    lane_lo, lane_hi = shard_range(world * N, rank, world)      # contiguous global lane ids of this rank
    assert lane_hi - lane_lo == N
    KG = K                                       # the captured sequence: all K launches (an odd K gets one tick-move node, include/soccer_hip.h)
    # The synthetic action trajectory is [max(K, W), 2, N] int8, read once per replay.  When it cannot stay in the 256 MB Infinity
    # Cache between replays (K = 1000: 2 GB) the handle is told that its action streams stream in from HBM (SOCCER_F_STREAM_ACTIONS:
    # non-temporal loads); a short sequence (the driver's K = 20: 40 MB) is re-read from the cache like the action buffer of an
    # RL loop, the library's default (include/soccer_hip.h).  --action-loads overrides the choice.
    KA = max(K, W, args.rollout, 1)
    act_bytes = max(K, W) * 2 * N
    stream_actions = {"auto": act_bytes > (96 << 20), "nt": True, "plain": False}[args.action_loads]
    b = SoccerBatch(N, 5, 4, args.slip, seed=0, autoreset=True, device=dev_index,
                    lane_offset=lane_lo, envs_per_thread=args.envs_per_thread, step_stats=False, stream_actions=stream_actions)
    comm = None
    if world > 1:
        comm = HostComm(rank, world, args.collective_timeout) if host_comm else RcclComm(b, rank, world, args.collective_timeout)

    # synthetic inputs, resident in HBM before the timed region: uniform-random joint actions for
    # every step; outputs stream into [K, N] trajectory buffers (nothing is cached or skipped)
    rng = np.random.default_rng(2024 + rank)
    acts = b.alloc((KA, 2, N), np.int8)
    rows = max(1, (64 << 20) // (2 * N))
    for k0 in range(0, KA, rows):
        acts.upload_rows(k0, rng.integers(0, 5, size=(min(rows, KA - k0), 2, N), dtype=np.int8))
    obs = b.alloc((K, N), np.uint16); rew = b.alloc((K, N), np.int8)
    term = b.alloc((K, N), np.uint8); trunc = b.alloc((K, N), np.uint8)
    act_row = 2 * N

    def enqueue(k):         # the 8-argument batched_step: 2 action streams in, 4 result streams out
        b.step_plain(acts.ptr + k * act_row, acts.ptr + k * act_row + N, obs.row(k), rew.row(k), term.row(k), trunc.row(k))

    b.reset()
    for k in range(W):
        enqueue(k % K)
    b.sync()
    graph = graph_s = None
    warm_replays, first_ms = 0, None
    if args.mode == "graph" and KG > 0:
        # the graph that is timed: exactly the K launches, nothing else
        b.graph_begin()
        for k in range(KG):
            enqueue(k)
        graph = b.graph_end()         # instantiated and uploaded (hipGraphUpload)
        # its twin for the device-side figures: the same launches between two clock stamps (each stamp is a one-thread kernel
        # of its own — ~1.9 us of device time that does not belong in a region that times K steps, which is why the timed graph
        # has none); replayed after the timed region under the same conditions
        b.graph_begin()
        b.timer_start()
        for k in range(KG):
            enqueue(k)
        b.timer_mark()
        graph_s = b.graph_end()
        # Untimed replays (the workload is stationary: advancing the state changes nothing): the timed region then pays none
        # of the path's one-off costs — three of the graph that is timed, each followed by a synchronisation like the timed
        # one (a replay's first and second launch have been seen to cost the host 3x the usual hipGraphLaunch call), one of
        # the twin.
        for _ in range(3):
            b.graph_launch(graph, 1); b.sync()
        b.graph_launch(graph_s, 1); first_ms = b.timer_read()
        warm_replays = 4
        b.sync()
    b.reset_stats()
    eager_args = None
    if graph is None:       # eager launches: device addresses resolved before the timed region
        eager_args = [(acts.ptr + k * act_row, acts.ptr + k * act_row + N, obs.row(k), rew.row(k), term.row(k), trunc.row(k), None)
                      for k in range(K)]

    # (names bound before the clock starts: at K = 20 the region is ~100 us and a Python attribute chain is ~0.3 us)
    graph_launch, handle, check = lib.soccer_graph_launch, b.h, b._check
    sync_call = lib.soccer_sync

    def device_sync():                           # hipStreamSynchronize on the handle's stream: the only stream with work
        rc = sync_call(handle)
        if rc:
            check(rc)
    rc_box, ev_box = [0], [None]
    if graph is not None:
        def launch():
            rc_box[0] = graph_launch(handle, graph, 1)
    else:
        step = lib.batched_step

        def launch():
            b.timer_start()
            for a in eager_args:
                check(step(handle, *a))
            b.timer_mark()
    def measure(job_barrier):
        t0_, t_enq_, t_end_ = timed_region(job_barrier, device_sync, launch, time.perf_counter)
        if graph is not None:
            check(rc_box[0])
            # device-side duration of the same K launches: the stamped twin, replayed after an idle synchronisation like the timed
            # region was, three times; the median
            reps = []
            for _ in range(3):
                device_sync()
                b.graph_launch(graph_s, 1); reps.append(b.timer_read())
            device_sync()
            return t0_, t_enq_, t_end_, sorted(reps)[1] * K / KG
        return t0_, t_enq_, t_end_, b.timer_read()
    # A rehearsal with more ranks than GPUs (--comm host): ranks that share a device take the region IN TURN, so that each rank's
    # clocks are its own (wall - device region = what the host adds, comparable with the N = 1 line); never a measured path.
    turns = host_comm and world > ndev.value
    if turns:
        for turn in range(world):
            comm.barrier()
            if turn == rank:
                t0, t_enq, t_end, ev_ms = measure(None)
        comm.barrier()
    else:
        t0, t_enq, t_end, ev_ms = measure(comm.barrier if comm else None)
    wall = t_end - t0
    per_rank = None
    wall_own, ev_own = wall, ev_ms
    if comm:
        # every rank's own clocks, so that a scaling line shows rank skew; the job's time is the slowest rank's
        allr = comm.gather_f64([wall, ev_ms])
        per_rank = [{"rank": r, "wall_us": float(allr[r, 0]) * 1e6, "device_region_us": float(allr[r, 1]) * 1e3,
                     "host_overhead_us": float(allr[r, 0]) * 1e6 - float(allr[r, 1]) * 1e3,
                     "launch_us": float(allr[r, 1]) * 1e3 / K} for r in range(world)]
        wall, ev_ms = float(allr[:, 0].max()), float(allr[:, 1].max())

    # ---- after the timed region: where the launches of a replay spend their time --------------------------
    # The same K launches captured once more with a clock stamp between every two of them (a one-thread kernel: each
    # adds one more kernel boundary, so the deltas are launch + ~1.5 us; what matters is how they differ along the
    # replay).  Replayed right after an idle synchronisation like the timed region, three times; the last replay is reported.
    launch_profile = None
    if graph is not None and world == 1 and 2 <= KG <= 200:
        b.graph_begin()
        b.stamp(2)
        for k in range(KG):
            enqueue(k); b.stamp(3 + k)
        gp = b.graph_end()
        for _ in range(3):
            device_sync()
            b.graph_launch(gp, 1); device_sync()          # (the slots are read after the synchronisation: no clearing needed)
        ticks, khz = b.stamps(2, KG + 1)
        d_us = np.diff(ticks.astype(np.int64)) / (khz * 1e-3)
        b.graph_destroy(gp)
        # what one stamp kernel adds to a delta: the same replay shape with nothing but stamps in it
        b.graph_begin()
        for k in range(8):
            b.stamp(2 + k)
        gs = b.graph_end()
        for _ in range(3):
            device_sync()
            b.graph_launch(gs, 1); device_sync()
        st, _ = b.stamps(2, 8)
        stamp_us = float(np.median(np.diff(st.astype(np.int64))[2:]) / (khz * 1e-3))
        b.graph_destroy(gs)
        rest = float(d_us[4:].mean()) if KG > 4 else None
        launch_profile = {"what": "device-clock deltas between stamps placed after every launch of one %d-launch replay "
                                  "(each delta = one step launch + one stamp kernel)" % KG,
                          "us": [round(float(x), 3) for x in d_us],
                          "first4_mean_us": float(d_us[:4].mean()), "rest_mean_us": rest,
                          "stamp_kernel_us": stamp_us,
                          "steady_launch_us": rest - stamp_us if rest is not None else None,
                          "steady_frac": (ALGO_BYTES_PER_ENV_STEP * N / ((rest - stamp_us) * 1e-6)) / 8e12 if rest is not None else None,
                          "note": "steady_launch_us = rest_mean_us - stamp_kernel_us (stamp to stamp in a stamps-only replay): an upper bound "
                                  "of the per-launch device time once the replay is under way (a stamp kernel between two step kernels "
                                  "costs more than between two stamps); us[0] carries the replay's cold start"}

    # ---- after the timed region: episode returns from the trajectories the timed steps wrote ---------
    # (the only cross-GPU exchange: one all-gather of int8 per-lane returns + a 3-bin sum)
    _, misuse = b.stats()
    assert misuse == 0
    last_ret = b.alloc(N, np.int8); ep_count = b.alloc(N, np.int32)
    hist = b.trajectory_returns(K, rew, term, trunc, N, last_return=last_ret, episode_count=ep_count).astype(np.int64)
    # cheap end-to-end sanity on this rank's real outputs of the timed steps (not a parity test)
    n_fin = int(hist.sum()); r_sum = int(hist[2]) - int(hist[0])
    assert n_fin < K * N and abs(r_sum) <= n_fin, "implausible outputs"
    assert int(ep_count.download().sum()) == n_fin
    for k in sorted({0, K // 2, K - 1}):
        o_k, r_k = obs.download_rows(k, 1), rew.download_rows(k, 1)
        te_k, tr_k = term.download_rows(k, 1), trunc.download_rows(k, 1)
        assert int(o_k.max()) < b.nS and int(np.abs(r_k).max()) <= 1 and int(te_k.max()) <= 1 and int(tr_k.max()) <= 1, "implausible outputs"
        assert not ((r_k != 0) & (te_k == 0)).any(), "a reward without a terminal step"
    if K >= 50:                     # long enough for goals to have been scored
        assert n_fin > 0 and hist[0] + hist[2] > 0, "implausible outputs"
    gather_ms = None
    if comm:
        gathered_d = b.alloc(world * N, np.int8)
        b.sync(); comm.barrier(); tg = time.perf_counter()
        comm.all_gather_lanes(b, last_ret, gathered_d, N)            # RCCL all-gather over xGMI, int8[N] per rank, global lane order
        hist = comm.sum_u64(hist.astype(np.uint64)).astype(np.int64)
        b.sync(); gather_ms = (time.perf_counter() - tg) * 1e3       # the job's ONLY exchange (first call: incl. RCCL set-up)
        gathered = gathered_d.download()
        assert gathered.size == world * N
        assert np.array_equal(gathered[lane_lo:lane_hi], last_ret.download()), "this rank's shard is not where its global lane ids say"
    else:
        gathered = last_ret.download()
    for x in (obs, rew, term, trunc):
        x.free()

    # ---- optional: fused T-step rollout (state in registers, same per-step results) -------------
    rollout = None
    if args.rollout > 0:
        T = args.rollout
        obs = b.alloc((T, N), np.uint16); rew = b.alloc((T, N), np.int8)
        term = b.alloc((T, N), np.uint8); trunc = b.alloc((T, N), np.uint8)
        roll = lambda: b.rollout(T, acts.ptr, acts.ptr + N, act_stride=2 * N, obs=obs, reward=rew, terminated=term,
                                 truncated=trunc, out_stride=N)
        roll()                                            # warm
        device_sync()
        reps = []
        for _ in range(5):
            b.timer_start(); roll(); reps.append(b.timer_stop())
        r_ms = sorted(reps)[len(reps) // 2]
        if comm:
            r_ms = float(comm.max_f64([r_ms])[0])
        bytes_per = 7 + 12.0 / T                         # 2 B actions in, 5 B out, state amortised over T
        rollout = {"steps_fused": T, "env_steps_per_s": world * N * T / (r_ms * 1e-3),
                   "bytes_per_env_step": bytes_per,
                   "achieved_GBps": bytes_per * world * N * T / (r_ms * 1e-3) / 1e9,
                   "frac_of_hbm_peak": bytes_per * world * N * T / (r_ms * 1e-3) / 1e9 / (HBM_PEAK_GBPS * world),
                   "kernel": "soccer::rollout_swar_kernel<0, %s, 1>" % ("1|2" if args.slip else "0")}   # as rocprofv3 prints it
        for x in (obs, rew, term, trunc):
            x.free()

    # ---- optional: BASELINE config 5 shape — both players sample from [nS, 5] mixed policies in-kernel --
    selfplay = None
    if args.rollout > 0:
        T = 100
        rngp = np.random.default_rng(94)
        da = b.alloc((b.nS, 4), np.uint16).upload(SoccerBatch.mixed_policy_thresholds(rngp.dirichlet(np.ones(5) * 0.7, size=b.nS)))
        db = b.alloc((b.nS, 4), np.uint16).upload(SoccerBatch.mixed_policy_thresholds(rngp.dirichlet(np.ones(5) * 0.7, size=b.nS)))
        b.rollout(T, sample_actions=True, mix_a=da, mix_b=db)               # warm
        device_sync()
        b.reset_stats(); h0 = b.stats()[0].astype(np.int64)
        reps = []
        for _ in range(3):
            b.timer_start(); b.rollout(T, sample_actions=True, mix_a=da, mix_b=db); reps.append(b.timer_stop())
        s_ms = sorted(reps)[1]
        hsum = b.stats()[0].astype(np.int64) - h0
        if comm:
            s_ms = float(comm.max_f64([s_ms])[0])
            hsum = comm.sum_u64(hsum.astype(np.uint64)).astype(np.int64)
        selfplay = {"horizon": T, "env_steps_per_s": world * N * T / (s_ms * 1e-3),
                    "return_hist_minus1_0_plus1_over_3_rollouts": [int(x) for x in hsum]}

    if rank == 0:
        # roofline.achieved / frac follow from the SAME clock as `value`: the host wall clock around the K steps.  The
        # device-side figure — clock stamps captured around the K launches, i.e. without the replay's
        # start-up latency and the host's wake-up — is reported next to it as frac_device.
        step_s = wall / K
        achieved = ALGO_BYTES_PER_ENV_STEP * N / step_s / 1e9
        launch_s = ev_ms * 1e-3 / K
        achieved_dev = ALGO_BYTES_PER_ENV_STEP * N / launch_s / 1e9
        # HBM-side bytes per launch: from a separate rocprofv3 --pmc run (tools/collect_profile.py), reported only when
        # that run measured THIS workload with THESE kernel sources (sha256 of csrc/ recorded next to the number)
        traffic, traffic_source = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile) and N == (1 << 20) and args.slip == 0.0:
            try:
                tj = json.load(open(tfile))
                traffic_source = {"file": "profiles/traffic.json", "profile": tj.get("build"), "commit": tj.get("commit"),
                                  "kernel": tj.get("kernel"), "csrc_sha256": tj.get("csrc_sha256"),
                                  "matches_this_build": tj.get("csrc_sha256") == csrc_sha256()}
                if traffic_source["matches_this_build"]:
                    traffic = tj.get("step_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic, traffic_source = None, None
        working_set = 12 * N + 7 * N * K              # resident state (6 B read + written in place) + the K rows of actions and results
        out = {
            "metric": "env-steps/sec (whole node) at batch=1M random joint actions; HBM GB/s vs peak",
            "value": world * N * K / wall, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall * 1e3 / K, "higher_is_better": True, "scaling": "weak",
            "untimed": {"eager_warmup_steps": W, "graph_replays_before_timing": warm_replays, "steps_per_replay": KG,
                        "first_replay_ms": first_ms},
            "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": {"workload": "%d envs/GPU x %d GPU(s), SoccerSimultaneous 5x4, slip_prob=%g, "
                                   "uniform-random joint actions, auto-reset, int8 SoA state, %s launches"
                                   % (N, world, args.slip, args.mode),
                       "lanes_per_gpu": N, "global_lanes": world * N, "slip_prob": args.slip,
                       "parallelism": "independent lane shards x%d" % world,
                       "action_loads": "%s (%s; action trajectory %d MB)" % ("non-temporal" if stream_actions else "plain", args.action_loads, act_bytes >> 20),
                       "host": {"hip_runtime": args.hip_runtime, "torch_in_process": "torch" in sys.modules,
                                "comm": comm.name if comm else None, "cpus": pinned, "placement": pin_note}},
            "timed_region": "job barrier + device sync | t0 | %s | this rank's device sync | t1;  job time = max over ranks%s"
                            % ("hipGraphLaunch of the %d captured launches" % KG if graph is not None else "%d eager launches" % K,
                               "; REHEARSAL: ranks share a device and take the region in turn" if turns else ""),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         # which memory the run's traffic is served from: the whole working set of a short run stays in the 256 MB
                         # Infinity Cache between replays; a long one streams from HBM.  Both are priced against the HBM peak.
                         "regime": "infinity-cache-resident" if working_set <= INFINITY_CACHE_BYTES else "hbm-streaming",
                         "working_set_bytes": working_set,
                         "bound_note": "bytes counted at the L2<->fabric boundary (FETCH_SIZE / WRITE_SIZE): Infinity-Cache hits "
                                       "are included, and the ~%d MB working set of a run fits the 256 MB Infinity Cache when K is small"
                                       % (working_set >> 20),
                         "kernel": "soccer::step_kernel_swar<0, %s, false, 1, false>" % ("2" if args.slip else "0"),   # as rocprofv3 prints it (slips the table form does not cover: 1)
                         "frac_from": "ms_per_step (host wall clock, the clock `value` uses)",
                         "achieved_device": achieved_dev, "frac_device": achieved_dev / HBM_PEAK_GBPS,
                         "launch_us": launch_s * 1e6, "device_region_us": ev_ms * 1e3,
                         "host_overhead_us": wall_own * 1e6 - ev_own * 1e3,
                         # where the host's share of the timed region goes (rank 0): the launch call(s), and from their return to the
                         # return of the one synchronisation that closes the region
                         "host_timeline_us": {"launch_call": (t_enq - t0) * 1e6, "launch_return_to_synchronised": (t_end - t_enq) * 1e6},
                         "device_figures_from": "a stamped twin of the timed graph (the same K launches between two device clock stamps), "
                                                "median of three replays after the timed region" if graph is not None else "HIP events around the eager launches",
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * N},
            "episodes": {"hist_minus1_0_plus1": [int(x) for x in hist],
                         "gathered_last_returns": int(gathered.size), "gather_allreduce_ms": gather_ms,
                         "gathered_mean": float(gathered.astype(np.float32).mean())},
        }
        if per_rank:
            out["per_rank"] = per_rank
        if launch_profile:
            out["launch_profile"] = launch_profile
        if rollout:
            out["fused_rollout"] = rollout
        if selfplay:
            out["selfplay_rollout_config5"] = selfplay
        if vec_env:
            out["vector_env_device"] = vec_env
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, args.slip, args.cpu_seconds)
        out["reference_python"] = REFERENCE_PYTHON
        print(json.dumps(out))
    if graph is not None:
        b.graph_destroy(graph); b.graph_destroy(graph_s)
    if comm:
        comm.close()
    b.close()


if __name__ == "__main__":
    main()
