"""Bring-up of the job's communicator: one process per GPU, one SoccerBatch handle per process.

Stepping needs no collective (lanes never interact; per-lane Philox is keyed by the GLOBAL lane id), so the only exchange
of a multi-GPU job is after a run — the all-gather of per-lane episode returns into global lane order and a few small
reductions (BASELINE configs[3]).  Two interchangeable backends with the same five methods
(barrier / sum_u64 / max_f64 / all_gather_lanes / close):

  RcclComm   RCCL over xGMI through the C ABI (soccer_comm_*): rank 0 draws the 128-byte unique id, every rank learns
             it through a file next to the launcher, ncclCommInitRank.  No torch in the process.
  HostComm   the same exchanges through files, host memory only: for rehearsing N ranks on FEWER GPUs than ranks (RCCL
             refuses two ranks on one device) and for the CPU tests of the launcher.  Never a measured path.

Ranks, world size and the rendezvous come from the environment torch.distributed.run sets (RANK, WORLD_SIZE, LOCAL_RANK,
MASTER_PORT); `python bench.py --gpus N` sets the same variables for the ranks it starts itself.
"""
import ctypes
import os
import time

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def rendezvous_dir():
    """One directory per job: every rank of a job has the same parent (the launcher) and the same MASTER_PORT."""
    base = os.environ.get("SOCCER_COMM_DIR") or os.path.join(
        os.environ.get("TMPDIR", "/tmp"), "soccer_comm_%d_%d_%s" % (os.getuid(), os.getppid(), os.environ.get("MASTER_PORT", "0")))
    os.makedirs(base, exist_ok=True)
    return base


def _publish(path, payload):
    tmp = "%s.tmp%d" % (path, os.getpid())
    with open(tmp, "wb") as f:
        f.write(payload)
    os.replace(tmp, path)                  # atomic: a reader sees nothing or everything


def _await(path, nbytes, deadline, newer_than=0.0):
    while True:
        try:
            if os.path.getmtime(path) >= newer_than:
                with open(path, "rb") as f:
                    data = f.read()
                if len(data) == nbytes:
                    return data
        except OSError:
            pass
        if time.monotonic() > deadline:
            raise TimeoutError("rendezvous: %s did not appear in time (is a rank missing?)" % path)
        time.sleep(0.0005)


class HostComm:
    """File-based stand-in with RcclComm's interface (host memory only; rehearsals and CPU tests)."""
    name = "host-files"

    def __init__(self, rank, world, timeout=120.0, directory=None):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self.dir = directory or rendezvous_dir()
        self.seq = 0

    def _exchange(self, payload):
        """every rank's payload (equal sizes), in rank order"""
        self.seq += 1
        deadline = time.monotonic() + self.timeout
        mine = os.path.join(self.dir, "x%d_%d" % (self.seq, self.rank))
        _publish(mine, payload)
        out = [payload if r == self.rank else _await(os.path.join(self.dir, "x%d_%d" % (self.seq, r)), len(payload), deadline)
               for r in range(self.world)]
        # whoever has finished exchange k has seen every rank's file k, so every rank has finished READING exchange k - 1
        prev = os.path.join(self.dir, "x%d_%d" % (self.seq - 1, self.rank))
        if self.seq > 1 and os.path.exists(prev):
            os.unlink(prev)
        return out

    def barrier(self):
        self._exchange(b"\x01")

    def sum_u64(self, values):
        a = np.ascontiguousarray(values, np.uint64)
        return np.sum([np.frombuffer(x, np.uint64) for x in self._exchange(a.tobytes())], axis=0, dtype=np.uint64)

    def max_f64(self, values):
        a = np.ascontiguousarray(values, np.float64)
        return np.max([np.frombuffer(x, np.float64) for x in self._exchange(a.tobytes())], axis=0)

    def gather_f64(self, values):
        """[world, len(values)]: every rank's small vector (its own clocks, say)"""
        a = np.ascontiguousarray(values, np.float64)
        return np.stack([np.frombuffer(x, np.float64) for x in self._exchange(a.tobytes())])

    def all_gather_lanes(self, batch, send, recv, bytes_per_rank):
        """device -> host -> files -> device: what RcclComm does in one ncclAllGather"""
        host = np.empty(int(bytes_per_rank), np.uint8)
        batch._check(batch.lib.soccer_memcpy_d2h(batch.h, host.ctypes.data, send.ptr, host.nbytes))
        flat = np.frombuffer(b"".join(self._exchange(host.tobytes())), np.uint8)
        batch._check(batch.lib.soccer_memcpy_h2d(batch.h, recv.ptr, flat.ctypes.data, flat.nbytes))

    def close(self):
        """Last exchange, then every rank leaves a `done` mark: it will read nothing any more.  Rank 0 removes the directory once
        all marks are there (no rank deletes a file another rank may still be reading)."""
        try:
            self.barrier()
            _publish(os.path.join(self.dir, "done_%d" % self.rank), b"\x01")
            if self.rank == 0:
                deadline = time.monotonic() + min(self.timeout, 30.0)
                for r in range(self.world):
                    _await(os.path.join(self.dir, "done_%d" % r), 1, deadline)
                for f in os.listdir(self.dir):
                    try:
                        os.unlink(os.path.join(self.dir, f))
                    except OSError:
                        pass
                os.rmdir(self.dir)
        except Exception:
            pass


class RcclComm:
    """RCCL over xGMI through libsoccer_hip.so's soccer_comm_* entry points, bound to one SoccerBatch handle."""
    name = "rccl"

    def __init__(self, batch, rank, world, timeout=120.0, directory=None):
        from . import _lib
        self.batch, self.rank, self.world = batch, int(rank), int(world)
        d = directory or rendezvous_dir()
        path = os.path.join(d, "rccl_unique_id")
        t_start = time.time()
        if self.rank == 0:
            buf = (ctypes.c_uint8 * _lib.COMM_ID_BYTES)()
            _lib.check(batch.lib, None, batch.lib.soccer_comm_unique_id(buf))
            uid = bytes(buf)
            _publish(path, uid)
        else:
            # (a file left behind by an earlier job that had the same launcher pid and port is older than this process)
            uid = _await(path, _lib.COMM_ID_BYTES, time.monotonic() + timeout, newer_than=t_start - 300.0)
        batch.comm_init(self.world, self.rank, uid)       # collective: returns when every rank has joined
        self._path = path
        self.barrier()
        if self.rank == 0:
            try:
                os.unlink(path); os.rmdir(d)
            except OSError:
                pass

    def barrier(self):
        self.batch.comm_barrier()

    def sum_u64(self, values):
        return self.batch.comm_sum(values)

    def max_f64(self, values):
        return self.batch.comm_max(values)

    def gather_f64(self, values):
        """[world, len(values)] through one small all-gather of device scratch"""
        a = np.ascontiguousarray(values, np.float64)
        b = self.batch
        send = b.alloc(a.size, np.float64).upload(a); recv = b.alloc((self.world, a.size), np.float64)
        b.all_gather(send, recv, a.nbytes)
        out = recv.download()
        send.free(); recv.free()
        return out

    def all_gather_lanes(self, batch, send, recv, bytes_per_rank):
        batch.all_gather(send, recv, bytes_per_rank)

    def close(self):
        try:
            self.batch.comm_destroy()
        except Exception:
            pass


def connect(batch, backend="rccl", timeout=120.0):
    """The communicator of this rank's handle, from the launcher's environment; None for a single-rank job."""
    rank, world, _ = env_rank_world()
    if world == 1:
        return None
    if backend == "rccl":
        return RcclComm(batch, rank, world, timeout)
    if backend == "host":
        return HostComm(rank, world, timeout)
    raise ValueError("comm backend must be 'rccl' or 'host'")
