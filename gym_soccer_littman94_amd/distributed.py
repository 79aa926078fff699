"""Multi-GPU plumbing: one process per GPU, independent lane shards, no data-path collective.

The lanes of the batch never interact, so G GPUs simply own G contiguous ranges of global lane ids;
per-lane Philox streams are keyed by the GLOBAL id, which makes every result independent of G.
The only exchange is after a run: an all_gather of per-lane episode returns (int8) and an all_reduce
of the 3-bin return histogram — `torch.distributed` with backend "nccl" (= RCCL over xGMI) on GPUs,
"gloo" in the CPU tests.
"""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), \
        int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(global_lanes, rank, world):
    """Contiguous [lo, hi) of `global_lanes` owned by `rank`; the first (global_lanes % world) ranks
    get one extra lane so that any lane count shards."""
    assert 0 <= rank < world and global_lanes >= world
    base, extra = divmod(int(global_lanes), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_lane_values(local, global_lanes, group=None):
    """all_gather per-lane values (1-D tensor on this rank's device) into global lane order.
    Equal shards (BASELINE config 4: 8 x 2^20) go through ONE all_gather_into_tensor into the output
    buffer; otherwise shards differ by one lane and are padded to the largest one and trimmed."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    base, extra = divmod(int(global_lanes), world)
    if extra == 0:
        assert local.numel() == base, "this rank's shard has %d lanes, expected %d" % (local.numel(), base)
        out = torch.empty(base * world, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    width = base + 1
    padded = torch.zeros(width, dtype=local.dtype, device=local.device)
    padded[: local.numel()] = local
    flat = torch.empty(width * world, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(flat, padded, group=group)
    out = []
    for r in range(world):
        lo, hi = shard_range(global_lanes, r, world)
        out.append(flat[r * width: r * width + hi - lo])
    return torch.cat(out)


def reduce_histogram(hist, device=None, group=None):
    """Sum the (-1, 0, +1) episode-return counts over ranks; returns a list of ints."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(x) for x in hist], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [int(x) for x in t.cpu()]


def gather_rank_values(values, device=None, group=None):
    """Every rank's small vector of float64 values (its own clocks, say) -> a [world, len(values)] CPU tensor on every
    rank.  One all_gather_into_tensor into a FLAT buffer (the form both gloo and RCCL accept)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    flat = torch.empty(world * mine.numel(), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(flat, mine, group=group)
    return flat.cpu().view(world, mine.numel())
