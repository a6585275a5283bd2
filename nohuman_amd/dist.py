"""Multi-GPU plumbing of the hot path: reads shard across ranks with the
database replicated in every GPU's HBM; the ONLY collective is the final sum of the per-rank
counters {fragments, classified, bases, table lookups} -- what kraken2's three stderr summary
integers (/root/reference/src/lib.rs:61-97) become when the run is spread over several devices.
One process per GPU, torch.distributed ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests)."""
from __future__ import annotations

import os


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, order-preserving split of n_items fragments over `world` ranks: rank r gets
    [lo, hi).  Sizes differ by at most one; concatenating the ranks' outputs in rank order restores
    input order (kraken2 writes reads in input order)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def reduce_counters(counters, elapsed_seconds: float, group=None):
    """counters: 1-D integer torch tensor of per-rank totals.  Returns (summed counters as a list of
    ints, max elapsed seconds over ranks).  No-op without an initialised process group."""
    import torch
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    # (gloo reduces host tensors: the CPU tests, and bench.py's one-GPU test mode)
    dev = "cpu" if on and dist.get_backend(group) == "gloo" else counters.device
    tot = counters.clone().to(dev)
    tmax = torch.tensor([float(elapsed_seconds)], dtype=torch.float64, device=dev)
    if on:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
    return [int(x) for x in tot.tolist()], float(tmax.item())


def gather_floats(value: float, device=None, group=None):
    """One float per rank, in rank order (e.g. each rank's kernel time for the bench line)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [float(value)]
    if dist.get_backend(group) == "gloo":
        device = "cpu"
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, mine, group=group)
    return [float(t.item()) for t in out]


def usable_cpu_count() -> int:
    """Threads the host side may really use: the scheduler affinity capped by the cgroup CPU quota
    (the GPU boxes expose 256 logical CPUs but run under a 16-CPU quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n
