"""File -> GPU sharding for batch runs (mirrors the reference's bounded per-file worker pool, cmd/jivetalking/pool.go:122-153:
files are independent units, one in flight per worker, no exchange between them).  One process per GPU; the only
cross-rank traffic is control-plane (a barrier and a MAX over wall-clock), never audio data — no RCCL collective on
the data path."""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def assign_files(n_files, world, rank, durations=None):
    """Longest-first greedy assignment of files to ranks.  Returns the file indices for `rank`.
    Deterministic: every rank computes the same partition without communication."""
    order = list(range(n_files))
    if durations is not None:
        order.sort(key=lambda i: (-float(durations[i]), i))
    loads = [0.0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        loads[r] += float(durations[i]) if durations is not None else 1.0
        if r == rank:
            mine.append(i)
    return mine


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX-reduce a python float over all ranks (identity when not distributed)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value, device="cpu"):
    """Every rank's value as a list (rank order); [value] when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]
