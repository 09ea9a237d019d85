"""Multi-GPU layout of a batch of independent streams.

Whole streams are sharded across ranks -- stream s lives on rank s // streams_per_gpu --
and never exchange data: a stream's three reference frames stay on its GPU.  The only
collectives are control-plane: a barrier around the timed region, MAX of the elapsed time
and an all-gather of per-stream checksums/counters (RCCL on GPUs, gloo in the CPU tests).
"""


def _torch():
    """torch and torch.distributed, imported on first use: stream_ids / stream_seed are plain arithmetic, and bench.py asks its
    worker processes for every stream's content before it pays for the import (1-2 minutes on a fresh box)."""
    import torch
    import torch.distributed as dist
    return torch, dist


def stream_ids(rank, world, streams_per_gpu):
    """Global stream ids owned by `rank` (weak scaling: the batch grows with the world)."""
    if not (0 <= rank < world) or streams_per_gpu < 0:
        raise ValueError("bad rank/world/streams_per_gpu")
    return list(range(rank * streams_per_gpu, (rank + 1) * streams_per_gpu))


def stream_seed(base_seed, stream_id):
    """Seed of a stream's synthetic content: a function of the GLOBAL stream id only, so a
    stream decodes to the same pictures wherever it is placed."""
    return int(base_seed) + 1000003 * int(stream_id)


def barrier(world):
    """Barrier over the ranks; a process group of one rank (torchrun --nproc-per-node 1) still goes
    through the backend, so that the collective path is exercised on a single-GPU box too."""
    torch, dist = _torch()
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reduce_results(elapsed, per_stream_values, device):
    """(max elapsed over ranks, values of all streams in global stream order)."""
    torch, dist = _torch()
    if not dist.is_initialized():
        return float(elapsed), [int(v) for v in per_stream_values]
    world = dist.get_world_size()
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    mine = torch.tensor([int(v) for v in per_stream_values], dtype=torch.int64, device=device)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return float(t.item()), [int(v) for p in parts for v in p.tolist()]


def reduce_max(values, device):
    torch, dist = _torch()
    if not dist.is_initialized():
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.tolist()]


def reduce_min(value, device):
    """Minimum of an integer over ranks (e.g. "every rank's parity check passed")."""
    torch, dist = _torch()
    if not dist.is_initialized():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def gather_floats(value, device):
    """One float of every rank, in rank order (e.g. every rank's own median step time: whether "RCCL saw N ranks" and how far
    the ranks are apart can then be read off the line rank 0 prints)."""
    torch, dist = _torch()
    if not dist.is_initialized():
        return [float(value)]
    world = dist.get_world_size()
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return [float(p.item()) for p in parts]
