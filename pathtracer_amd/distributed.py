"""Multi-GPU plumbing: one process per GPU, sample-index sharding, one integer all-reduce at readback.

Every pixel-sample is independent (reference src/tracer.rs:628-646) and the accumulator is an exact u32 sum
(src/tracer.rs:258-269), so rank r of W renders the global sample indices {s : s % W == r} and the ranks'
accumulators are summed once with RCCL (torch.distributed backend "nccl" is RCCL on ROCm).  The result is bit
identical to one GPU rendering all the samples.  torch is used for the process group and the collective only.
"""
from __future__ import annotations

import os
from typing import List, Tuple


def env_rank() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torch.distributed.run environment; (0, 1, 0) without it."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def shard_sample_indices(rank: int, world: int, n_local: int, first_local: int = 0) -> List[int]:
    """Global sample indices rendered by `rank` for its local samples first_local .. first_local + n_local - 1."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return [rank + (first_local + k) * world for k in range(n_local)]


def split_samples(total: int, world: int) -> List[int]:
    """Local sample counts when `total` global samples 0..total-1 are dealt round-robin to `world` ranks."""
    return [len(range(r, total, world)) for r in range(world)]


def allreduce_sum_(tensor):
    """In-place SUM all-reduce of an int32 accumulator tensor over the default process group (no-op for one rank).
    u32 sums stay below 2^31 for any practical sample count (255 per sample), so the int32 view is exact."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("PTGPU_FORCE_COLLECTIVE")):
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def combine_accumulators(gpu, local_samples: int):
    """All-reduce a PtGpu handle's accumulator across ranks and set the global sample count on every rank.
    Returns the global sample count."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 and not (dist.is_initialized() and os.environ.get("PTGPU_FORCE_COLLECTIVE")):
        return local_samples  # (the env switch runs the export -> all-reduce -> import path on one rank, for testing)
    n_words = gpu.width * gpu.height * 3
    buf = torch.empty(n_words, dtype=torch.int32, device=torch.device("cuda", gpu.cfg.device))
    gpu.export_pixels32(buf.data_ptr())
    if dist.get_backend() == "nccl":
        allreduce_sum_(buf)
        total = torch.tensor([local_samples], dtype=torch.int64, device=buf.device)
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    else:
        # a process group without device collectives (gloo: several ranks sharing one GPU, a debugging set-up): the
        # same sums through host memory
        host = buf.cpu()
        allreduce_sum_(host)
        buf.copy_(host)
        total = torch.tensor([local_samples], dtype=torch.int64)
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize(buf.device)
    gpu.import_pixels32(buf.data_ptr(), int(total.item()))
    return int(total.item())
