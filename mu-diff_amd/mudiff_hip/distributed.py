"""Data-parallel plumbing for the sampling path: one process per GPU, slices sharded contiguously,
weights replicated.  The only collectives are ONE flattened broadcast per model at load (RCCL over xGMI
on the GPUs; the reference's per-tensor `broadcast_params`, engine/train.py:188-190, sent 288 + 318 small
messages) and scalar reductions for reporting.  There is no per-step exchange: slices are independent.
Backend-agnostic (works over gloo on CPU tensors, which is how the CPU tests cover the N > 1 path)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of `n_items` slices for `rank`: sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_parameters(module, src=0):
    """Replicate `module`'s parameters and buffers from `src` with one flattened broadcast."""
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    if not tensors or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    dist.broadcast(flat, src=src)
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t).to(t.dtype))
        off += t.numel()
    return flat.numel() * 4


def max_over_ranks(value, device):
    """MAX-reduce a python float (the timed-region length) over all ranks."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device):
    """SUM-reduce a list of floats (metric accumulators: sum PSNR, sum SSIM, count)."""
    t = torch.tensor(list(values), device=device, dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()
