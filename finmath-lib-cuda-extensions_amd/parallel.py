"""Path sharding over GPUs: one process per GPU, rank g owns a contiguous block of paths of EVERY
vector; element-wise work needs no communication; the only exchange is one small collective over the per-rank
reduction partials {Σ, Σ², min, max}.  The reference has no multi-GPU support at all (README.md:33-35 names GPU memory
as its limit) — this is new capability, built on torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the
CPU tests).  Pure host logic: nothing here touches the device library."""
from __future__ import annotations

import math


def path_shard(n_total: int, world_size: int, rank: int, align: int = 4):
    """Contiguous block [offset, offset+count) of rank `rank`.  Boundaries are multiples of `align` paths (one Philox
    call covers 4 consecutive paths) except the very end; blocks differ in size by at most `align`."""
    if not (0 <= rank < world_size) or n_total < 0:
        raise ValueError("bad shard request")
    groups = (n_total + align - 1) // align
    base, extra = divmod(groups, world_size)
    g0 = rank * base + min(rank, extra)
    g1 = g0 + base + (1 if rank < extra else 0)
    off, end = min(g0 * align, n_total), min(g1 * align, n_total)
    return off, end - off


def combine_moments(gathered):
    """gathered: tensor/array [world, k, 4] of per-rank {Σ, Σ², min, max} → [k, 4] for the union of the shards.
    NaN in any rank's min/max propagates (java.lang.Math.min/max semantics, as on the device)."""
    import torch
    g = torch.as_tensor(gathered)
    out = torch.empty(g.shape[1:], dtype=g.dtype, device=g.device)
    out[:, 0] = g[:, :, 0].sum(0)
    out[:, 1] = g[:, :, 1].sum(0)
    mn, mx = g[:, :, 2], g[:, :, 3]
    out[:, 2] = torch.where(torch.isnan(mn).any(0), torch.full_like(mn[0], math.nan), mn.min(0).values)
    out[:, 3] = torch.where(torch.isnan(mx).any(0), torch.full_like(mx[0], math.nan), mx.max(0).values)
    return out


def all_gather_moments(local, group=None):
    """ONE collective: all-gather of this rank's [k,4] fp64 partials, combined on every rank.  Works on device tensors
    (RCCL) and on CPU tensors (gloo)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    local = local.contiguous()
    buf = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf.view(-1), local.view(-1), group=group)
    return combine_moments(buf)


def average_and_variance(moments_sum, moments_sumsq, n_total: int):
    """Mean and (population) variance from Σ, Σ² over n_total paths: E[X], E[X²] - E[X]²."""
    mean = moments_sum / n_total
    return mean, moments_sumsq / n_total - mean * mean
