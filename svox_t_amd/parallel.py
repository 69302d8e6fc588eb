"""Multi-GPU rendering: one process per GPU, rays sharded, tree replicated.

The reference has no multi-device code at all (grep for
nccl / distributed finds nothing), so this is new.  The path shards naturally:
rays are independent in the forward (svox_t/csrc/rt_kernel.cu:661-670, no
inter-thread communication) and the backward meets only at the sum over rays of
the feature gradient.

    partitioning   every rank holds the full tree (child, data, features);
                   rank r renders the contiguous ray range shard_bounds(Q, W, r)
                   (row tiles of the image when rays are row-major pixels)
    forward        local render, then ONE all-gather of [Q/W, C+1] fp32 tiles
    backward       local backward of the rank's own rows of grad_out, then ONE
                   all-reduce(sum) of grad_features [M, K]

`torch.distributed` with backend "nccl" is RCCL on ROCm (xGMI inside a node);
the same code runs on "gloo" for the CPU tests.  The render function is
injectable so the collective logic can be tested without a GPU.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist
from torch import autograd


def shard_bounds(Q: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of `Q` items for `rank` of `world`: the
    first Q % world ranks get one extra item."""
    base, rem = divmod(Q, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rays(rays, world: int, rank: int):
    """Slice a Rays namedtuple (origins, dirs, viewdirs) to this rank's range."""
    lo, hi = shard_bounds(rays.origins.shape[0], world, rank)
    return type(rays)(*(t[lo:hi].contiguous() for t in rays))


def _gather_rows(local: torch.Tensor, Q: int, group) -> torch.Tensor:
    """All-gather row shards of unequal length into the full [Q, ...] tensor."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    cap = (Q + world - 1) // world                  # rows per rank, padded
    tail = local.shape[1:]
    if local.shape[0] != cap:
        pad = local.new_zeros((cap,) + tuple(tail))
        pad[:local.shape[0]] = local
        local = pad
    full = local.new_empty((world * cap,) + tuple(tail))
    dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    if Q == world * cap:
        return full
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(Q, world, r)
        parts.append(full[r * cap:r * cap + (hi - lo)])
    return torch.cat(parts, dim=0)


class _ShardedRender(autograd.Function):
    """features -> full image, computed cooperatively.  Every rank must call it
    with the same features / rays and apply the same loss to the result, so
    that grad_out is identical on all ranks (data-parallel convention)."""

    @staticmethod
    def forward(ctx, features, render_fn, rays, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        Q = rays.origins.shape[0]
        local_rays = shard_rays(rays, world, rank)
        with torch.enable_grad():
            feats = features.detach().requires_grad_(True)
            local_out = render_fn(feats, local_rays)
        ctx.feats, ctx.local_out, ctx.group = feats, local_out, group
        ctx.bounds = shard_bounds(Q, world, rank)
        return _gather_rows(local_out.detach(), Q, group)

    @staticmethod
    def backward(ctx, grad_full):
        lo, hi = ctx.bounds
        if hi > lo:
            (g,) = torch.autograd.grad(ctx.local_out, ctx.feats, grad_full[lo:hi].contiguous())
        else:
            g = torch.zeros_like(ctx.feats)
        g = g.contiguous()
        if dist.get_world_size(ctx.group) > 1:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None, None, None


def render_sharded(renderer_or_fn, features: torch.Tensor, rays,
                   group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Render `rays` [Q, 3] cooperatively on all ranks of `group`.

    :param renderer_or_fn: a VolumeRenderer, or any callable (features, Rays) -> [q, C+1]
    :return: the full [Q, C+1] output on every rank; differentiable wrt
             `features` (gradient all-reduced over ranks).
    """
    fn: Callable = renderer_or_fn if not hasattr(renderer_or_fn, "forward") \
        else (lambda f, r: renderer_or_fn(f, r))
    return _ShardedRender.apply(features, fn, rays, group)


def broadcast_tree(tree, src: int = 0, group: Optional[dist.ProcessGroup] = None) -> None:
    """Make every rank's replica of the tree identical to rank `src`'s
    (buffers and the feature table).  Shapes must already agree."""
    for t in (tree.child, tree.data, tree.parent_depth, tree.invradius, tree.offset, tree.features.data):
        dist.broadcast(t, src=src, group=group)
