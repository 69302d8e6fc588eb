"""Multi-GPU rendering: one process per GPU, rays sharded, tree replicated.

The reference has no multi-device code at all (grep for
nccl / distributed finds nothing), so this is new.  The path shards naturally:
rays are independent in the forward (svox_t/csrc/rt_kernel.cu:661-670, no
inter-thread communication) and the backward meets only at the sum over rays of
the feature gradient.

    partitioning   every rank holds the full tree (child, data, features);
                   rank r renders the contiguous ray range shard_bounds(Q, W, r)
                   (row tiles of the image when rays are row-major pixels)
                   -- render_sharded -- or whole cameras i = r, r + W, ...
                   in image mode -- render_cameras
    forward        local render, then ONE all-gather of [Q/W, C+1] fp32 tiles
                   (gather_pixels_async: under the backward; to one rank if only one needs it)
    backward       local backward of the rank's own rows of grad_out, then ONE
                   all-reduce(sum) of grad_features [M, K] -- or, across the batches of a
                   gradient-accumulation step, OverlappedGradReducer: row chunks on a side
                   stream under the next batch's forward and backward

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


def image_shard_bounds(height: int, width: int, world: int, rank: int) -> Tuple[int, int]:
    """[lo, hi) ray range of `rank` when the rays are the row-major pixels of a
    height x width image: whole bands of 8 rows per rank where the image allows it, so
    that every shard is itself an image the kernels can walk in 8x8 tiles (and the
    two-kernel backward can merge per tile)."""
    if height % 8 == 0 and width % 8 == 0 and height // 8 >= world:
        lo, hi = shard_bounds(height // 8, world, rank)
        return lo * 8 * width, hi * 8 * width
    return shard_bounds(height * width, world, rank)


def _gather_shards(local: torch.Tensor, bounds, group) -> torch.Tensor:
    """All-gather row shards of unequal length (bounds[r] = [lo, hi) of rank r, contiguous
    and in rank order) into the full tensor."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    cap = max(hi - lo for lo, hi in bounds)           # rows per rank, padded
    tail = local.shape[1:]
    if local.shape[0] != cap:
        pad = local.new_zeros((cap,) + tuple(tail))
        pad[:local.shape[0]] = local
        local = pad
    full = local.new_empty((world * cap,) + tuple(tail))
    dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    if all(hi - lo == cap for lo, hi in bounds):
        return full
    return torch.cat([full[r * cap:r * cap + (hi - lo)] for r, (lo, hi) in enumerate(bounds)], dim=0)


def _gather_rows(local: torch.Tensor, Q: int, group) -> torch.Tensor:
    """All-gather balanced row shards (shard_bounds) into the full [Q, ...] tensor."""
    world = dist.get_world_size(group)
    return _gather_shards(local, [shard_bounds(Q, world, r) for r in range(world)], group)


class _ShardedRender(autograd.Function):
    """features -> full image, computed cooperatively.  Every rank must call it
    with the same features / rays and apply the same loss to the result, so
    that grad_out is identical on all ranks (data-parallel convention)."""

    @staticmethod
    def forward(ctx, features, render_fn, rays, group, image_shape):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        Q = rays.origins.shape[0]
        if image_shape is not None and image_shape[0] * image_shape[1] == Q:
            bounds = [image_shard_bounds(image_shape[0], image_shape[1], world, r) for r in range(world)]
        else:
            image_shape = None
            bounds = [shard_bounds(Q, world, r) for r in range(world)]
        lo, hi = bounds[rank]
        local_rays = type(rays)(*(t[lo:hi].contiguous() for t in rays))
        local_shape = None
        if image_shape is not None and (hi - lo) % image_shape[1] == 0 and hi > lo:
            local_shape = ((hi - lo) // image_shape[1], image_shape[1])
        with torch.enable_grad():
            feats = features.detach().requires_grad_(True)
            local_out = render_fn(feats, local_rays, local_shape)
        ctx.feats, ctx.local_out, ctx.group = feats, local_out, group
        ctx.bounds = (lo, hi)
        return _gather_shards(local_out.detach(), bounds, group)

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
        return g, None, None, None, None


def render_sharded(renderer_or_fn, features: torch.Tensor, rays,
                   group: Optional[dist.ProcessGroup] = None, image_shape=None) -> torch.Tensor:
    """Render `rays` [Q, 3] cooperatively on all ranks of `group`.

    :param renderer_or_fn: a VolumeRenderer, or any callable (features, Rays) -> [q, C+1]
    :param image_shape: optional (H, W): the rays are the row-major pixels of an image;
           ranks then get whole bands of 8 rows and render them as images (8x8 tiles)
    :return: the full [Q, C+1] output on every rank; differentiable wrt
             `features` (gradient all-reduced over ranks).
    """
    if hasattr(renderer_or_fn, "forward"):
        fn: Callable = lambda f, r, shape: renderer_or_fn(f, r, image_shape=shape)
    else:
        fn = lambda f, r, shape: renderer_or_fn(f, r)
    return _ShardedRender.apply(features, fn, rays, group, image_shape)


class _CameraSet(autograd.Function):
    """features -> [n_cam, H, W, C+1]: camera i is rendered by rank i % world
    (image mode: the kernels generate the rays, nothing but the pose travels);
    ONE all-gather of the images forward, ONE all-reduce of the gradient backward."""

    @staticmethod
    def forward(ctx, features, render_fn, c2ws, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        n_cam = c2ws.shape[0]
        mine = list(range(rank, n_cam, world))
        per_rank = (n_cam + world - 1) // world
        with torch.enable_grad():
            feats = features.detach().requires_grad_(True)
            local = [render_fn(feats, c2ws[i]) for i in mine]
        ctx.feats, ctx.local, ctx.group, ctx.mine = feats, local, group, mine
        if world == 1:
            return torch.stack([im.detach() for im in local])
        # ranks with fewer cameras pad with zero images so that the gather is regular;
        # the image shape is taken from a rendered camera (ranks without any learn it
        # from rank 0, which always has camera 0)
        shape = torch.tensor(list(local[0].shape) if local else [0, 0, 0], device=c2ws.device)
        dist.broadcast(shape, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        H, W, C = (int(v) for v in shape.tolist())
        buf = features.new_zeros((per_rank, H, W, C))
        for j, im in enumerate(local):
            buf[j] = im.detach()
        full = features.new_empty((world * per_rank, H, W, C))
        dist.all_gather_into_tensor(full, buf, group=group)
        # slot r * per_rank + j holds camera r + j * world
        order = [(i % world) * per_rank + i // world for i in range(n_cam)]
        return full[torch.tensor(order, device=full.device)]

    @staticmethod
    def backward(ctx, grad_full):
        g = torch.zeros_like(ctx.feats)
        for im, i in zip(ctx.local, ctx.mine):
            (gi,) = torch.autograd.grad(im, ctx.feats, grad_full[i].contiguous())
            g += gi
        if dist.get_world_size(ctx.group) > 1:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None, None, None


def render_cameras(renderer_or_fn, features: torch.Tensor, c2ws: torch.Tensor,
                   group: Optional[dist.ProcessGroup] = None, **persp_kwargs) -> torch.Tensor:
    """Render a set of cameras, one (or a few) per rank: SURVEY.md 8(d) config 5,
    eight 1024 x 1024 cameras on eight GPUs gathered to [8, 1024, 1024, C+1].

    :param renderer_or_fn: a VolumeRenderer (its `render_persp` is used with
           `persp_kwargs`: width, height, fx, fy, fast), or any callable
           (features, c2w) -> [H, W, C+1]
    :param c2ws: [n_cam, 3 or 4, 4] camera-to-world matrices, identical on all ranks
    :return: [n_cam, H, W, C+1] on every rank, differentiable wrt `features`
             (gradient all-reduced over ranks)
    """
    if hasattr(renderer_or_fn, "render_persp"):
        fn: Callable = lambda f, c: renderer_or_fn.render_persp(f, c, **persp_kwargs)
    else:
        fn = renderer_or_fn
    return _CameraSet.apply(features, fn, c2ws, group)


class OverlappedGradReducer:
    """All-reduce(sum) of the feature gradient off the critical path.

    `start(grad)` hands over a gradient [M, K] as it comes out of a backward: it is reduced in
    place, in row-range chunks (each chunk one collective: on xGMI a chunk of a few tens of MB
    keeps all links busy without one 75 MB -- or 578 MB at depth 9 -- transfer holding the
    queue), on a side stream that waits for the backward's last kernel and nothing else (RCCL's
    own stream for the "nccl" backend).  The compute stream goes on -- the next batch's forward
    and backward run while the chunks travel.  `wait()` makes the compute stream (gloo: the host)
    wait for the reduced gradient; call it before the gradient is read.  One gradient is in
    flight at a time (`start` waits for the previous one).

    When that overlap is legitimate: gradient accumulation -- the sum of the gradients of several
    ray batches / cameras before one update of the features, the usual arrangement when every
    rank renders whole cameras -- and inference-time sensitivity passes.  A trainer that updates
    the features after EVERY batch must wait() before the update; what still overlaps then is the
    pixel gather (under the backward, gather_pixels_async) and the next batch's ray upload.
    bench.py --gpus N times the accumulation arrangement and says so.
    """

    def __init__(self, dist_module=dist, group=None, backend: str = "nccl", chunk_bytes: int = 32 << 20):
        self.dist, self.group, self.backend = dist_module, group, backend
        self.chunk_bytes = int(chunk_bytes)
        self._works, self._grad = [], None
        self._side = torch.cuda.Stream() if backend == "nccl" and torch.cuda.is_available() else None

    def chunks(self, grad: torch.Tensor):
        """Row ranges [lo, hi) of `grad` of at most chunk_bytes each (at least one row)."""
        M = grad.shape[0]
        row_bytes = max(1, grad[0].numel() * grad.element_size()) if M else 1
        rows = max(1, self.chunk_bytes // row_bytes)
        return [(lo, min(M, lo + rows)) for lo in range(0, M, rows)]

    def start(self, grad: torch.Tensor) -> None:
        self.wait()
        if grad is None or grad.numel() == 0 or self.dist.get_world_size(self.group) == 1:
            return
        assert grad.is_contiguous()
        self._grad = grad                                   # stays alive while it travels
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(grad.device))
            grad.record_stream(self._side)
            with torch.cuda.stream(self._side):
                self._works = [self.dist.all_reduce(grad[lo:hi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
                               for lo, hi in self.chunks(grad)]
        else:
            self._works = [self.dist.all_reduce(grad[lo:hi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
                           for lo, hi in self.chunks(grad)]

    def wait(self) -> Optional[torch.Tensor]:
        """The reduced gradient handed to the last start() (None if there was none)."""
        for w in self._works:
            w.wait()                                        # nccl: the current stream waits; gloo: the host does
        self._works = []
        g, self._grad = self._grad, None
        return g


class _Done:
    def wait(self):
        return True


def gather_pixels_async(dist_module, gathered: torch.Tensor, local: torch.Tensor, backend: str = "nccl",
                        group=None, dst: Optional[int] = None):
    """Start collecting every rank's [q, C+1] pixels into `gathered` [world * q, C+1] and return a
    handle with .wait(): with "nccl" the transfer runs beside whatever the compute stream does next
    (the backward).  dst = r: only rank r needs the image (a viewer, a logger): a gather to that
    rank moves 1/world of what the all-gather moves over every other rank's links."""
    world = dist_module.get_world_size(group)
    if world == 1:
        gathered.copy_(local)
        return _Done()
    if dst is not None:
        me = dist_module.get_rank(group)
        outs = list(gathered.chunk(world)) if me == dst else None
        if backend == "nccl":
            return dist_module.gather(local, outs, dst=dst, group=group, async_op=True)
        dist_module.gather(local, outs, dst=dst, group=group)
        return _Done()
    if backend == "nccl":
        return dist_module.all_gather_into_tensor(gathered, local, group=group, async_op=True)
    dist_module.all_gather(list(gathered.chunk(world)), local, group=group)
    return _Done()


def broadcast_tree(tree, src: int = 0, group: Optional[dist.ProcessGroup] = None) -> None:
    """Make every rank's replica of the tree identical to rank `src`'s
    (buffers and the feature table).  Shapes must already agree."""
    for t in (tree.child, tree.data, tree.parent_depth, tree.invradius, tree.offset, tree.features.data):
        dist.broadcast(t, src=src, group=group)
    # the collective wrote into `features.data` (no version bump) and maybe into child / data: what the
    # operator layer derived from the old contents (acceleration grid, a cached sigma bitmask) is stale
    try:
        from svox_t_amd import csrc as _C
    except ImportError:          # (the CPU tests of the collective logic run without the HIP library)
        _C = None
    if _C is not None:
        _C.invalidate_caches(tree.child, tree.data, tree.features)
    else:
        for t in (tree.child, tree.data, tree.features):
            torch.autograd.graph.increment_version(t)
