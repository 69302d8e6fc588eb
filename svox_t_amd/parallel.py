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


# Test switch: run every collective even in a group of ONE rank (where each is the identity and is normally skipped), so
# that a one-GPU box exercises the real backend's calls -- RCCL all-reduce / all-gather / broadcast launches, the side
# stream hand-offs, async work handles -- exactly as an N-rank job issues them (tests/test_gpu_nccl_world1.py).
FORCE_COLLECTIVES = False


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
    if world == 1 and not FORCE_COLLECTIVES:
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
        if dist.get_world_size(ctx.group) > 1 or FORCE_COLLECTIVES:
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


def _held():
    """csrc.features_held() where the operator layer is loaded (a rank's cameras share one feature state), else nothing."""
    import contextlib
    import sys
    mod = sys.modules.get("svox_t_amd.csrc")
    return mod.features_held() if mod is not None and hasattr(mod, "features_held") else contextlib.nullcontext()


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
        with torch.enable_grad(), _held():
            # (one feature state for all of this rank's cameras: its sigma bitmask / exponentials table are built once)
            feats = features.detach().requires_grad_(True)
            local = [render_fn(feats, c2ws[i]) for i in mine]
        ctx.feats, ctx.local, ctx.group, ctx.mine = feats, local, group, mine
        if world == 1 and not FORCE_COLLECTIVES:
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
        if dist.get_world_size(ctx.group) > 1 or FORCE_COLLECTIVES:
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


def _peer(dist_module, group, r: int) -> int:
    return dist_module.get_global_rank(group, r) if group is not None else r


def _exchange(dist_module, sends, recvs, group):
    """One round of point-to-point traffic with every peer at once: sends / recvs = [(tensor, group rank)].  On the
    "nccl" backend (RCCL) the batch is one grouped launch -- every xGMI link of the mesh carries its pair's bytes at the
    same time, which is what an all-to-all is on this fabric; gloo runs the same calls for the CPU tests."""
    ops = []
    for t, r in sends:
        if t.numel():
            ops.append(dist_module.P2POp(dist_module.isend, t, _peer(dist_module, group, r), group))
    for t, r in recvs:
        if t.numel():
            ops.append(dist_module.P2POp(dist_module.irecv, t, _peer(dist_module, group, r), group))
    if ops:
        for w in dist_module.batch_isend_irecv(ops):
            w.wait()


def direct_all_reduce(dist_module, grad: torch.Tensor, group=None) -> torch.Tensor:
    """all-reduce(sum) of a row-major gradient [M, ...] in place, built for a point-to-point mesh (xGMI: a link per
    pair of GPUs, no switch) instead of a ring:

        reduce-scatter   rank r owns the contiguous row range shard_bounds(M, W, r); every rank sends each peer that
                         peer's range of its local gradient -- W - 1 transfers of S / W bytes, one per link, all at once
                         -- and adds the W - 1 pieces it receives to its own range;
        all-gather       every rank sends its summed range to each peer, again one transfer of S / W per link.

    Per link and direction 2 S / W bytes move in two rounds: 2 S / (W B) seconds at B bytes/s per link, against
    2 (W - 1) S / (W B) for a ring that uses one link per step (NOTEBOOK.md 7 has the table).  No padding, no staging copy of
    the gradient: ranges are views, ragged splits are fine.  Adds happen in rank order on the owner (own, then peers
    ascending): the result is the same on every rank, bit for bit."""
    W, me = dist_module.get_world_size(group), dist_module.get_rank(group)
    if (W == 1 and not FORCE_COLLECTIVES) or grad.numel() == 0:
        return grad
    assert grad.is_contiguous()
    M = grad.shape[0]
    bounds = [shard_bounds(M, W, r) for r in range(W)]
    lo, hi = bounds[me]
    peers = [r for r in range(W) if r != me]
    recv = grad.new_empty((len(peers), hi - lo) + tuple(grad.shape[1:]))
    _exchange(dist_module, [(grad[bounds[r][0]:bounds[r][1]], r) for r in peers], [(recv[j], r) for j, r in enumerate(peers)], group)
    own = grad[lo:hi]
    for j in range(len(peers)):
        own += recv[j]
    _exchange(dist_module, [(own, r) for r in peers], [(grad[bounds[r][0]:bounds[r][1]], r) for r in peers], group)
    return grad


def touched_blocks(grad: torch.Tensor, block_rows: int) -> torch.Tensor:
    """uint8 [ceil(M / block_rows)]: 1 where a block of rows holds a non-zero entry.  (A row whose contributions cancel to
    exactly zero counts as untouched: leaving out a zero changes no sum.)"""
    M = grad.shape[0]
    nblk = (M + block_rows - 1) // block_rows
    flat = grad.reshape(M, -1)
    full = M // block_rows
    mask = torch.zeros((nblk,), dtype=torch.uint8, device=grad.device)
    if full:
        mask[:full] = (flat[:full * block_rows].reshape(full, -1) != 0).any(dim=1)
    if full < nblk:
        mask[full] = (flat[full * block_rows:] != 0).any()
    return mask


def sparse_all_reduce(dist_module, grad: torch.Tensor, group=None, block_rows: int = 256, mask: Optional[torch.Tensor] = None):
    """all-reduce(sum) of grad [M, K] in place, moving only the blocks of `block_rows` rows that somebody touched.

    A camera's backward touches the feature rows its rays cross; rows no rank touched are zero everywhere and need not
    travel at all, rows one rank touched need no reduction.  Same two rounds as direct_all_reduce, over blocks:
        0  all-gather of the ranks' block masks (ceil(M / block_rows) bytes each), read on the host: sizes are known
           to both ends of every transfer (one host synchronisation per exchange: tens of microseconds, which is why
           this form is for gradients of tens of MB and up);
        1  every rank sends each owner the blocks of that owner's range it touched, compacted; the owner adds them in;
        2  every owner sends each peer the blocks of its range that ANYONE touched.
    Bytes per link: (what the rank touched of a range) one way, (the union over ranks of a range) the other, instead
    of S / W each way.  Returns (grad, stats) with stats = blocks sent in round 1 / round 2 / a dense exchange's.
    The adds are in rank order on the owner, as in direct_all_reduce: equal to it as floats (==; a row whose only
    contributions are -0.0 keeps its sign here, where the dense exchange's + 0.0 turns it into +0.0)."""
    W, me = dist_module.get_world_size(group), dist_module.get_rank(group)
    if (W == 1 and not FORCE_COLLECTIVES) or grad.numel() == 0:
        return grad, {"round1_blocks": 0, "round2_blocks": 0, "dense_blocks": 0}
    assert grad.is_contiguous() and grad.dim() == 2
    M, K = grad.shape
    nblk = (M + block_rows - 1) // block_rows
    if mask is None:
        mask = touched_blocks(grad, block_rows)
    masks = mask.new_empty((W, nblk))
    if hasattr(dist_module, "all_gather_into_tensor") and grad.is_cuda:
        dist_module.all_gather_into_tensor(masks, mask, group=group)
    else:
        dist_module.all_gather(list(masks.unbind(0)), mask, group=group)
    masks_h = masks.cpu().bool()                                   # the one host synchronisation
    union = masks_h.any(dim=0)
    bb = [shard_bounds(nblk, W, r) for r in range(W)]               # owner ranges, in blocks
    peers = [r for r in range(W) if r != me]
    full = M // block_rows                                          # the last block may be short: it travels padded

    def blocks_view(t):
        """[full, block_rows, K] view of the whole blocks (+ the short last block handled by pad / unpad below)."""
        return t[:full * block_rows].view(full, block_rows, K)

    tail = None
    if full < nblk:                                                 # a padded copy of the short last block
        tail = grad.new_zeros((block_rows, K))
        tail[:M - full * block_rows] = grad[full * block_rows:]

    def gather(ids):
        """the blocks `ids` (a LongTensor on the host) as one contiguous [n, block_rows, K] buffer"""
        ids = ids.to(grad.device)
        out = grad.new_empty((ids.numel(), block_rows, K))
        whole = ids < full
        if whole.any():
            out[whole] = blocks_view(grad)[ids[whole]]
        if (~whole).any():
            out[~whole] = tail
        return out

    def scatter(ids, buf, add):
        nonlocal tail
        ids = ids.to(grad.device)
        whole = ids < full
        if whole.any():
            if add:
                blocks_view(grad).index_add_(0, ids[whole], buf[whole])
            else:
                blocks_view(grad).index_copy_(0, ids[whole], buf[whole])
        if (~whole).any():
            if add:
                tail += buf[~whole][0]
            else:
                tail = buf[~whole][0].clone()

    idx = torch.arange(nblk)
    # round 1: my touched blocks of each owner's range -> that owner; what the peers touched of mine <- them
    send_ids = {r: idx[bb[r][0]:bb[r][1]][masks_h[me, bb[r][0]:bb[r][1]]] for r in peers}
    recv_ids = {r: idx[bb[me][0]:bb[me][1]][masks_h[r, bb[me][0]:bb[me][1]]] for r in peers}
    sbuf = {r: gather(send_ids[r]) for r in peers}
    rbuf = {r: grad.new_empty((recv_ids[r].numel(), block_rows, K)) for r in peers}
    _exchange(dist_module, [(sbuf[r], r) for r in peers], [(rbuf[r], r) for r in peers], group)
    for r in peers:                                                 # rank order: the same sums on every owner
        if recv_ids[r].numel():
            scatter(recv_ids[r], rbuf[r], add=True)
    # round 2: the union-touched blocks of my range -> everyone; theirs <- them
    mine = idx[bb[me][0]:bb[me][1]][union[bb[me][0]:bb[me][1]]]
    own = gather(mine)
    theirs = {r: idx[bb[r][0]:bb[r][1]][union[bb[r][0]:bb[r][1]]] for r in peers}
    rbuf2 = {r: grad.new_empty((theirs[r].numel(), block_rows, K)) for r in peers}
    _exchange(dist_module, [(own, r) for r in peers], [(rbuf2[r], r) for r in peers], group)
    for r in peers:
        if theirs[r].numel():
            scatter(theirs[r], rbuf2[r], add=False)
    if tail is not None:
        grad[full * block_rows:] = tail[:M - full * block_rows]
    stats = {"round1_blocks": int(sum(v.numel() for v in send_ids.values())),
             "round2_blocks": int(mine.numel()) * len(peers),
             "dense_blocks": int(sum(bb[r][1] - bb[r][0] for r in peers)) + (bb[me][1] - bb[me][0]) * len(peers)}
    return grad, stats


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

    def __init__(self, dist_module=dist, group=None, backend: str = "nccl", chunk_bytes: int = 32 << 20,
                 mode: str = "all_reduce"):
        """mode: "all_reduce" -- the backend's own all-reduce per row chunk (RCCL chooses ring / tree and its channels);
        "direct" -- direct_all_reduce: reduce-scatter and all-gather as two rounds of simultaneous point-to-point
        transfers, one per link of the mesh (NOTEBOOK.md 7: the form priced for xGMI); "touched" -- sparse_all_reduce
        (only row blocks somebody touched; one host synchronisation per exchange, so start() then blocks the host)."""
        assert mode in ("all_reduce", "direct", "touched")
        self.dist, self.group, self.backend, self.mode = dist_module, group, backend, mode
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
        if grad is None or grad.numel() == 0 or (self.dist.get_world_size(self.group) == 1 and not FORCE_COLLECTIVES):
            return
        assert grad.is_contiguous()
        self._grad = grad                                   # stays alive while it travels

        def issue():
            if self.mode == "direct":
                direct_all_reduce(self.dist, grad, self.group)     # (its waits are stream waits under "nccl": the host goes on)
                return []
            if self.mode == "touched":
                sparse_all_reduce(self.dist, grad, self.group)
                return []
            return [self.dist.all_reduce(grad[lo:hi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
                    for lo, hi in self.chunks(grad)]
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(grad.device))
            grad.record_stream(self._side)
            with torch.cuda.stream(self._side):
                self._works = issue()
                self._side_done = torch.cuda.Event()
                self._side_done.record(self._side)
        else:
            self._works = issue()

    def wait(self) -> Optional[torch.Tensor]:
        """The reduced gradient handed to the last start() (None if there was none)."""
        for w in self._works:
            w.wait()                                        # nccl: the current stream waits; gloo: the host does
        self._works = []
        ev = getattr(self, "_side_done", None)
        if ev is not None and self._grad is not None:       # (the direct / touched exchanges end with kernels of the side stream)
            torch.cuda.current_stream(self._grad.device).wait_event(ev)
            self._side_done = None
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
    if world == 1 and not FORCE_COLLECTIVES:
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
