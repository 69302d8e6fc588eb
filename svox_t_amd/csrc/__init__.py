"""`svox_t_amd.csrc` -- the operator boundary, shaped like the reference's
pybind11 module `svox_t.csrc` (svox_t/csrc/svox.cpp:73-145) so that the
`autograd.Function`s and `N3Tree._spec` that sit on top of it read the same.

Same class names (`RaysSpec`, `TreeSpec`, `CameraSpec`, `RenderOptions`), same
attribute names, same function names and argument order for the hot path:

    volume_render(tree, rays, opt)                    -> Tensor [Q, C+1]
    volume_render_backward(tree, rays, opt, grad)     -> Tensor [M, K]
    render_depth(tree, rays, opt)                     -> Tensor [Q, 1]
    opacity_render(tree, rays, opt)                   -> Tensor [Q, 1]
    opacity_render_backward(tree, rays, opt, grad)    -> Tensor [M, K]
    query_vertical(tree, indices)                     -> (values, node_ids, data_ids, leaf_node)
    query_vertical_backward(tree, indices, grad)      -> Tensor [M, K]

Underneath there is no pybind: tensors are marshalled to raw device pointers
and handed to the C ABI of libsvoxt_hip.so (include/svoxt.h) through ctypes,
on torch's *current* stream (the reference launches on the legacy default
stream, rt_kernel.cu:1373).  Outputs are allocated by torch's caching
allocator.  Precondition failures raise RuntimeError like the reference's
TORCH_CHECKs (data_spec.hpp:38-43); unlike the reference, launch errors are
raised too.

There is NO CPU fallback here: if the shared library is missing the import
fails, and non-GPU tensors are rejected.
"""
from __future__ import annotations

import ctypes
import os
import weakref

import torch

from ._abi import (ABI_VERSION, EXPORTS, FORMAT_ASG, FORMAT_RGBA, FORMAT_SG, FORMAT_SH, LIB_PATH, LISTS_BEGUN, LISTS_FWD_AGENT_FENCE,
                   LISTS_FWD_NO_OVERLAP, LISTS_FWD_ONE_KERNEL, LISTS_FWD_TWO_KERNELS, LISTS_GRAD_ZEROED, LISTS_NATIVE_MATH,
                   LISTS_TEST_DROP, LISTS_TEST_NOPOLL, LISTS_TEST_STALE, _CLists, _CMotion, _COptions, _CRays, _CTree, _lib)
from ._marshal import (ACCEL_BRICKS, ACCEL_LOG2, CameraSpec, RaysSpec, RenderOptions, TreeSpec, _ACCEL_CACHE, _accel_for, _accel_log2_for,
                       _call, _check_input, _numel, _pack_camera, _pack_opts, _pack_rays, _pack_tree, _pack_tree_accel, _ptr,
                       _on, _stream, get_out_data_dim)


# ---------------------------------------------------------------------------
# Switches.  Five environment variables, read HERE once at import (the library itself reads none): SVOXT_LIB
# (_abi.py: another build of the same ABI), SVOXT_ACCEL_LOG2 (_marshal.py), SVOXT_SORT_RAYS, and the two TOLERANCE modes
# SVOXT_BWD_EXACT / SVOXT_NATIVE_MATH.  Everything else below is a module attribute that tests assign to take a
# route the defaults do not (a cross-check of one kernel family against another: all result-neutral) -- r05: they were
# environment switches too until then, fifteen in all.  README.md has the table.
# ---------------------------------------------------------------------------
def _env_flag(name: str, default: str) -> bool:
    return os.environ.get(name, default) not in ("", "0")


# SVOXT_SORT_RAYS: "auto" (batches that are not images are rendered in svoxt_ray_order's order from
# SORT_RAYS_MIN rays on), "0" never, "1" always; RaysSpec.sort per call
SORT_RAYS = os.environ.get("SVOXT_SORT_RAYS", "auto")
SORT_RAYS_MIN = 16384
# --- the two TOLERANCE modes (not result-neutral; both off by default; NOTEBOOK.md 4)
# SVOXT_BWD_EXACT=0: the single-march backward -- accum = sum_c g_c * out_c from the forward's output, one
# sweep over the lists instead of two: equal up to the rounding of that sum, which moves 36 % of the
# sigma-column entries by more than 1e-5 of their own value (tests/test_gpu_query_and_misc.py)
BWD_EXACT = _env_flag("SVOXT_BWD_EXACT", "1")
# SVOXT_NATIVE_MATH=1: RGBA-style rows of 8 / 16 / 32 floats are SHADED with the hardware's exponential and
# reciprocal (forward and per-tile backward; include/svoxt.h SVOXT_LISTS_NATIVE_MATH); the stepping stays
# bit-exact.  Outputs within 1e-5 relative, gradients within 1e-5 of the tight scale (tested at full size).
NATIVE_MATH = _env_flag("SVOXT_NATIVE_MATH", "0")
# --- test attributes (result-neutral routes; no environment variable)
SIGMA_MASK = True         # False: the two-kernel forward's march gathers sigma instead of reading a bit per feature row
BWD_LIST_SAMPLES = 96     # samples kept per ray for the backward (8 bytes each; longer rays march the rest); 0: never record
FWD_LIST_SAMPLES = 96     # the same for the scratch lists of a forward nobody differentiates; 0: always the one-kernel forward
LIST_POOL = True          # False: dense sample lists (every ray owns `cap` slots) instead of 4 KB blocks from a pool
AUTO_PLAN = True          # False: plain volume_render / volume_render_backward calls hand nothing from forward to backward
FWD_SPLIT = ""            # "0" / "1": the forward as one kernel / as march + shade wherever both exist ("": the library's default)
BWD_GATHER = 1            # 0: always the per-ray backward; 1: the per-tile one for images / sorted batches; 2: whenever the payload allows
FWD_OVERLAP = True        # False: march and shade of the two-kernel forward as two launches instead of fwd_roles_kernel
EXP_TABLE = True          # False: rows of 8 / 16 / 32 floats without the per-row exponentials table (svoxt_tree.exp_table)
BWD_TERMS = True          # False: no hand-over between the sweeps of the exact backwards (every row gathered twice)
BWD_FUSED = True          # False: list walk and per-tile merge of an image's backward as two kernels
BWD_XF_FUSED = True       # False: view rotations (transformation_matrices) through the two-kernel form they took until r04
ROLES_FLAGS = 0           # OR of LISTS_TEST_* / LISTS_FWD_AGENT_FENCE: test and measurement switches of the one-launch forward


def _list_flags(native: bool = False) -> int:
    f = LISTS_NATIVE_MATH if native else 0
    if FWD_SPLIT == "0":
        f |= LISTS_FWD_ONE_KERNEL
    elif FWD_SPLIT not in ("", "0"):
        f |= LISTS_FWD_TWO_KERNELS
    if not FWD_OVERLAP:
        f |= LISTS_FWD_NO_OVERLAP
    return f | ROLES_FLAGS


# ---------------------------------------------------------------------------
# Sigma bitmask (include/svoxt.h, svoxt_sigma_mask_build): one bit per feature row, for the march of
# the two-kernel forward.  Derived from the CONTENT of `features`, which torch's version counter does not
# vouch for: `param.data.add_(...)` and `tree.features.data[key] = v` -- the reference's own idiom --
# leave it unchanged.  The mask is therefore built afresh by EVERY forward that uses it
# (M / 8 bytes written, one line of the table read per row: 0.012 ms at depth 8, 0.10 ms for the 578 MB
# table of depth 9), inside whatever the caller times -- unless the caller has declared the table static
# (TreeSpec.static_features, set by N3Tree.static_features = True: an inference server's frozen tree):
# then it is cached on the tensor object, its version counter and its data pointer like the
# acceleration grid.  SIGMA_MASK False: never.  (r02: depth 9 / 578 MB of features,
# forward 1.29 -> 1.05 ms; depth 8 / 66 MB, where the gather still hits the Infinity Cache,
# forward+backward 955 -> 972 Mrays/s with the rebuild inside the step.)
_SIGMA_CACHE: dict = {}
_FEATURES_HELD = [0]      # > 0: inside features_held()


class features_held:
    """`with features_held():` -- the caller promises that no feature tensor changes inside the block (several cameras
    rendered from one feature state: parallel.render_cameras).  What this module derives from the CONTENT of a feature
    tensor -- the sigma bitmask, the exponentials table of rows of 8 / 16 / 32 floats -- is then built once per tensor
    instead of once per forward (eight cameras at 1024 x 1024 / depth 9 / K = 32: seven passes of 0.18 ms and seven
    tables of 578 MiB less); everything built inside is dropped when the outermost block ends."""

    def __enter__(self):
        _FEATURES_HELD[0] += 1
        return self

    def __exit__(self, *exc):
        _FEATURES_HELD[0] -= 1
        if _FEATURES_HELD[0] == 0:
            for k in [k for k, v in _SIGMA_CACHE.items() if len(v) > 5 and v[5]]:
                _SIGMA_CACHE.pop(k, None)
        return False


def _attach_sigma_mask(tree: TreeSpec, ct: _CTree, thresh: float, keep: bool, table: bool = False, begin=None):
    """ct.sigma_mask <- the bitmask of this feature content; table: also ct.exp_table <- the rows' exponentials (rows of
    8 / 16 / 32 floats), built in the same pass.  Returns the table tensor (or None): a recording forward keeps it on its
    lists for the backward of the same feature content.  begin: the SampleLists of the recording forward that follows --
    where the mask is built here and now, the same launch leaves their tables in the state that forward starts from
    (svoxt_sigma_mask_build_fill) and the lists say so (LISTS_BEGUN)."""
    f = tree.features
    if not SIGMA_MASK or ct.M == 0:
        return None
    table = bool(table and EXP_TABLE and f.dim() == 2 and f.is_contiguous() and f.dtype == torch.float32)
    key = id(f)
    mask = etab = None
    held = _FEATURES_HELD[0] > 0
    keep = (keep and bool(getattr(tree, "static_features", False))) or held
    if keep:
        ent = _SIGMA_CACHE.get(key)
        if ent is not None and ent[0]() is f and ent[1] == (f._version, f.data_ptr()) and ent[2] == thresh \
                and (ent[4] is not None or not table):
            mask, etab = ent[3], (ent[4] if table else None)
    if mask is None:
        dev = f.device
        with _on(dev):
            mask = torch.empty((_lib.svoxt_sigma_mask_bytes(ct.M) // 8,), dtype=torch.int64, device=dev)
            if table:
                etab = torch.empty_like(f, requires_grad=False)
                _call("svoxt_exp_table_build", ctypes.byref(ct), ctypes.c_float(thresh), _ptr(mask), _ptr(etab), _stream(dev))
            elif begin is not None and begin.both is not None:
                _call("svoxt_sigma_mask_build_fill", ctypes.byref(ct), ctypes.c_float(thresh), _ptr(mask), _ptr(begin.both),
                      begin.both.numel() * 4, _stream(dev))
                begin.begun = True
            else:
                _call("svoxt_sigma_mask_build", ctypes.byref(ct), ctypes.c_float(thresh), _ptr(mask), _stream(dev))
        if keep:
            _SIGMA_CACHE[key] = (weakref.ref(f, lambda _r, _k=key: _SIGMA_CACHE.pop(_k, None)),
                                 (f._version, f.data_ptr()), thresh, mask, etab, held)
    ct.sigma_mask = mask.data_ptr()
    ct.sigma_mask_thresh = thresh
    ct._keepalive_mask = mask
    if etab is not None:
        ct.exp_table = etab.data_ptr()
        ct._keepalive_etab = etab
    return etab


# GRAD_SCRATCH False (test attribute): every backward over sample lists allocates and fills its padded gradient buffer itself.
# Default: the padded [M, stride] buffer the atomics go to is kept between steps (per device, stream and shape)
# and svoxt_compact_rows_clear leaves it zeroed while it produces the dense gradient: one fill of M * stride
# floats per step less; result-neutral
GRAD_SCRATCH = True
_GRAD_SCRATCH: dict = {}      # (device, stream, M, stride) -> [buffer, known to be all zeros]


def _grad_scratch(dev, M: int, stride: int):
    key = (dev.index or 0, torch.cuda.current_stream(dev).cuda_stream, M, stride)
    ent = _GRAD_SCRATCH.get(key)
    if ent is None:
        if len(_GRAD_SCRATCH) >= 2:
            _GRAD_SCRATCH.clear()                     # (a tree that is refined changes M: no pile of old shapes)
        ent = _GRAD_SCRATCH[key] = [torch.zeros((M, stride), dtype=torch.float32, device=dev), True]
    elif not ent[1]:
        ent[0].zero_()                                # a call that failed half way left it in an unknown state
    ent[1] = False                                    # ... until svoxt_compact_rows_clear has been enqueued
    return ent


def invalidate_caches(*tensors) -> None:
    """Drop what this module derived from the CONTENT of the given tensors (child / data: the acceleration
    grid; features: a cached sigma bitmask) and bump their torch version counters.  For writers that bypass
    the counters: `t.data[...] = v`, raw-pointer kernels, collectives into `.data`.  No arguments: everything."""
    if not tensors:
        _ACCEL_CACHE.clear()
        _SIGMA_CACHE.clear()
        _GRAD_SCRATCH.clear()
        return
    ids = {id(t) for t in tensors if isinstance(t, torch.Tensor)}
    for k in [k for k, ent in _ACCEL_CACHE.items() if k[0] in ids or id(ent[2]()) in ids]:
        _ACCEL_CACHE.pop(k, None)
    for k in ids:
        _SIGMA_CACHE.pop(k, None)
    for t in tensors:
        if isinstance(t, torch.Tensor):
            torch.autograd.graph.increment_version(t)


# ---------------------------------------------------------------------------
# Hot-path operators
# ---------------------------------------------------------------------------

# Pooled lists (include/svoxt.h, svoxt_sample_lists.blocktab): the records live in 4 KB blocks handed
# out per (tile, 8 list positions) from a pool, so memory follows the samples that exist instead of
# cap x rays (800x800, depth-8 SH9: 47 MB of records in a 491 MB dense buffer).  How large a pool a
# batch needs is learned from the forwards before it: now and then ("a look") the 32 block counters are
# copied to pinned host memory without waiting -- ONE 2 KB copy, no kernel -- and read when the copy has
# landed.  A ray that finds no block stops recording and marches the rest, so a pool that is too small
# costs time, never correctness.  LIST_POOL False: dense lists.
#   The size is STICKY (r05): it changes only when a look leaves the band
#   [POOL_LOW, POOL_HIGH] of the current capacity -- above (or dry): grow at once; below: shrink after
#   POOL_SHRINK_LOOKS such looks in a row -- and sizes are rounded up to 1/16 of a power of two, so the
#   steady state of a training loop asks the allocator for the same block sizes step after step (r04 sized
#   to 1.25x the last eight looks: a new size at almost every look, a new cudaMalloc behind it).
#   The first two forwards of a shape are looked at right away and set the size directly.
#   freeze_pools(True): no look is started and none is applied (bench.py around its timed steps).
# The hint is kept per KIND of forward (recording / scratch: a no-grad forward between training steps of the
# same shape needs a different pool: scripts/persp_timing.py).
_POOL_HINT: dict = {}       # (tiles, S, kind) -> [blocks, pending (event, pinned counters, capacity) or None, forwards seen, ran dry, low looks]
POOL_LOW, POOL_HIGH, POOL_SHRINK_LOOKS, POOL_LOOK_EVERY = 0.5, 0.85, 4, 16
_POOL_FROZEN = [False]


def freeze_pools(on: bool = True) -> None:
    """While on, the pooled lists keep the sizes they have: no usage look is started, none is applied."""
    _POOL_FROZEN[0] = bool(on)


def _pool_quant(n: int) -> int:
    """n rounded up to a multiple of 1/16 of the power of two below it (at least 32)."""
    q = max(32, 1 << max(0, int(n).bit_length() - 5))
    return (int(n) + q - 1) // q * q


def _pool_blocks_for(tiles: int, S: int, kind: str = "record") -> int:
    full = tiles * (S // 8)
    ent = _POOL_HINT.setdefault((tiles, S, kind), [min(full, tiles * 12), None, 0, False, 0])
    if _POOL_FROZEN[0]:
        return max(1, ent[0])
    # (a pool known to have run dry: the look that is under way is waited for -- a dry pool costs more than the wait)
    if ent[1] is not None and (ent[1][0].query() or ent[3]):
        if ent[3]:
            ent[1][0].synchronize()
        used = (int(ent[1][1].view(32, 16)[:, 0].max().item()) + 1) * 32        # the fullest of the 32 parts sets the need
        cap = ent[1][2]
        ent[1] = None
        ent[3] = used >= cap
        want = min(full, max(tiles, _pool_quant(int(used * 1.25) + 64)))
        if ent[3]:
            ent[0], ent[4] = min(full, cap * 2), 0
        elif ent[2] <= 2 or used > POOL_HIGH * cap:
            ent[0], ent[4] = (want if (ent[2] <= 2 or want > cap) else cap), 0
        elif used < POOL_LOW * cap:
            ent[4] += 1
            if ent[4] >= POOL_SHRINK_LOOKS:
                ent[0], ent[4] = want, 0
        else:
            ent[4] = 0
    return max(1, ent[0])


_PINNED: list = []


def _pinned_counters():
    """A pinned int32[512] (a small ring: pinned allocations are slow to make)."""
    if len(_PINNED) < 8:
        _PINNED.append(torch.empty((512,), dtype=torch.int32, pin_memory=True))
        return _PINNED[-1]
    _PINNED.append(_PINNED.pop(0))
    return _PINNED[-1]


class SampleLists:
    """Per-ray lists of composited samples recorded by a forward (see
    include/svoxt.h, svoxt_sample_lists).  Opaque to callers: pass it back to
    volume_render_backward."""

    def __init__(self, Q, S, device, pooled=None, kind="record"):
        S = (S + 7) // 8 * 8                                 # whole 64-byte lines of 8 records per lane
        tiles = (Q + 63) // 64
        self.pooled = (LIST_POOL if pooled is None else bool(pooled)) and S <= 512
        self.tiles = tiles
        self.kind = kind
        if self.pooled:
            self.pool_blocks = (_pool_blocks_for(tiles, S, kind) + 31) // 32 * 32      # 32 equal parts, a counter each
            nt = (tiles * (S // 8) + 1) // 2 * 2      # (even: tile_state behind it holds 64-bit queue entries -- 8-byte aligned)
            # table, then the 32 counters, then the tile states, ready queues and their counters (march and shade in one launch): one fill
            # (rounded up to whole 16-byte words: svoxt_sigma_mask_build_fill can then do lists_begin's fill)
            both = torch.empty(((nt + 32 * 16 + 17 * tiles + 514 + 3) // 4 * 4,), dtype=torch.int32, device=device)
            self.blocktab, self.pool_next, self.tile_state = both[:nt], both[nt:nt + 32 * 16], both[nt + 32 * 16:]
            self.both = both
        else:
            # rec[tile][block][lane][8]: every ray owns S slots
            self.pool_blocks = tiles * (S // 8)
            self.blocktab = self.pool_next = self.tile_state = self.both = None
        self.begun = False      # the tables were filled by the launch that built the sigma bitmask (LISTS_BEGUN)
        self.rec = torch.empty((self.pool_blocks * 512, 2), dtype=torch.int32, device=device)
        self.aux = torch.empty((Q, 4), dtype=torch.int32, device=device)
        self.S = S
        self.coef = None        # allocated by the backward when it takes the two-kernel route,
        self.consumed = False   # which rewrites `rec`: the lists then serve no second backward
        self.terms = None       # (att, e0, e1, e2) per record slot for the exact one-kernel backward
        self.terms_state = 0    # 2: filled by the forward; 0: scratch (the backward's first sweep fills it)
        self.exp_table = None   # rows of 8 / 16 / 32 floats: the exponentials table the forward built (svoxt_tree.exp_table)
        self.flags = 0          # svoxt_sample_lists.flags: how the kernels that write / read these lists work

    def note_usage(self):
        """After the forward that filled the lists was enqueued: remember how much of the pool it took
        (read later, without waiting)."""
        if not self.pooled:
            return
        ent = _POOL_HINT.get((self.tiles, self.S, self.kind))
        if ent is None or ent[1] is not None or _POOL_FROZEN[0]:
            return
        ent[2] += 1
        if ent[2] > 2 and ent[2] % POOL_LOOK_EVERY and not ent[3]:   # the first forwards of a shape, then every 16th (every one while the pool runs dry)
            return
        host = _pinned_counters()
        host.copy_(self.pool_next, non_blocking=True)             # one 2 KB copy: the 32 counters, 64 bytes apart
        ev = torch.cuda.Event()
        ev.record()
        ent[1] = (ev, host, self.pool_blocks)

    def c_struct(self):
        return _CLists(self.rec.data_ptr(), self.aux.data_ptr(), self.S,
                       None if self.coef is None else self.coef.data_ptr(),
                       0 if self.coef is None else self.coef.numel() * 4,
                       None if self.terms is None else self.terms.data_ptr(),
                       0 if self.terms is None else self.terms.numel() * 4,
                       None if self.blocktab is None else self.blocktab.data_ptr(),
                       self.pool_blocks if self.pooled else 0,
                       None if self.pool_next is None else self.pool_next.data_ptr(),
                       self.terms_state, self.flags | (LISTS_BEGUN if self.begun else 0),
                       None if self.tile_state is None else self.tile_state.data_ptr())


def _list_cap(ct: "_CTree", base: int) -> int:
    """Records a single ray may list.  Pooled lists make a high cap free (memory follows the samples
    that exist): 192 instead of 96, so that the rays of a depth-9 tree that composite up to ~190
    samples need no tail launch -- except with view rotations, whose two-kernel backward sizes a
    second buffer by the cap."""
    if LIST_POOL and base == 96 and ct.xform is None:
        return 192
    return base


def can_record(tree: TreeSpec, opt: RenderOptions) -> bool:
    """True when volume_render(..., record=True) can hand sample lists to the backward
    (a specialised payload, all components; any thresholds since r05: include/svoxt.h svoxt_can_record)."""
    ct, co = _pack_tree(tree), _pack_opts(opt)
    return bool(_lib.svoxt_can_record(ctypes.byref(ct), ctypes.byref(co)))


def ray_order(tree: TreeSpec, rays: RaysSpec, opt: RenderOptions) -> torch.Tensor:
    """Permutation (int64 [Q], for torch indexing) that sorts a ray batch by the Morton code of
    each ray's entry point into the tree's cube (include/svoxt.h, svoxt_ray_order): 64
    consecutive rays of the sorted batch cross the same leaves.  Not in the reference."""
    return _ray_order32(tree, rays, opt).long()


def _ray_order32(tree: TreeSpec, rays: RaysSpec, opt: RenderOptions) -> torch.Tensor:
    """... as the library writes it (int32), for svoxt_gather_rays / svoxt_permute_rows."""
    ct, cr, co = _pack_tree(tree), _pack_rays(rays), _pack_opts(opt)
    dev = tree.features.device
    with _on(dev):
        perm = torch.empty((cr.Q,), dtype=torch.int32, device=dev)
        nbytes = _lib.svoxt_ray_order_workspace_bytes(cr.Q)
        if nbytes < 0:
            raise RuntimeError("svoxt_ray_order_workspace_bytes failed")
        ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
        _call("svoxt_ray_order", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co), _ptr(perm), _ptr(ws),
              nbytes, _stream(dev))
    return perm


def _permute_rows(src: torch.Tensor, perm32: torch.Tensor, scatter: bool) -> torch.Tensor:
    """dst[i] = src[perm[i]] (scatter False) or dst[perm[i]] = src[i] over rows of a float32 [Q, cols] tensor
    (the library's kernel: torch's index kernels take 0.136 ms for 640 000 16-byte rows on this target)."""
    if src.dtype != torch.float32 or src.dim() != 2 or not src.is_contiguous():
        src = src.contiguous().float()
    dev = src.device
    with _on(dev):
        dst = torch.empty_like(src)
        _call("svoxt_permute_rows", _ptr(src), _ptr(perm32), _ptr(dst), src.shape[0], src.shape[1], 1 if scatter else 0,
              _stream(dev))
    return dst


# ---------------------------------------------------------------------------
# What a plain caller gets.  The reference's autograd functions call
#     out  = _C.volume_render(tree, rays, opt)                         (renderer.py:63)
#     grad = _C.volume_render_backward(tree, rays, opt, grad_out)      (renderer.py:72)
# with the SAME spec objects (they sit on the ctx, renderer.py:64-66) and nothing else.  Everything
# this implementation adds lives below these two calls: the forward of a differentiable feature
# table records the sample lists, a batch that is not an image is rendered in svoxt_ray_order's
# order, and both are left on the rays spec object for the backward that arrives with it -- which
# then replays the lists (per-tile, one kernel) instead of marching.  A backward whose forward left
# nothing (another spec object, features changed in place since, AUTO_PLAN False) marches, as
# the reference does.  The explicit forms (record=True / lists= / fwd_output=) remain.
# ---------------------------------------------------------------------------
# Batches that are not declared images are rendered in a coherent order (svoxt_ray_order,
# include/svoxt.h: sort by the rays' entry points into the cube, three gathers, a scatter --
# ~0.3 ms per 640 000 rays) from SORT_RAYS_MIN rays on, and their backward then takes the
# per-tile route: 640 000 rays forward+backward, shuffled within one camera 1.47 -> 0.93 ms,
# drawn from 8 cameras 1.48 -> 1.29 ms, row-major but not declared an image 1.29 -> 0.92 ms
# (profiles/r01_s_ray_order_timing.txt).  SORT_RAYS (SVOXT_SORT_RAYS) "0" never, "1" always; RaysSpec.sort per call.


class _Plan:
    """What a forward leaves on its rays / camera spec for the backward of the same call."""
    __slots__ = ("kind", "features", "fkey", "tkey", "rkey", "optkey", "lists", "out", "over", "perm", "rays", "__weakref__")


def _opt_key(opt):
    return (float(opt.step_size), float(opt.background_brightness), int(opt.format), int(opt.basis_dim),
            int(opt.ndc_width), int(opt.ndc_height), float(opt.ndc_focal), int(opt.min_comp), int(opt.max_comp),
            float(opt.sigma_thresh), float(opt.stop_thresh))


def _tkey(t):
    """Identity of a tensor's content as far as torch can vouch for it: storage address, version, shape."""
    if not isinstance(t, torch.Tensor) or t.numel() == 0:
        return None
    return (t.data_ptr(), t._version, tuple(t.shape))


def _tree_key(tree):
    """What the recorded lists depend on besides the features: the topology and the view rotations."""
    return (_tkey(tree.child), _tkey(tree.data), int(tree.n_internal or 0), _tkey(tree.offset), _tkey(tree.scaling),
            _tkey(tree.transformation_matrices))


def _rays_key(rays):
    """... and the rays: the reference's backward re-marches whatever the spec holds NOW (renderer.py:64-72),
    so lists recorded for other ray values must not be replayed."""
    if isinstance(rays, CameraSpec):
        return ("cam", _tkey(rays.c2w), float(rays.fx), float(rays.fy), int(rays.width), int(rays.height))
    return ("rays", _tkey(rays.origins), _tkey(rays.dirs), _tkey(rays.vdirs),
            int(getattr(rays, "image_width", 0) or 0), int(getattr(rays, "image_height", 0) or 0),
            _tkey(getattr(rays, "order", None)))


def _wants_sort(rays) -> bool:
    if isinstance(rays, CameraSpec) or not isinstance(rays.origins, torch.Tensor) or rays.origins.dim() != 2:
        return False                        # (a malformed spec: _pack_rays will say what is wrong)
    Q = rays.origins.shape[0]
    w, h = int(getattr(rays, "image_width", 0) or 0), int(getattr(rays, "image_height", 0) or 0)
    if Q == 0 or (w * h == Q and w % 8 == 0 and h % 8 == 0) or getattr(rays, "coherent", False):
        return False                        # an image is walked in 8x8 tiles: coherent as it stands
    s = getattr(rays, "sort", None)
    if s is None:
        s = SORT_RAYS == "1" or (SORT_RAYS == "auto" and Q >= SORT_RAYS_MIN)
    return bool(s)


# A batch that IS a row-major pinhole image but was not declared one -- what the reference's API gives a caller: its
# RaysSpec has no image fields (data_spec.hpp:52-65) -- is recognised (r05): one origin for all rays and directions that
# move by a small step along a row and jump where a row wraps (svoxt_image_probe: one small launch, four words back).
# The first wrap gives the width; it must divide the batch, width and height must be multiples of 8 and the second row
# must wrap where the first did.  The batch is then walked in 8 x 8 pixel tiles like a declared image instead of being
# sorted: the reference-API route of the headline 1 102 -> the hinted route's figure.
# Results are per ray and do not depend on the walk: a wrong guess can only cost time.  That is what lets a training loop,
# which hands over NEW tensors every step, go without the host read: the answer is remembered per set of tensor objects
# (their versions and storage), and once IMAGE_TRUST_AFTER batches of a size in a row gave the same answer the next one
# is TAKEN to give it too -- its probe is still enqueued and read a step later, without waiting; one that disagrees ends
# the trust (waiting for the read drains the queue the host had filled ahead: 0.51 -> 0.77 ms per step at 800 x 800).
# DETECT_IMAGES False: off.
DETECT_IMAGES = True
IMAGE_TRUST_AFTER = 3
_IMAGE_SHAPES: dict = {}        # id(dirs tensor) -> [weakrefs to dirs and origins, (addresses, versions, Q), (H, W) or None, unread probe or None]
_IMAGE_TRUST: dict = {}         # Q -> [last answer, answers like it in a row, the entry of _IMAGE_SHAPES whose probe is unread or None]
_PROBE_RING: list = [None, 0]   # pinned int32 [64, 8], the next ticket


def _probe_row():
    """(row of the pinned ring, ticket): a probe's five words land there; the ticket tells a late reader whose they are."""
    if _PROBE_RING[0] is None:
        _PROBE_RING[0] = torch.zeros((64, 8), dtype=torch.int32, pin_memory=True)
    _PROBE_RING[1] = _PROBE_RING[1] % 0x7ffffff0 + 1
    return _PROBE_RING[0][_PROBE_RING[1] % 64], _PROBE_RING[1]


def _image_shape_from(info, Q):
    if info[0] == 1 and info[1] >= 1:
        W = info[2] + 1
        if W >= 8 and W % 8 == 0 and Q % W == 0 and (Q // W) % 8 == 0 and (info[1] < 2 or info[3] + 1 == 2 * W):
            return (Q // W, W)
    return None


def _settle_probe(ent, Q):
    """The probe of a batch whose answer was taken on trust has arrived: the entry gets the answer it gave, the trust its due."""
    _, row, ticket = ent[4]
    ent[4] = None
    tr = _IMAGE_TRUST.get(Q)
    if tr is not None and tr[2] is ent:
        tr[2] = None
    info = row.tolist()
    if info[4] != ticket:                  # the ring came round before anybody looked: nothing learnt, nothing trusted
        if tr is not None:
            tr[1] = 0
        return
    got, assumed = _image_shape_from(info, Q), ent[3]
    ent[3] = got
    if tr is not None:
        if got == assumed and got == tr[0]:
            tr[1] += 1
        else:
            tr[0], tr[1] = got, 1


def _detect_image(rays):
    if not DETECT_IMAGES:
        return None
    o, d = rays.origins, rays.dirs
    Q = o.shape[0]
    if Q < 4096 or Q % 64 or d.dtype != torch.float32 or not d.is_cuda or getattr(rays, "sort", None) is not None:
        return None
    # (the tensor OBJECTS, their versions and storage: an address alone comes back with other rays in it once a batch is
    # freed -- r05: a shuffled batch at the address of last step's image was walked as that image, 0.73 -> 2.2 ms)
    key = id(d)
    ent = _IMAGE_SHAPES.get(key)
    stamp = (o.data_ptr(), d.data_ptr(), o._version, d._version, Q)
    if ent is not None and ent[0]() is d and ent[1]() is o and ent[2] == stamp:
        if ent[4] is not None and ent[4][0].query():
            _settle_probe(ent, Q)
        return ent[3]
    tr = _IMAGE_TRUST.setdefault(Q, [None, 0, None])
    if tr[2] is not None and tr[2][4] is not None and tr[2][4][0].query():      # an earlier batch's probe has arrived
        _settle_probe(tr[2], Q)
    cr = _pack_rays(rays)
    dev = d.device
    with _on(dev):
        words = torch.empty((256,), dtype=torch.int32, device=dev)              # (SVOXT_IMAGE_PROBE_WORDS)
        row, ticket = _probe_row()
        _call("svoxt_image_probe", ctypes.byref(cr), _ptr(words), ticket, _stream(dev))
        row.copy_(words[:8], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
    ent = [weakref.ref(d, lambda _r, _k=key: _IMAGE_SHAPES.pop(_k, None)), weakref.ref(o), stamp, None, None]
    if tr[1] >= IMAGE_TRUST_AFTER:
        ent[3], ent[4] = tr[0], (ev, row, ticket)            # taken on trust; read when it has arrived (the next call that looks)
        if tr[2] is None or tr[2][4] is None:
            tr[2] = ent
    else:
        ev.synchronize()
        info = row.tolist()
        ent[3] = _image_shape_from(info, Q) if info[4] == ticket else None
        if ent[3] == tr[0]:
            tr[1] += 1
        else:
            tr[0], tr[1] = ent[3], 1
    if len(_IMAGE_SHAPES) >= 64:
        _IMAGE_SHAPES.clear()
    _IMAGE_SHAPES[key] = ent
    return ent[3]


def _in_coherent_order(tree, rays, opt):
    """(rays spec to render, perm): the batch with svoxt_ray_order's permutation attached when that pays
    (RaysSpec.order: the kernels walk the batch in that order, every ray's data stays where it is, so
    perm -- what a caller would have to undo -- is None), else as it is."""
    if not _wants_sort(rays):
        return rays, None
    shape = _detect_image(rays)
    if shape is not None:        # an undeclared image: walked in tiles, nothing to sort
        s = RaysSpec()
        s.origins, s.dirs, s.vdirs = rays.origins, rays.dirs, rays.vdirs
        s.image_height, s.image_width = shape
        s.sort = False
        return s, None
    s = RaysSpec()
    s.origins, s.dirs, s.vdirs = rays.origins, rays.dirs, rays.vdirs
    s.order = _ray_order32(tree, rays, opt)      # the kernels read it: no gather of the rays, no scatter of the pixels
    s.sort = False
    s.coherent = True           # neighbouring rays revisit the same leaves: the per-tile backward pays
    return s, None


def _to_caller_order(out_sorted, perm):
    if perm is None:
        return out_sorted
    return _permute_rows(out_sorted, perm, scatter=True)


def _need_grad(tree, rays):
    """Will a backward follow?  The caller's autograd function may say so (rays.need_grad, set by
    this package's own functions from ctx.needs_input_grad); the reference's do not, and inside an
    autograd.Function's forward grad mode is off either way: then the feature table decides."""
    ng = getattr(rays, "need_grad", None)
    return bool(tree.features.requires_grad) if ng is None else bool(ng)


def _planned_forward(kind, render, tree, rays, opt):
    """render(tree, rays_spec, opt, record) -> (out, lists): the forward of a plain call of operator `kind`."""
    rr, perm = _in_coherent_order(tree, rays, opt)
    if _need_grad(tree, rays):
        out, lists = render(tree, rr, opt, True)
        p = _Plan()
        f = tree.features
        p.kind, p.features, p.fkey, p.optkey = kind, f, (f._version, f.data_ptr()), _opt_key(opt)
        p.tkey, p.rkey = _tree_key(tree), _rays_key(rays)
        # `out` is the very tensor autograd hangs this call's node on; the node keeps the ctx, the ctx the
        # spec, the spec this plan: a reference to `out` itself would close a cycle that only the cyclic
        # collector breaks, and a forward nobody runs a backward for would keep its lists (tens to hundreds
        # of MB) until then.  A detached alias has no grad_fn and shares storage and version counter.
        # (rays: only a spec made here -- the caller's own would be a cycle spec -> plan -> spec)
        p.lists, p.out, p.over, p.perm, p.rays = lists, out.detach(), out._version, perm, (None if rr is rays else rr)
        rays._svoxt_plan = p
    else:
        out = render(tree, rr, opt, False)
        rays._svoxt_plan = None
    return _to_caller_order(out, perm)


def _take_plan(kind, tree, rays, opt):
    """The plan the forward of the same operator left on this spec -- or None (the backward then marches, as
    the reference's does) if anything the lists were recorded for may have changed since: the feature table
    (object, version, storage), the topology, the view rotations, the rays / camera, the options."""
    p = getattr(rays, "_svoxt_plan", None)
    if p is None:
        return None
    rays._svoxt_plan = None              # the lists serve one backward; a second one marches
    f = tree.features
    if p.kind != kind or p.features is not f or p.fkey != (f._version, f.data_ptr()) or p.optkey != _opt_key(opt) \
            or p.tkey != _tree_key(tree) or p.rkey != _rays_key(rays):
        return None
    return p


# ---------------------------------------------------------------------------
# Payloads one step away from a specialised one (r05).  The list kernels and the per-tile backward exist
# for three channels (x 1 / 4 / 9 / 16 / 25 basis functions or lobes) and for RGBA-style rows of 4 / 8 / 16 / 32 floats;
# the reference is generic in both (rt_kernel.cu:293-306).  A payload with FEWER channels -- one or two channels with a
# basis, an RGBA-style row of 2 .. 31 floats of another width -- is rendered as the next specialised one with DUMMY channels:
# feature columns of zeros in front of sigma, zeros in the upstream gradient.  Every channel's chain is its own, so the
# real channels' pixels are the same bits; a dummy channel's sigmoid is 0.5, but its upstream gradient is 0, so it adds
# +0.0 to total_color (the reference's sums over the channels, :410-425, 470-476: the real channels first, in order, then
# x + 0.0 = x) and to sum_c g_c, and its own gradient columns -- exact zeros -- are dropped.  Cost: one copy of the
# feature table in, one of the gradient out (75 MB at depth 8: ~0.03 ms each) against the generic kernels' accumulators in
# memory and marching backward (800 x 800 / depth 8, forward+backward: SH4 x 2 2.09 ms, a row of 6 floats 1.94:
# profiles/r05_generic_timing.txt has both ways).  PAD_PAYLOADS = False (test attribute): the generic kernels, as before.
# A component SUB-RANGE of a basis (min_comp / max_comp: the reference's `for i = min_comp .. max_comp`, rt_kernel.cu:295-298)
# goes the same way: the coefficients outside the range are zeros in the copy and the kernels run the full range -- the
# products outside it are (+-)0, the sums the same floats up to the sign of a zero that exp() does not see -- and the
# gradient columns outside the range, which the reference never touches, come back as zeros.
# More channels than three with a basis and rows wider than 32 floats stay generic.
# ---------------------------------------------------------------------------
PAD_PAYLOADS = True


def _pad_layout(tree: TreeSpec, opt: RenderOptions):
    """None, or (K', columns of real channels, dummy columns): the specialised payload this one is rendered as."""
    f = tree.features
    if not PAD_PAYLOADS or not isinstance(f, torch.Tensor) or f.dim() != 2 or f.dtype != torch.float32 or not f.is_cuda:
        return None
    K, fmt, bd = f.shape[1], int(opt.format), int(opt.basis_dim)
    if fmt == FORMAT_RGBA:
        if K < 2 or K > 32 or K in (4, 8, 16, 32):
            return None
        Kp = 4 if K < 4 else 8 if K < 8 else 16 if K < 16 else 32
        return Kp, K - 1, Kp - K, 1, None
    if bd not in (1, 4, 9, 16, 25) or (K - 1) % bd:
        return None
    C = (K - 1) // bd
    lo, hi = int(opt.min_comp), int(opt.max_comp)
    sub = (lo, hi) != (0, bd - 1)
    if C < 1 or C > 3 or (C == 3 and not sub) or (sub and not (0 <= lo <= hi < bd)):
        return None
    return 3 * bd + 1, C * bd, (3 - C) * bd, bd, ((lo, hi) if sub else None)


def _padded(tree: TreeSpec, rays, lay):
    """The tree spec with the padded feature table (kept on the rays / camera spec for the backward of the same call:
    the plan the forward leaves there names THIS tensor)."""
    f = tree.features
    key = (id(f), f._version, f.data_ptr(), lay)
    ent = getattr(rays, "_svoxt_pad", None)
    if ent is not None and ent[0] == key:
        return ent[1]
    Kp, real, dummy, w, sub = lay
    with torch.no_grad():
        fp = torch.cat([f[:, :real], f.new_zeros((f.shape[0], dummy)), f[:, real:]], dim=1)
        if sub is not None:
            fp[:, :real] *= _sub_mask(f, real, w, sub)
    fp.requires_grad_(f.requires_grad)
    tp = TreeSpec()
    tp.__dict__.update(tree.__dict__)
    tp.features = fp
    rays._svoxt_pad = (key, tp)
    return tp


def _sub_mask(like, real, w, sub):
    """float [real]: 1 for the coefficients min_comp .. max_comp of every channel, 0 outside."""
    i = torch.arange(real, device=like.device) % w
    return ((i >= sub[0]) & (i <= sub[1])).to(like.dtype)


def _full_range(opt: RenderOptions) -> RenderOptions:
    o = RenderOptions()
    o.__dict__.update(opt.__dict__)
    o.min_comp, o.max_comp = 0, int(opt.basis_dim) - 1
    return o


def _pad_cols(x, real_cols, dummy_cols):
    """[n, real + 1] -> [n, real + dummy + 1]: zeros in front of the last column."""
    return torch.cat([x[:, :real_cols], x.new_zeros((x.shape[0], dummy_cols)), x[:, real_cols:]], dim=1)


def _drop_cols(x, real_cols):
    """... and back: the real columns and the last one."""
    return torch.cat([x[:, :real_cols], x[:, -1:]], dim=1)


# More channels than a specialised kernel holds -- more than three with an SH / SG / ASG basis (the per-tile kernels keep three
# coefficients per record in LDS and registers), more than 31 in an RGBA-style row (rows of 32 floats are the widest with a
# channel-lane kernel) -- are rendered in GROUPS, each group as the specialised payload (a last group that is short goes
# through PAD_PAYLOADS like any short payload): channels are independent of each other in the forward (rt_kernel.cu:293-307:
# one sum per channel), and the gradient is linear in the upstream gradient (:365-494), so the groups' gradients add: a
# channel's coefficients get their group's, sigma gets the sum, the alpha column's upstream gradient goes with the first
# group alone.  Forward: the same bits per channel.  Gradient: the sigma entries are sums of per-group terms where the
# reference adds all channels into one total_color first -- the same value to rounding, held to the tight scale by the
# tests.  (r05; 800 x 800 / depth 8, SH9 x 4 forward+backward: 5.2 ms on the generic kernels, 1.35 in two groups.)
# GROUP_PAYLOADS False: off.
GROUP_PAYLOADS = True


def _group_layout(tree: TreeSpec, opt: RenderOptions):
    """None, or (coefficients per channel, channels, channels per group)."""
    f = tree.features
    if not GROUP_PAYLOADS or not isinstance(f, torch.Tensor) or f.dim() != 2 or f.dtype != torch.float32 or not f.is_cuda:
        return None
    if _numel(tree._weight_accum) or _numel(getattr(tree, "transformation_matrices", None)):
        return None
    K, fmt, bd = f.shape[1], int(opt.format), int(opt.basis_dim)
    if fmt == FORMAT_RGBA:
        return (1, K - 1, 31) if 32 < K <= 8 * 31 + 1 else None
    if fmt not in (FORMAT_SH, FORMAT_SG, FORMAT_ASG) or bd not in (1, 4, 9, 16, 25) or (K - 1) % bd \
            or (int(opt.min_comp), int(opt.max_comp)) != (0, bd - 1):
        return None
    C = (K - 1) // bd
    return (bd, C, 3) if 4 <= C <= 12 else None


def _groups(tree: TreeSpec, rays, lay):
    """[(tree spec of the group, rays spec of the group, first channel, channels)]: the groups' feature tables cut out of the
    caller's (their channels' columns + sigma), a spec object per group so that each keeps its own plan; kept on the caller's
    rays / camera spec for the backward of the same call."""
    f = tree.features
    key = (id(f), f._version, f.data_ptr(), lay)
    ent = getattr(rays, "_svoxt_groups", None)
    if ent is not None and ent[0] == key:
        return ent[1]
    bd, C, per = lay
    out = []
    for c0 in range(0, C, per):
        n = min(per, C - c0)
        with torch.no_grad():
            fg = torch.cat([f[:, c0 * bd:(c0 + n) * bd], f[:, -1:]], dim=1)
        fg.requires_grad_(f.requires_grad)
        tg = TreeSpec()
        tg.__dict__.update(tree.__dict__)
        tg.features = fg
        rg = type(rays)()
        rg.__dict__.update({k: v for k, v in rays.__dict__.items() if not k.startswith("_svoxt_")})
        out.append((tg, rg, c0, n))
    rays._svoxt_groups = (key, out)
    return out



def volume_render(tree: TreeSpec, rays: RaysSpec, opt: RenderOptions, record: bool = False):
    """rt_kernel.cu:1362-1379.

    `record=True` (not in the reference) asks for the sample lists a following
    volume_render_backward can replay; the call then returns (out, lists), with
    lists = None when recording does not apply (a payload without a specialised
    kernel, a component sub-range, BWD_LIST_SAMPLES = 0)."""
    lay = _pad_layout(tree, opt) if not record else None
    if lay is not None:
        _, real, dummy, w, sub = lay
        out = volume_render(_padded(tree, rays, lay), rays, opt if sub is None else _full_range(opt))
        return _drop_cols(out, real // w) if dummy else out
    glay = _group_layout(tree, opt) if not record else None
    if glay is not None:
        outs = [(volume_render(tg, rg, opt), n) for tg, rg, _, n in _groups(tree, rays, glay)]
        return torch.cat([o[:, :n] for o, n in outs] + [outs[0][0][:, -1:]], dim=1)
    if not record and AUTO_PLAN:
        return _planned_forward("volume", lambda t, r, o, rec: _volume_render(t, r, o, rec), tree, rays, opt)
    return _volume_render(tree, rays, opt, record)


# which kernels the last volume_render / volume_render_backward took (for bench.py's labels)
LAST_ROUTE = {"forward": None, "backward": None, "forward_terms": False}


def _volume_render(tree, rays, opt, record):
    # (r05) A forward whose march is wavefronts of its own -- every recording forward, and the two-kernel forward of rows of
    # 8 / 16 / 32 floats -- waits for the grid cell and nothing else: for them the grid in 4 x 4 x 4 bricks and a level finer
    # (fwd_roles_kernel at 800 x 800 / depth 8: 0.248 -> 0.234 ms; march_rec_kernel + shade_chan_kernel at 1024 x 1024 /
    # depth 9: 0.837 -> 0.801); the one-kernel forward, short of issue slots, keeps the row-major grid (include/svoxt.h,
    # SVOXT_ACCEL_BRICKS).
    bricks = _numel(tree._weight_accum) == 0 and (bool(record) or (int(opt.format) == FORMAT_RGBA and tree.features.shape[1] in (8, 16, 32)
                                                                    and FWD_SPLIT != "0"))
    ct, cr, co = _pack_tree_accel(tree, bricks), _pack_rays(rays), _pack_opts(opt)
    dev = tree.features.device
    lists = None
    wide = co.format == FORMAT_RGBA and ct.K in (8, 16, 32)
    # (per-leaf weight accumulation: the library's one-kernel forward alone adds the weights -- fwd_split_payload -- so no
    # scratch, mask or table is made for a forward that would not touch them; r05: its untouched scratch was read as the
    # pool's use, the r04 mistake of the 3-channel payloads over again)
    split = ((FWD_SPLIT != "0") if FWD_SPLIT != "" else wide) and ct.weight_accum is None
    can_rec = _lib.svoxt_can_record(ctypes.byref(ct), ctypes.byref(co)) if (record and BWD_LIST_SAMPLES > 0 and cr.Q > 0) else 0
    will_record = bool(can_rec)
    if can_rec == 2:
        # SG / ASG: lists serve the exact per-tile backward alone -- recorded when that is the backward that will run
        # (the conditions of _volume_render_backward), else the backward marches by itself as before
        tiled = cr.image_width > 0 and cr.image_width % 8 == 0 and cr.image_height % 8 == 0
        coherent = tiled or bool(getattr(rays, "coherent", False))
        will_record = bool(BWD_EXACT and BWD_TERMS and BWD_FUSED and (BWD_GATHER == 2 or (BWD_GATHER == 1 and coherent)))
    lflags = _list_flags(native=NATIVE_MATH and wide)
    # per-leaf view rotations, SH 1 / 4 / 9: march and shade in one launch too (no hand-over: their backward recomputes)
    xf_roles = bool(will_record and ct.xform and FWD_SPLIT == "" and FWD_OVERLAP and LIST_POOL and BWD_XF_FUSED and ct.N == 2
                    and ct.weight_accum is None and co.format == FORMAT_SH and co.basis_dim in (1, 4, 9)
                    and ct.K == 3 * co.basis_dim + 1)
    if xf_roles:
        lflags |= LISTS_FWD_TWO_KERNELS
    fills = _lib.svoxt_fwd_fills_terms(ctypes.byref(ct), ctypes.byref(co), lflags) \
        if (will_record and BWD_EXACT and BWD_TERMS and BWD_FUSED and BWD_GATHER) else 0
    # the march of the two-kernel forward reads a bit per row instead of gathering sigma -- where that
    # forward is what runs and the stop rule (which needs sigma itself) does not apply
    etab = None
    lists = None
    if will_record:
        with _on(dev):
            lists = SampleLists(cr.Q, _list_cap(ct, BWD_LIST_SAMPLES), dev)
    if ((split or fills == 3 or xf_roles) and FWD_SPLIT != "0") if will_record else (split and (co.stop_thresh == 0.0 or wide)):
        # rows of 8 / 16 / 32 floats in exact mode: the same pass leaves the rows' exponentials for the shade kernel
        # (and, on the lists, for the per-tile backward of this feature content)
        # (lists for a backward hold every sample with sigma > 0 whatever the forward's threshold: the mask of threshold 0)
        etab = _attach_sigma_mask(tree, ct, 0.0 if will_record else float(co.sigma_thresh), keep=not will_record,
                                  table=wide and split and not NATIVE_MATH, begin=lists)
    LAST_ROUTE["forward_terms"] = False
    # march and shade of a 3-channel payload as ONE launch (fwd_roles_kernel): the conditions of the library's launch_fwd_roles
    roles = bool(FWD_OVERLAP and LIST_POOL and ct.sigma_mask and ct.N == 2 and ct.xform is None and ct.weight_accum is None
                 and cr.Q > 0 and ((co.format == FORMAT_RGBA and ct.K == 4) or
                                   (co.format in (FORMAT_SH, FORMAT_SG, FORMAT_ASG) and co.basis_dim in (1, 4, 9, 16)
                                    and ct.K == 3 * co.basis_dim + 1 and (co.format == FORMAT_SH or ct.accel))))
    LAST_ROUTE["forward"] = (("march_rec_kernel + shade_chan_kernel (two-kernel forward, channels on lanes"
                              + (", native exp / rcp)" if NATIVE_MATH else ")") if wide else
                              "fwd_roles_kernel (march + shade_tile in one launch)" if roles else
                              "march_rec_kernel + shade_tile_kernel (two-kernel forward)") if split
                             else ("render_fwd_kernel" if can_rec or not record else "render_fwd_generic_kernel (no specialised instance: accumulators in "
                                   "memory as the reference keeps them)")) + (", recording sample lists" if will_record else "")
    with _on(dev):
        out = torch.empty((cr.Q, get_out_data_dim(opt, ct.K)), dtype=torch.float32, device=dev)
        if will_record:
            # (the walk of an image's tiles goes on the lists: the backward replays them the way they were recorded,
            # whatever svoxt_set_super_tile_rows says by then)
            lists.flags = lflags | max(0, _lib.svoxt_image_walk(ctypes.byref(ct), ctypes.byref(cr)))
            lists.bricks = bool(ct.accel and (ct.accel_log2 & 0x100))
            lists.exp_table = etab
            if fills:
                # the exact per-tile backward will want (att, e0, e1, e2) of every sample: this forward has them
                lists.terms = torch.empty((lists.pool_blocks * 512 * 4,), dtype=torch.float32, device=dev)
                lists.terms_state = fills
            cl = lists.c_struct()
            _call("svoxt_volume_render_fwd_record", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co),
                  _ptr(out), ctypes.byref(cl), _stream(dev))
            lists.note_usage()
            LAST_ROUTE["forward_terms"] = lists.terms_state in (2, 3)
            if lists.terms_state == 3 and FWD_SPLIT == "":
                LAST_ROUTE["forward"] = ("fwd_roles_kernel (march + shade_tile in one launch)" if roles else
                                         "march_rec_kernel + shade_tile_kernel (two-kernel forward)") + ", recording sample lists"
            if xf_roles:
                LAST_ROUTE["forward"] = "fwd_roles_kernel<XF> (march + shade_tile in one launch, a basis per record), recording sample lists"
        elif split and FWD_LIST_SAMPLES > 0 and cr.Q > 0:
            # scratch for the two-kernel forward (march, then shade per tile; the library falls
            # back to the one-kernel forward for payloads it does not cover).  (r04: only where that forward is what
            # runs -- the one-kernel forward never touched its scratch, and what note_usage then read as the pool's
            # use was uninitialised memory: a garbage size hint, looked at again after every forward)
            if LIST_POOL:
                scratch = SampleLists(cr.Q, _list_cap(ct, FWD_LIST_SAMPLES), dev, kind="scratch")
                scratch.flags = lflags
                cl = scratch.c_struct()
                _call("svoxt_volume_render_fwd_scratch", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co),
                      _ptr(out), ctypes.byref(cl), 0, _stream(dev))
                scratch.note_usage()
            else:
                nbytes = _lib.svoxt_fwd_workspace_bytes(cr.Q, FWD_LIST_SAMPLES)
                ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
                _call("svoxt_volume_render_fwd_ws", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co),
                      _ptr(out), _ptr(ws), nbytes, lflags, _stream(dev))
        else:
            _call("svoxt_volume_render_fwd", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co),
                  _ptr(out), _stream(dev))
    return (out, lists) if record else out


# The backward's arithmetic.  Default (BWD_EXACT): `accum` and the ray's final transmittance are added
# up sequentially over the ray's samples as the reference's first pass does (rt_kernel.cu:365-437),
# every gradient contribution bit-identical to the reference's formulas.  BWD_EXACT False opts
# into the single march: accum = sum_c g_c * out_c from the forward's output -- one sweep over the
# lists instead of two (0.34 vs 0.43 ms on the headline workload, r02).
# BWD_GATHER 0: always the per-ray backward (every sample's row goes to memory as shaped atomics); 1: the
# per-tile one (lists -> per-tile merge in LDS -> one atomic row per tile, window and feature row) for
# batches declared as images or in svoxt_ray_order's order; 2: whenever the payload allows.


def volume_render_backward(tree: TreeSpec, rays: RaysSpec, opt: RenderOptions,
                           grad_output: torch.Tensor, lists: SampleLists = None,
                           fwd_output: torch.Tensor = None) -> torch.Tensor:
    """rt_kernel.cu:1402-1426.  `lists` (optional, not in the reference): what
    volume_render(..., record=True) returned for the same tree / rays / options;
    `fwd_output` (optional): the output of that forward, which saves the
    backward its first pass (include/svoxt.h, svoxt_volume_render_bwd_replay).
    Neither given: what the forward of the same spec objects left behind (see _Plan)."""
    lay = _pad_layout(tree, opt) if (lists is None and fwd_output is None) else None
    if lay is not None:
        _check_input(grad_output, "grad_output")
        _, real, dummy, w, sub = lay
        if grad_output.dim() != 2 or grad_output.shape[1] != real // w + 1:
            raise RuntimeError("grad_output must be float32 [Q, C+1]")
        gp = volume_render_backward(_padded(tree, rays, lay), rays, opt if sub is None else _full_range(opt),
                                    _pad_cols(grad_output, real // w, dummy // w) if dummy else grad_output)
        rays._svoxt_pad = None
        gp = _drop_cols(gp, real) if dummy else gp
        if sub is not None:
            gp[:, :real] *= _sub_mask(gp, real, w, sub)
        return gp
    glay = _group_layout(tree, opt) if (lists is None and fwd_output is None) else None
    if glay is not None:
        _check_input(grad_output, "grad_output")
        bd, C, _ = glay
        if grad_output.dim() != 2 or grad_output.shape[1] != C + 1:
            raise RuntimeError("grad_output must be float32 [Q, C+1]")
        grad = torch.empty_like(tree.features)
        for tg, rg, c0, n in _groups(tree, rays, glay):
            alpha = grad_output[:, C:] if c0 == 0 else grad_output.new_zeros((grad_output.shape[0], 1))
            gg = volume_render_backward(tg, rg, opt, torch.cat([grad_output[:, c0:c0 + n], alpha], dim=1))
            grad[:, c0 * bd:(c0 + n) * bd] = gg[:, :n * bd]
            if c0 == 0:
                grad[:, -1] = gg[:, -1]
            else:
                grad[:, -1] += gg[:, -1]
        rays._svoxt_groups = None
        return grad
    if lists is None and fwd_output is None and AUTO_PLAN:
        p = _take_plan("volume", tree, rays, opt)
        if p is not None:
            _check_input(grad_output, "grad_output")
            g = grad_output if p.perm is None else _permute_rows(grad_output, p.perm, scatter=False)
            fo = p.out if (p.lists is not None and p.out._version == p.over) else None
            if fo is not None and fo.dim() == 3:
                fo = fo.view(-1, fo.shape[2])
            return _volume_render_backward(tree, p.rays if p.rays is not None else rays, opt, g, p.lists, fo)
    return _volume_render_backward(tree, rays, opt, grad_output, lists, fwd_output)


def _volume_render_backward(tree, rays, opt, grad_output, lists, fwd_output):
    # (the grid the forward that recorded the lists went through: a backward with lists marches overflowed rays at most)
    ct, cr, co = _pack_tree_accel(tree, bool(getattr(lists, "bricks", False))), _pack_rays(rays), _pack_opts(opt)
    _check_input(grad_output, "grad_output")
    if grad_output.dtype != torch.float32 or grad_output.dim() != 2 or grad_output.shape[0] != cr.Q:
        raise RuntimeError("grad_output must be float32 [Q, C+1]")
    dev = tree.features.device
    M, K = tree.features.shape
    # Accumulate into rows that start on 64-byte boundaries (fewer memory-side
    # atomic requests per row), then hand back the dense [M, K] the caller expects.
    stride = K if (K <= 8 or K % 16 == 0) else (K + 15) // 16 * 16
    if lists is not None and lists.consumed:
        lists = None              # a second backward over the same forward: the lists were rewritten
    # the two-kernel backward pays off when a wavefront's 64 rays revisit the same leaves,
    # i.e. for image batches walked in 8x8 tiles (0.94 -> 0.54 ms on the headline workload);
    # on shuffled rays it loses to the one-kernel backward (2.4 vs 1.65 ms)
    tiled = cr.image_width > 0 and cr.image_width % 8 == 0 and cr.image_height % 8 == 0
    coherent = tiled or bool(getattr(rays, "coherent", False))      # or sorted by svoxt_ray_order
    # rows wider than 32 floats (SH16: 49, SH25: 76): per tile only as ONE kernel over the hand-over the recording
    # forward left (lists.terms_state 2 / 3), exact arithmetic, no view rotations
    wide_sh = K > 32 and lists is not None and co.format in (FORMAT_SH, FORMAT_SG, FORMAT_ASG) and co.basis_dim in (16, 25) and \
        K == 3 * co.basis_dim + 1 and lists.terms is not None and lists.terms_state in (2, 3) and \
        BWD_EXACT and BWD_FUSED and BWD_TERMS and ct.xform is None
    gather = lists is not None and (K <= 32 or wide_sh) and grad_output.shape[1] == 4 and ct.N == 2 and \
        (BWD_GATHER == 2 or (BWD_GATHER == 1 and coherent))
    if lists is not None and co.format in (FORMAT_SG, FORMAT_ASG) and not (
            gather and BWD_FUSED and BWD_EXACT and ct.xform is None and lists.terms is not None and lists.terms_state == 3):
        lists = None              # SG / ASG lists serve the exact per-tile backward only: march instead
        gather = False
    with _on(dev):
        kept = _grad_scratch(dev, M, stride) if (GRAD_SCRATCH and lists is not None and stride != K and M > 0) else None
        buf = kept[0] if kept is not None else torch.empty((M, stride), dtype=torch.float32, device=dev)
        if lists is not None:
            if lists.aux.shape[0] != cr.Q or lists.aux.device != dev:
                raise RuntimeError("sample lists do not belong to this ray batch")
            fo = None
            # (the single-march tolerance mode differentiates the forward's own sum: only without thresholds)
            if fwd_output is not None and not BWD_EXACT and co.sigma_thresh == 0.0 and co.stop_thresh == 0.0:
                _check_input(fwd_output, "fwd_output")
                if fwd_output.shape != grad_output.shape or fwd_output.dtype != torch.float32:
                    raise RuntimeError("fwd_output must match grad_output")
                fo = fwd_output
            # (fo None: the fused kernel's exact form -- the only one view rotations take: SH up to 9 basis functions)
            fused = gather and BWD_FUSED and (ct.xform is None or (fo is None and BWD_EXACT and BWD_XF_FUSED and co.format == FORMAT_SH
                                                                   and co.basis_dim in (1, 4, 9)))
            wide = co.format == FORMAT_RGBA and grad_output.shape[1] in (8, 16, 32) and K == grad_output.shape[1]
            # ... per tile (grad_wide_kernel) for coherent batches on N = 2 trees, else per ray (render_bwd_kernel<ONEPASS>)
            wide_tile = wide and fo is None and BWD_TERMS and BWD_FUSED and ct.N == 2 and ct.xform is None and \
                (BWD_GATHER == 2 or (BWD_GATHER == 1 and coherent))
            if wide and fo is None and BWD_TERMS and not gather:
                # one sigmoid pass instead of two: the first sweep leaves a float per list slot for the second
                need = lists.pool_blocks * 512 * (2 if wide_tile else 1)       # (attenuation, total_color) / total_color alone
                if lists.terms is None or lists.terms.numel() < need:
                    lists.terms = torch.empty((need,), dtype=torch.float32, device=dev)
                lists.terms_state = 0
            if gather and not fused:
                # with view rotations a second plane holds each sample's rotated direction
                planes = 2 if ct.xform is not None else 1
                lists.coef = torch.empty((planes * lists.S, cr.Q, 4), dtype=torch.float32, device=dev)
                lists.consumed = True         # the two-kernel form rewrites rec
            cl = lists.c_struct()
            if wide_tile and lists.exp_table is not None and not (lists.flags & LISTS_NATIVE_MATH):
                ct.exp_table = lists.exp_table.data_ptr()      # (the table of the feature content the lists were recorded with)
            LAST_ROUTE["backward"] = (
                ("grad_fused_kernel<EXACT, XF> (two sweeps over the lists, a basis per record + per-tile merge)" if ct.xform is not None else
                 "grad_fused_kernel<EXACT> (two sweeps over the lists + per-tile merge)" if fo is None else
                 "grad_fused_kernel (one sweep over the lists + per-tile merge; accum from the forward's output)") if fused else
                "render_bwd_kernel<GATHER> + grad_merge_kernel (list walk, then per-tile merge)" if gather else
                ("grad_wide_kernel (two sweeps over the lists, sigmoids per record once + once per distinct row; per-tile merge"
                 + (", native exp / rcp)" if lists.flags & LISTS_NATIVE_MATH else
                    ", exponentials from the forward's table)" if lists.exp_table is not None else ")"))
                if wide_tile else
                "render_bwd_kernel<ONEPASS> (two list walks, sigmoids formed once; one atomic row + one sigma atomic per sample)"
                if (wide and fo is None and BWD_TERMS) else
                "render_bwd_kernel<REPLAY> (list walk, one atomic row per sample)")
            if fused or wide_tile:
                cl.coef_bytes = -1            # list walk and merge as one kernel: no buffer, rec stays as recorded
            if kept is not None:
                cl.flags |= LISTS_GRAD_ZEROED
            _call("svoxt_volume_render_bwd_replay", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co),
                  _ptr(grad_output), grad_output.shape[1], _ptr(buf), stride, ctypes.byref(cl), _ptr(fo),
                  _stream(dev))
        else:
            LAST_ROUTE["backward"] = "render_bwd_kernel (marches, one atomic row per sample)"
            ws_bytes = _lib.svoxt_bwd_workspace_bytes(cr.Q, BWD_LIST_SAMPLES)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev) if ws_bytes > 0 else None
            _call("svoxt_volume_render_bwd", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co),
                  _ptr(grad_output), grad_output.shape[1], _ptr(buf), stride, _ptr(ws), ws_bytes, _stream(dev))
        if stride == K:
            grad = buf
        else:
            grad = torch.empty((M, K), dtype=torch.float32, device=dev)
            if kept is not None:
                _call("svoxt_compact_rows_clear", _ptr(buf), M, K, stride, _ptr(grad), _stream(dev))
                kept[1] = True
            else:
                _call("svoxt_compact_rows", _ptr(buf), M, K, stride, _ptr(grad), _stream(dev))
    return grad


def volume_render_image(tree: TreeSpec, cam: CameraSpec, opt: RenderOptions, record: bool = False):
    """rt_kernel.cu:1382-1399: [height, width, C+1] image of a pinhole camera; the
    rays (cam2world_ray, and maybe_world2ndc when opt.ndc_width >= 0) are generated
    inside the kernels.  (The reference's version cannot run: it allocates and
    dispatches on the int32 index tensor, :1390-1393.)"""
    res = volume_render(tree, cam, opt, record=record)      # (a plain call leaves its plan on `cam`)
    out = res[0] if record else res
    out = out.view(int(cam.height), int(cam.width), -1)
    return (out, res[1]) if record else out


def volume_render_image_backward(tree: TreeSpec, cam: CameraSpec, opt: RenderOptions,
                                 grad_output: torch.Tensor, lists: SampleLists = None,
                                 fwd_output: torch.Tensor = None) -> torch.Tensor:
    """rt_kernel.cu:1428-1452; grad_output [height, width, C+1]."""
    _check_input(grad_output, "grad_output")
    if grad_output.dim() != 3 or grad_output.shape[0] != cam.height or grad_output.shape[1] != cam.width:
        raise RuntimeError("grad_output must be float32 [height, width, C+1]")
    flat = grad_output.view(-1, grad_output.shape[2])
    fo = None if fwd_output is None else fwd_output.view(-1, grad_output.shape[2])
    return volume_render_backward(tree, cam, opt, flat, lists=lists, fwd_output=fo)


def render_depth(tree: TreeSpec, rays: RaysSpec, opt: RenderOptions) -> torch.Tensor:
    """rt_kernel.cu:1506-1523."""
    ct, cr, co = _pack_tree_accel(tree), _pack_rays(rays), _pack_opts(opt)
    dev = tree.features.device
    with _on(dev):
        depth = torch.empty((cr.Q, 1), dtype=torch.float32, device=dev)
        _call("svoxt_render_depth", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co),
              _ptr(depth), _stream(dev))
    return depth


def opacity_render(tree: TreeSpec, rays: RaysSpec, opt: RenderOptions, record: bool = False):
    """rt_kernel.cu:1574-1591.  `record=True` (not in the reference): also return the
    sample lists opacity_render_backward can walk."""
    if not record and AUTO_PLAN:
        return _planned_forward("opacity", lambda t, r, o, rec: _opacity_render(t, r, o, rec), tree, rays, opt)
    return _opacity_render(tree, rays, opt, record)


def _opacity_render(tree, rays, opt, record):
    ct, cr, co = _pack_tree_accel(tree), _pack_rays(rays), _pack_opts(opt)
    dev = tree.features.device
    lists = None
    with _on(dev):
        out = torch.empty((cr.Q, 1), dtype=torch.float32, device=dev)
        if record and BWD_LIST_SAMPLES > 0 and cr.Q > 0:
            lists = SampleLists(cr.Q, _list_cap(ct, BWD_LIST_SAMPLES), dev)
            lists.flags = max(0, _lib.svoxt_image_walk(ctypes.byref(ct), ctypes.byref(cr)))
            cl = lists.c_struct()
            _call("svoxt_opacity_render_fwd_record", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co),
                  _ptr(out), ctypes.byref(cl), _stream(dev))
        else:
            _call("svoxt_opacity_render_fwd", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co),
                  _ptr(out), _stream(dev))
    return (out, lists) if record else out


def opacity_render_backward(tree: TreeSpec, rays: RaysSpec, opt: RenderOptions,
                            grad_output: torch.Tensor, lists: SampleLists = None) -> torch.Tensor:
    """rt_kernel.cu:1593-1616.  `lists` (optional, not in the reference): what
    opacity_render(..., record=True) returned for the same tree / rays / options; not given:
    what the forward of the same spec objects left behind (see _Plan)."""
    if lists is None and AUTO_PLAN:
        p = _take_plan("opacity", tree, rays, opt)
        if p is not None:
            _check_input(grad_output, "grad_output")
            g = grad_output if p.perm is None else _permute_rows(grad_output, p.perm, scatter=False)
            return _opacity_render_backward(tree, p.rays if p.rays is not None else rays, opt, g, p.lists)
    return _opacity_render_backward(tree, rays, opt, grad_output, lists)


def _opacity_render_backward(tree, rays, opt, grad_output, lists):
    ct, cr, co = _pack_tree_accel(tree), _pack_rays(rays), _pack_opts(opt)
    _check_input(grad_output, "grad_output")
    if grad_output.dtype != torch.float32 or grad_output.numel() != cr.Q:
        raise RuntimeError("grad_output must be float32 [Q, 1]")
    dev = tree.features.device
    if lists is not None and lists.consumed:
        lists = None
    with _on(dev):
        grad = torch.empty_like(tree.features)
        if lists is not None:
            if lists.aux.shape[0] != cr.Q or lists.aux.device != dev:
                raise RuntimeError("sample lists do not belong to this ray batch")
            lists.consumed = True
            cl = lists.c_struct()
            _call("svoxt_opacity_render_bwd_replay", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co),
                  _ptr(grad_output), _ptr(grad), 0, ctypes.byref(cl), _stream(dev))
        else:
            _call("svoxt_opacity_render_bwd", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co),
                  _ptr(grad_output), _ptr(grad), _stream(dev))
    return grad


from ._extras import (assign_vertical, build_octree, bwd_check, bwd_counters, calc_corners, construct_tree,  # noqa: E402,F401
                      count_forward, count_touched, grid_weight_render, motion_feature_render,
                      motion_feature_render_backward, motion_render, p2v, p2v_backward, quantize_median_cut,
                      query_vertical, query_vertical_backward, refine_leaves, warp_vertices, warp_vertices_backward)
