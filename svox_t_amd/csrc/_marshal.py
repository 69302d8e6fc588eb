"""Torch tensors -> the C ABI's structs: the spec classes of the reference's pybind11 module (svox_t/csrc/svox.cpp:74-117),
their checks (data_spec.hpp:38-43, 57-64, 85-110) and pointer extraction, the launch helpers (_call, _stream), and the
acceleration-grid cache every marching entry point shares (_pack_tree_accel)."""
from __future__ import annotations

import ctypes
import os
import sys
import weakref

import torch

from ._abi import _CMotion, _COptions, _CRays, _CTree, _lib

# ---------------------------------------------------------------------------
# Spec classes (svox.cpp:74-117): default-constructed, read/write attributes.
# ---------------------------------------------------------------------------

class RaysSpec:
    def __init__(self):
        self.origins = None
        self.dirs = None
        self.vdirs = None
        # optional (not in the reference): the batch is a row-major image of this
        # size; the kernels then walk it in 8x8 tiles.  0 = unknown.
        self.image_width = 0
        self.image_height = 0
        # optional (not in the reference): render the batch in svoxt_ray_order's order (True / False;
        # None = decide by size, SORT_RAYS below) / the batch already is in such an order
        self.sort = None
        self.coherent = False
        # optional (not in the reference): int32 [Q] permutation (svoxt_ray_order's); launch thread i then works
        # on ray order[i] -- the batch is walked in that order, nothing is gathered or scattered
        self.order = None


class TreeSpec:
    def __init__(self):
        self.features = None
        self.data = None
        self.child = None
        self.parent_depth = None
        self.extra_data = None
        self.offset = None
        self.scaling = None
        self._weight_accum = None
        self.joint_features = None
        self.skinning_weights = None
        self.joint_index = None
        self.n_internal = 0
        self.transformation_matrices = None
        # optional (not in the reference): the caller's promise that `features` is not written behind
        # torch's back (through `.data`, a raw pointer, a storage swap) while this tree is rendered, so that
        # data derived from its CONTENT (the sigma bitmask) may be cached across forwards on the tensor's
        # version counter.  False: such data is rebuilt by every forward.  N3Tree.static_features sets it.
        self.static_features = False


class CameraSpec:
    def __init__(self):
        self.c2w = None
        self.fx = 0.0
        self.fy = 0.0
        self.width = 0
        self.height = 0


class RenderOptions:
    def __init__(self):
        self.step_size = 0.0
        self.background_brightness = 0.0
        self.format = 0
        self.basis_dim = 0
        self.ndc_width = 0
        self.ndc_height = 0
        self.ndc_focal = 0.0
        self.min_comp = 0
        self.max_comp = 0
        self.sigma_thresh = 0.0
        self.stop_thresh = 0.0


# ---------------------------------------------------------------------------
# Marshalling
# ---------------------------------------------------------------------------

def _check_input(x, name):
    """CHECK_INPUT (data_spec.hpp:38-43)."""
    if not isinstance(x, torch.Tensor):
        raise RuntimeError(f"{name} must be a tensor")
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _numel(x):
    return 0 if x is None else x.numel()


def _ptr(x):
    return None if (x is None or x.numel() == 0) else x.data_ptr()


def _pack_tree(tree: TreeSpec) -> _CTree:
    """TreeSpec.check() (data_spec.hpp:85-110) + pointer extraction."""
    _check_input(tree.features, "features")
    _check_input(tree.data, "data")
    _check_input(tree.child, "child")
    if tree.parent_depth is not None:
        _check_input(tree.parent_depth, "parent_depth")
    _check_input(tree.offset, "offset")
    _check_input(tree.scaling, "scaling")
    for nm in ("extra_data", "_weight_accum", "joint_features", "skinning_weights",
               "joint_index", "transformation_matrices"):
        t = getattr(tree, nm)
        if _numel(t):
            _check_input(t, nm)
    if tree.features.dtype != torch.float32 or tree.features.dim() != 2:
        raise RuntimeError("features must be a float32 [M, K] tensor")
    if tree.child.dtype != torch.int32 or tree.child.dim() != 4:
        raise RuntimeError("child must be an int32 [n, N, N, N] tensor")
    if tree.data.dtype != torch.int32 or tree.data.numel() != tree.child.numel():
        raise RuntimeError("data must be an int32 [n, N, N, N, 1] tensor matching child")
    if tree.offset.dtype != torch.float32 or tree.scaling.dtype != torch.float32 \
            or tree.offset.numel() != 3 or tree.scaling.numel() != 3:
        raise RuntimeError("offset / scaling must be float32 tensors of 3 elements")
    dev = tree.features.device
    for nm in ("data", "child", "offset", "scaling"):
        if getattr(tree, nm).device != dev:
            raise RuntimeError(f"{nm} must be on the same device as features")
    n_internal = int(tree.n_internal) if tree.n_internal else tree.child.shape[0]
    if n_internal > tree.child.shape[0]:
        raise RuntimeError("n_internal exceeds the capacity of child")
    c = _CTree()
    c.features = _ptr(tree.features)
    c.M, c.K = tree.features.shape
    c.N = tree.child.shape[1]
    c.data = tree.data.data_ptr()
    c.child = tree.child.data_ptr()
    c.n_internal = n_internal
    c.offset = tree.offset.data_ptr()
    c.scaling = tree.scaling.data_ptr()
    if _numel(tree.extra_data):
        if tree.extra_data.dtype != torch.float32 or tree.extra_data.dim() != 2:
            raise RuntimeError("extra_data must be a float32 2-D tensor")
        c.extra_data = tree.extra_data.data_ptr()
        c.extra_rows, c.extra_cols = tree.extra_data.shape
    if _numel(tree._weight_accum):
        if tree._weight_accum.dtype != torch.float32 or \
                tree._weight_accum.numel() != tree.child.numel():
            raise RuntimeError("_weight_accum must be float32 with one entry per leaf slot")
        c.weight_accum = tree._weight_accum.data_ptr()
    if _numel(tree.transformation_matrices):
        x = tree.transformation_matrices
        if x.dtype != torch.float32 or x.dim() != 3 or x.shape[0] != tree.features.shape[0] or \
                tuple(x.shape[1:]) not in ((3, 3), (4, 4)):
            raise RuntimeError("transformation_matrices must be float32 [M, 3, 3] or [M, 4, 4]")
        _check_input(x, "transformation_matrices")
        c.xform = x.data_ptr()
        c.xform_dim = x.shape[1]
    return c


def _pack_camera(cam: "CameraSpec") -> _CRays:
    """CameraSpec.check() (data_spec.hpp:120-125): the ray batch is the image of a
    pinhole camera; the kernels generate the rays themselves."""
    _check_input(cam.c2w, "c2w")
    if not cam.c2w.is_floating_point() or cam.c2w.dim() != 2 or cam.c2w.shape[1] != 4:
        raise RuntimeError("c2w must be a floating point [3 or 4, 4] matrix")
    if cam.c2w.dtype != torch.float32 or cam.c2w.shape[0] < 3:
        raise RuntimeError("c2w must be float32 with at least 3 rows")
    w, h = int(cam.width), int(cam.height)
    if w < 1 or h < 1:
        raise RuntimeError("camera width / height must be positive")
    c = _CRays()
    c.Q = w * h
    c.image_width, c.image_height = w, h
    c.c2w, c.fx, c.fy = cam.c2w.data_ptr(), float(cam.fx), float(cam.fy)
    return c


def _pack_rays(rays) -> _CRays:
    """RaysSpec.check() (data_spec.hpp:57-64); a CameraSpec selects camera mode."""
    if isinstance(rays, CameraSpec):
        return _pack_camera(rays)
    for nm in ("origins", "dirs", "vdirs"):
        t = getattr(rays, nm)
        _check_input(t, nm)
        if not t.is_floating_point():
            raise RuntimeError(f"{nm} must be floating point")
        if t.dtype != torch.float32:
            raise RuntimeError(f"{nm} must be float32 (the HIP path is fp32 only)")
        if t.dim() != 2 or t.shape[1] != 3:
            raise RuntimeError(f"{nm} must have shape [Q, 3]")
    Q = rays.origins.shape[0]
    if rays.dirs.shape[0] != Q or rays.vdirs.shape[0] != Q:
        raise RuntimeError("origins, dirs and vdirs must have the same number of rays")
    c = _CRays()
    c.origins, c.dirs, c.vdirs = _ptr(rays.origins), _ptr(rays.dirs), _ptr(rays.vdirs)
    c.Q = Q
    w, h = int(getattr(rays, "image_width", 0) or 0), int(getattr(rays, "image_height", 0) or 0)
    order = getattr(rays, "order", None)
    if order is not None:
        _check_input(order, "order")
        if order.dtype != torch.int32 or order.dim() != 1 or order.shape[0] != Q:
            raise RuntimeError("order must be int32 [Q]")
        c.order = _ptr(order)
    elif w * h == Q:
        c.image_width, c.image_height = w, h
    return c


def _pack_opts(opt: RenderOptions) -> _COptions:
    return _COptions(float(opt.step_size), float(opt.background_brightness),
                     int(opt.format), int(opt.basis_dim),
                     int(opt.ndc_width), int(opt.ndc_height), float(opt.ndc_focal),
                     int(opt.min_comp), int(opt.max_comp),
                     float(opt.sigma_thresh), float(opt.stop_thresh))


def _stream(device):
    """torch's CURRENT stream on `device` as the library takes it (the raw handle: torch.cuda.current_stream() builds a
    Stream object around it first -- 4.6 us a call, five calls a step)."""
    idx = device.index
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx))


class _NoContext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_CONTEXT = _NoContext()


def _on(device):
    """`with _on(dev):` -- torch.cuda.device(dev) where dev is not the current device already (the library launches on the
    current device; the context manager costs ~2 us each way, ten a step, and a single-GPU process never needs it)."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_CONTEXT
    return torch.cuda.device(device)


def _call(name, *args):
    rc = getattr(_lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {_lib.svoxt_last_error().decode()}")


def get_out_data_dim(opt: RenderOptions, K: int) -> int:
    """get_out_data_dim (rt_kernel.cu:1352-1358)."""
    n = _lib.svoxt_out_data_dim(ctypes.byref(_pack_opts(opt)), K)
    if n < 0:
        raise RuntimeError("invalid RenderOptions for get_out_data_dim")
    return n


# acceleration-grid resolution: None = chosen from the tree size, 0 = no grid, g = 2^g cells per axis (<= 8)
ACCEL_LOG2 = (lambda v: None if v is None else max(0, min(8, int(v))))(os.environ.get("SVOXT_ACCEL_LOG2"))

# ---------------------------------------------------------------------------
# Acceleration grid cache.  The grid (include/svoxt.h, svoxt_accel_build) is
# derived from the *contents* of child and data.  An entry therefore hangs off
# the `child` tensor OBJECT (weakly: it dies with the tensor, so a recycled
# device address can never alias a stale grid) and is valid only while the torch
# version counters of child and data are unchanged (every in-place torch op
# bumps them, as N3Tree.refine / construct_tree do) and `data` is the same
# tensor object.  A writer that bypasses the version counter (`child.data[...] = v`, a kernel of its own)
# must call invalidate_caches(); N3Tree.refine / construct_tree / parallel.broadcast_tree do.
# ACCEL_LOG2 (SVOXT_ACCEL_LOG2) = 0 disables the grid, = g forces a resolution; default: chosen from the tree size.
# ---------------------------------------------------------------------------
_ACCEL_CACHE: dict = {}     # (id(child tensor), bricks) -> (weakref to it, ...); entries are dropped when the tensor dies


def _switch(name: str):
    """A test switch as the PACKAGE holds it now (tests set svox_t_amd.csrc.<name>; this module's own global is the default)."""
    pkg = sys.modules.get(__package__)
    return getattr(pkg, name, globals()[name])


def _accel_log2_for(n_internal: int, N: int, feature_bytes: int = 0, marching: bool = False) -> int:
    forced = _switch("ACCEL_LOG2")
    if forced is not None:
        return forced
    if N != 2 or n_internal < 64:
        return 0
    slots = n_internal * 8
    # the smallest grid with at least as many cells as the tree has leaf slots
    # (D=8 shell tree: 128^3 cells = 16 MiB against 7.6 MiB of topology) ...
    g = max(4, min(7, -(-slots.bit_length() // 3)))
    # ... and one level finer (256^3 = 64 MiB of 4-byte cells at most) while cells, node pairs and the feature table
    # still fit the 256 MiB Infinity Cache together: every leaf crossing of a depth-8 tree is then ONE
    # dependent load instead of two (r02, 800x800 on the depth-8 tree: forward 0.246 -> 0.226 ms;
    # with a backward behind it no change, 972 Mrays/s either way).  Past the cache the finer grid
    # loses (depth 9, 578 MB of features: 805 -> 794 Mrays/s forward).
    finer = 4 * (1 << (3 * (g + 1))) + 64 * n_internal
    if g + 1 <= 8 and feature_bytes > 0 and finer + feature_bytes <= 224 * (1 << 20):
        g += 1
    # (r05) ... and one level finer whatever the cache holds when the march is wavefronts of its own that wait for these
    # loads and nothing else (`marching`: the two-kernel / one-launch forwards; the grid then lies in bricks too): depth 9 /
    # 578 MB of features at 1024 x 1024, g 7 -> 8 in bricks: forward 0.837 -> 0.801 ms, forward + backward 439 -> 446
    # Mrays/s (g 8 row-major 0.828; g 7 in bricks 0.837) -- a dependent 8-byte load per crossing less.
    elif marching and g + 1 <= 8:
        g += 1
    return g


ACCEL_BRICKS = None     # test switch: True / False force the grid's layout; None: the caller's choice (_pack_tree_accel)
SVOXT_ACCEL_BRICKS = 0x100


def _accel_for(tree: TreeSpec, ct: _CTree, bricks: bool = False):
    forced = _switch("ACCEL_BRICKS")
    bricks = bool(bricks if forced is None else forced)
    g = _accel_log2_for(ct.n_internal, ct.N, tree.features.numel() * tree.features.element_size(), marching=bricks)
    if g == 0 or ct.N != 2 or max(ct.M, ct.n_internal) >= (1 << 27) - 1:       # (4-byte cells: 27 index bits)
        return None, 0
    bricks = bricks and g >= 2
    key = (id(tree.child), bricks)      # (a tree rendered both ways -- training steps and forward-only views -- keeps both grids)
    ent = _ACCEL_CACHE.get(key)
    if ent is not None:
        cref, cv, dref, dv, n_int, eg, cells = ent
        if cref() is tree.child and cv == (tree.child._version, tree.child.data_ptr()) and dref() is tree.data \
                and dv == (tree.data._version, tree.data.data_ptr()) and n_int == ct.n_internal and eg == g:
            return cells, g | (SVOXT_ACCEL_BRICKS if bricks else 0)
    dev = tree.child.device
    with _on(dev):
        nbytes = _lib.svoxt_accel_bytes(g, ct.n_internal)       # grid cells + (child, data) pairs
        cells = torch.empty((nbytes // 8, 2), dtype=torch.int32, device=dev)
        _call("svoxt_accel_build", ctypes.byref(ct), g | (SVOXT_ACCEL_BRICKS if bricks else 0), _ptr(cells), _stream(dev))
    # (versions AND data pointers: `tensor.data = other` swaps the storage without touching the version counter)
    _ACCEL_CACHE[key] = (weakref.ref(tree.child, lambda _r, _k=key: _ACCEL_CACHE.pop(_k, None)),
                         (tree.child._version, tree.child.data_ptr()), weakref.ref(tree.data),
                         (tree.data._version, tree.data.data_ptr()), ct.n_internal, g, cells)
    return cells, g | (SVOXT_ACCEL_BRICKS if bricks else 0)


def _drop_accel(child) -> None:
    """Forget the grids (either layout) built from this child tensor."""
    for b in (False, True):
        _ACCEL_CACHE.pop((id(child), b), None)


def _pack_tree_accel(tree: TreeSpec, bricks: bool = False) -> _CTree:
    """_pack_tree + the (cached) acceleration grid for the marching kernels; `bricks`: in the layout that pays where a
    kernel waits for the cell load alone (include/svoxt.h, SVOXT_ACCEL_BRICKS)."""
    ct = _pack_tree(tree)
    cells, g = _accel_for(tree, ct, bricks)
    if cells is not None:
        ct.accel = cells.data_ptr()
        ct.accel_log2 = g
        ct._keepalive = cells
    return ct
