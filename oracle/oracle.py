"""ctypes front end of the CPU oracle (oracle/svoxt_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py -- never by svox_t_amd/.  All arrays are
numpy, C-contiguous; nothing here touches a GPU or /root/reference.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import NamedTuple, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsvoxt_oracle.so")

FORMAT_RGBA, FORMAT_SH, FORMAT_SG, FORMAT_ASG = 0, 1, 2, 3


class RenderOptions(ctypes.Structure):
    """svox_t/csrc/include/data_spec.hpp:129-145, same field order."""
    _fields_ = [
        ("step_size", ctypes.c_float),
        ("background_brightness", ctypes.c_float),
        ("format", ctypes.c_int),
        ("basis_dim", ctypes.c_int),
        ("ndc_width", ctypes.c_int),
        ("ndc_height", ctypes.c_int),
        ("ndc_focal", ctypes.c_float),
        ("min_comp", ctypes.c_int),
        ("max_comp", ctypes.c_int),
        ("sigma_thresh", ctypes.c_float),
        ("stop_thresh", ctypes.c_float),
    ]


def make_options(step_size=1e-3, background_brightness=1.0, format=FORMAT_RGBA,
                 basis_dim=-1, min_comp=0, max_comp=None, sigma_thresh=0.0,
                 stop_thresh=0.0) -> RenderOptions:
    if max_comp is None:
        max_comp = basis_dim - 1
    return RenderOptions(step_size, background_brightness, format, basis_dim,
                         -1, -1, 0.0, min_comp, max_comp, sigma_thresh, stop_thresh)


class Counters(NamedTuple):
    rays_hit: int
    steps: int
    levels: int
    valid: int
    active: int


def build(force: bool = False) -> str:
    """Compile the oracle with oracle/Makefile (g++)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "svoxt_oracle.cpp")):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        assert _lib.svoxt_oracle_sizeof_options() == ctypes.sizeof(RenderOptions)
    return _lib


def num_threads() -> int:
    return lib().svoxt_oracle_num_threads()


def set_num_threads(n: int) -> None:
    lib().svoxt_oracle_set_num_threads(int(n))


def use_libm_exp(on: bool) -> None:
    """Switch the oracle's expf between the portable fixed-sequence one (default;
    bit-identical to the HIP kernels' pexpf) and glibc's expf."""
    lib().svoxt_oracle_use_libm_exp(int(bool(on)))


def expf(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib().svoxt_oracle_expf(x.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(x.size),
                            y.ctypes.data_as(ctypes.c_void_p))
    return y


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(np.asarray(a), dtype=dtype)


class Tree:
    """Host copy of the tree tensors the kernels read."""

    def __init__(self, features, data, child, offset=(0, 0, 0), scaling=(1, 1, 1),
                 extra=None, dtype=np.float32):
        self.dtype = dtype
        self.features = _c(features, dtype)
        self.data = _c(data, np.int32)
        self.child = _c(child, np.int32)
        assert self.child.ndim == 4 and self.data.size == self.child.size
        self.N = int(self.child.shape[1])
        self.M, self.K = (int(v) for v in self.features.shape)
        self.offset = _c(offset, dtype)
        self.scaling = _c(scaling, dtype)
        self.extra = None if extra is None else _c(extra, dtype)

    def astype(self, dtype):
        return Tree(self.features, self.data, self.child, self.offset, self.scaling,
                    self.extra, dtype=dtype)

    def _args(self, with_extra=True):
        a = [_p(self.features), ctypes.c_int64(self.M), ctypes.c_int(self.K),
             _p(self.data), _p(self.child), ctypes.c_int(self.N),
             _p(self.offset), _p(self.scaling)]
        if with_extra:
            if self.extra is None:
                a += [None, ctypes.c_int(0), ctypes.c_int(0)]
            else:
                a += [_p(self.extra), ctypes.c_int(self.extra.shape[0]),
                      ctypes.c_int(self.extra.shape[1])]
        return a


def out_data_dim(opt: RenderOptions, K: int) -> int:
    """get_out_data_dim (rt_kernel.cu:1352-1358) = C + 1."""
    if opt.format != FORMAT_RGBA:
        return (K - 1) // opt.basis_dim + 1
    return K


def _rays(tree: Tree, origins, dirs, vdirs):
    o, d, v = (_c(x, tree.dtype) for x in (origins, dirs, vdirs))
    assert o.shape == d.shape == v.shape and o.shape[1] == 3
    return o, d, v


class transformation_matrices:
    """`with transformation_matrices(x):` -- per-leaf [M, 3, 3] view rotations
    (TreeSpec.transformation_matrices) for the f32 render / backward calls inside."""

    def __init__(self, xform):
        self.x = _c(xform, np.float32)
        assert self.x.ndim == 3 and self.x.shape[1:] in ((3, 3), (4, 4))

    def __enter__(self):
        lib().svoxt_oracle_set_transformation_matrices(_p(self.x), int(self.x.shape[1]))
        return self

    def __exit__(self, *exc):
        lib().svoxt_oracle_set_transformation_matrices(None, 3)


def volume_render(tree: Tree, origins, dirs, vdirs, opt: RenderOptions, count=False):
    o, d, v = _rays(tree, origins, dirs, vdirs)
    Q = o.shape[0]
    out = np.zeros((Q, out_data_dim(opt, tree.K)), dtype=tree.dtype)
    cnt = np.zeros(5, dtype=np.int64)
    fn = lib().svoxt_oracle_volume_render_f32 if tree.dtype == np.float32 \
        else lib().svoxt_oracle_volume_render_f64
    fn(*tree._args(), _p(o), _p(d), _p(v), ctypes.c_int64(Q), ctypes.byref(opt),
       _p(out), _p(cnt) if count else None)
    return (out, Counters(*cnt.tolist())) if count else out


def volume_render_weights(tree: Tree, origins, dirs, vdirs, opt: RenderOptions):
    """volume_render with tree._weight_accum set (rt_kernel.cu:266-267,309-311; svox.py:948-969):
    returns (out, weight_accum [n_internal, N, N, N] float64) -- per leaf slot, the sum of the
    compositing weights of the samples taken in that leaf."""
    assert tree.dtype == np.float32
    o, d, v = _rays(tree, origins, dirs, vdirs)
    Q = o.shape[0]
    out = np.zeros((Q, out_data_dim(opt, tree.K)), dtype=np.float32)
    wacc = np.zeros(tree.child.shape, dtype=np.float64)
    lib().svoxt_oracle_volume_render_weights_f32(*tree._args(), _p(o), _p(d), _p(v), ctypes.c_int64(Q),
                                                  ctypes.byref(opt), _p(out), _p(wacc))
    return out, wacc


def volume_render_backward(tree: Tree, origins, dirs, vdirs, opt: RenderOptions,
                           grad_output, want_abs=False):
    """Returns grad [M, K] float64 (and sum |contribution| if want_abs; want_abs="both": also the
    tighter scale that prices `accum` by the reference's own sequential addends, see
    trace_ray_backward in svoxt_oracle.cpp -- (grad, abs_sum, abs_sum_tight))."""
    o, d, v = _rays(tree, origins, dirs, vdirs)
    g = _c(grad_output, tree.dtype)
    Q = o.shape[0]
    assert g.shape[0] == Q
    grad = np.zeros((tree.M, tree.K), dtype=np.float64)
    absum = np.zeros((tree.M, tree.K), dtype=np.float64) if want_abs else None
    if want_abs == "both":
        assert tree.dtype == np.float32
        tight = np.zeros((tree.M, tree.K), dtype=np.float64)
        lib().svoxt_oracle_volume_render_backward_scales_f32(
            *tree._args(), _p(o), _p(d), _p(v), ctypes.c_int64(Q), ctypes.byref(opt),
            _p(g), ctypes.c_int(g.shape[1]), _p(grad), _p(absum), _p(tight))
        return grad, absum, tight
    fn = lib().svoxt_oracle_volume_render_backward_f32 if tree.dtype == np.float32 \
        else lib().svoxt_oracle_volume_render_backward_f64
    fn(*tree._args(), _p(o), _p(d), _p(v), ctypes.c_int64(Q), ctypes.byref(opt),
       _p(g), ctypes.c_int(g.shape[1]), _p(grad), _p(absum))
    return (grad, absum) if want_abs else grad


def opacity_render(tree: Tree, origins, dirs, vdirs, opt: RenderOptions):
    o, d, v = _rays(tree, origins, dirs, vdirs)
    Q = o.shape[0]
    out = np.zeros((Q, 1), dtype=np.float32)
    lib().svoxt_oracle_opacity_render_f32(*tree._args(False), _p(o), _p(d), _p(v),
                                          ctypes.c_int64(Q), ctypes.byref(opt), _p(out))
    return out


def render_depth(tree: Tree, origins, dirs, vdirs, opt: RenderOptions, count=False):
    o, d, v = _rays(tree, origins, dirs, vdirs)
    Q = o.shape[0]
    out = np.zeros((Q, 1), dtype=np.float32)
    cnt = np.zeros(5, dtype=np.int64)
    lib().svoxt_oracle_render_depth_f32(*tree._args(False), _p(o), _p(d), _p(v),
                                        ctypes.c_int64(Q), ctypes.byref(opt), _p(out),
                                        _p(cnt) if count else None)
    return (out, Counters(*cnt.tolist())) if count else out


def ray_steps(tree: Tree, origins, dirs, vdirs, opt: RenderOptions):
    """(leaf crossings, composited samples) per ray, int32 [Q] each (analysis aid)."""
    assert tree.dtype == np.float32
    o, d, v = _rays(tree, origins, dirs, vdirs)
    Q = o.shape[0]
    steps = np.zeros(Q, np.int32)
    active = np.zeros(Q, np.int32)
    lib().svoxt_oracle_ray_steps_f32(*tree._args(with_extra=False), _p(o), _p(d), _p(v), ctypes.c_int64(Q),
                                     ctypes.byref(opt), _p(steps), _p(active))
    return steps, active


def motion_render(tree: Tree, origins, dirs, vdirs, opt: RenderOptions):
    """motion_render (rt_kernel.cu:698-778, 1480-1504), float32 only.  Returns
    (joint distances [Q, J], depth [Q, 1], hit_point [Q, 3], data_idx [Q, 1] int64);
    tree.extra = joint positions [J, >=3]."""
    assert tree.dtype == np.float32 and tree.extra is not None and tree.extra.shape[1] >= 3
    o, d, v = _rays(tree, origins, dirs, vdirs)
    Q, J = o.shape[0], tree.extra.shape[0]
    out = np.zeros((Q, J), np.float32)
    depth = np.zeros((Q, 1), np.float32)
    hit = np.zeros((Q, 3), np.float32)
    idx = np.zeros((Q, 1), np.int64)
    lib().svoxt_oracle_motion_render_f32(*tree._args(), _p(o), _p(d), _p(v), ctypes.c_int64(Q),
                                         ctypes.byref(opt), _p(out), _p(depth), _p(hit), _p(idx))
    return out, depth, hit, idx


class Motion:
    """joint_features [n_joints, F], skinning_weights [M, B], joint_index [M, B] (TreeSpec fields)."""

    def __init__(self, joint_features, skinning_weights, joint_index, dtype=np.float32):
        self.jf = _c(joint_features, dtype)
        self.sw = _c(skinning_weights, dtype)
        self.ji = _c(joint_index, np.int32)
        assert self.sw.shape == self.ji.shape and self.jf.shape[1] <= 32
        assert self.ji.min() >= 0 and self.ji.max() < self.jf.shape[0]

    def astype(self, dtype):
        return Motion(self.jf, self.sw, self.ji, dtype)

    def _args(self):
        return [_p(self.jf), ctypes.c_int(self.jf.shape[0]), ctypes.c_int(self.jf.shape[1]),
                _p(self.sw), _p(self.ji), ctypes.c_int(self.ji.shape[1])]


def motion_feature_render(tree: Tree, motion: Motion, origins, dirs, vdirs, opt: RenderOptions):
    """motion_feature_render (rt_kernel.cu:886-981, 1525-1543): [Q, F]."""
    assert motion.jf.dtype == tree.dtype and motion.sw.shape[0] == tree.M
    o, d, v = _rays(tree, origins, dirs, vdirs)
    Q = o.shape[0]
    out = np.zeros((Q, motion.jf.shape[1]), tree.dtype)
    fn = lib().svoxt_oracle_motion_feature_render_f32 if tree.dtype == np.float32 \
        else lib().svoxt_oracle_motion_feature_render_f64
    fn(*tree._args(with_extra=False), *motion._args(), _p(o), _p(d), _p(v), ctypes.c_int64(Q),
       ctypes.byref(opt), _p(out))
    return out


def motion_feature_render_backward(tree: Tree, motion: Motion, origins, dirs, vdirs, opt: RenderOptions,
                                   grad_output, want_abs=False):
    """d/d joint_features of motion_feature_render: grad [n_joints, F] float64
    (see svoxt_oracle.cpp: the derivative of the forward; the reference's own
    backward, rt_kernel.cu:983-1061, reads an uninitialised local)."""
    assert motion.jf.dtype == tree.dtype and motion.sw.shape[0] == tree.M
    o, d, v = _rays(tree, origins, dirs, vdirs)
    g = _c(grad_output, tree.dtype)
    Q = o.shape[0]
    assert g.shape == (Q, motion.jf.shape[1])
    grad = np.zeros(motion.jf.shape, np.float64)
    absum = np.zeros(motion.jf.shape, np.float64) if want_abs else None
    fn = lib().svoxt_oracle_motion_feature_render_backward_f32 if tree.dtype == np.float32 \
        else lib().svoxt_oracle_motion_feature_render_backward_f64
    fn(*tree._args(with_extra=False), *motion._args(), _p(o), _p(d), _p(v), ctypes.c_int64(Q),
       ctypes.byref(opt), _p(g), _p(grad), _p(absum))
    return (grad, absum) if want_abs else grad


def query(tree: Tree, points):
    p = _c(points, np.float32)
    Q = p.shape[0]
    values = np.zeros((Q, tree.K), dtype=np.float32)
    node_ids = np.zeros(Q, dtype=np.int64)
    data_ids = np.zeros(Q, dtype=np.int64)
    lib().svoxt_oracle_query_f32(*tree._args(False), _p(p), ctypes.c_int64(Q),
                                 _p(values), _p(node_ids), _p(data_ids))
    return values, node_ids, data_ids


def query_backward(tree: Tree, points, grad_output):
    p = _c(points, np.float32)
    g = _c(grad_output, np.float32)
    grad = np.zeros((tree.M, tree.K), dtype=np.float64)
    lib().svoxt_oracle_query_backward_f32(ctypes.c_int64(tree.M), ctypes.c_int(tree.K),
                                          _p(tree.data), _p(tree.child), ctypes.c_int(tree.N),
                                          _p(tree.offset), _p(tree.scaling), _p(p),
                                          ctypes.c_int64(p.shape[0]), _p(g), _p(grad))
    return grad


def camera_rays(c2w, fx, fy, width, height, ndc=None):
    """(origins, dirs, vdirs) float32 [H*W, 3] of a pinhole camera, row-major:
    cam2world_ray + maybe_world2ndc (rt_kernel.cu:1153-1190).  c2w: [3,4] or [4,4];
    ndc: None or (ndc_width, ndc_height, ndc_focal)."""
    c = np.ascontiguousarray(np.asarray(c2w, dtype=np.float32)[:3, :4])
    n = int(width) * int(height)
    o = np.empty((n, 3), np.float32)
    d = np.empty((n, 3), np.float32)
    v = np.empty((n, 3), np.float32)
    nw, nh, nf = (-1, -1, 0.0) if ndc is None else ndc
    lib().svoxt_oracle_camera_rays(_p(c), ctypes.c_float(fx), ctypes.c_float(fy), int(width), int(height),
                                   int(nw), int(nh), ctypes.c_float(nf), _p(o), _p(d), _p(v))
    return o, d, v


def basis(format: int, basis_dim: int, dirs, extra=None):
    d = _c(dirs, np.float32)
    out = np.zeros((d.shape[0], basis_dim), dtype=np.float32)
    e = None if extra is None else _c(extra, np.float32)
    lib().svoxt_oracle_basis_f32(ctypes.c_int(format), ctypes.c_int(basis_dim), _p(e),
                                 ctypes.c_int(0 if e is None else e.shape[0]),
                                 ctypes.c_int(0 if e is None else e.shape[1]),
                                 _p(d), ctypes.c_int64(d.shape[0]), _p(out))
    return out


def algorithmic_bytes_forward(cnt: Counters, Q: int, K: int, C: int) -> int:
    """B_f = sum_r [36 + 4(C+1)] + sum_steps (4L + 4 + 4v + 4(K-1)a)."""
    return Q * (36 + 4 * (C + 1)) + 4 * cnt.levels + 4 * cnt.steps + 4 * cnt.valid \
        + 4 * (K - 1) * cnt.active


def algorithmic_bytes_backward(cnt: Counters, Q: int, M: int, K: int, C: int) -> int:
    """B_b (two re-marches, K-1 colour atomics + 1 sigma atomic
    per active sample, an atomic counted as 4 B read + 4 B write, zero-init)."""
    per_march = 4 * cnt.levels + 4 * cnt.steps + 4 * cnt.valid + 4 * (K - 1) * cnt.active
    return 4 * M * K + Q * (36 + 4 * (C + 1)) + 2 * per_march + cnt.active * (8 * (K - 1) + 8)
