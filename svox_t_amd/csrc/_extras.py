"""The operators of `svox_t.csrc` around the render path: point query (svox_kernel.cu:45-94, 240-324), the roofline
counters, the motion variants (rt_kernel.cu:698-1061), point skinning (svox_kernel.cu:123-211), octree construction
(svox.py:160-161, 488-560) -- marshalling only, one C-ABI call each (two for the builder)."""
from __future__ import annotations

import ctypes

import torch

from ._abi import _CMotion, _CTree, _lib
from ._marshal import (RaysSpec, RenderOptions, TreeSpec, _ACCEL_CACHE, _drop_accel, _call, _check_input, _numel, _pack_opts, _pack_rays,
                       _on, _pack_tree, _pack_tree_accel, _ptr, _stream)

def _check_indices(indices):
    """check_indices (svox_kernel.cu:36-40)."""
    _check_input(indices, "indices")
    if indices.dim() != 2:
        raise RuntimeError("indices must be 2-D")
    if not indices.is_floating_point():
        raise RuntimeError("indices must be floating point")
    if indices.dtype != torch.float32 or indices.shape[1] != 3:
        raise RuntimeError("indices must be float32 [Q, 3]")


def query_vertical(tree: TreeSpec, indices: torch.Tensor):
    """svox_kernel.cu:274-324.  Returns (values [Q,K], node_ids [Q] int64,
    data_ids [Q] int64, leaf_node [U,4] int64).

    Differences from the reference, all where its result is undefined:
    rows of `values` for empty leaves are zeros (reference: uninitialised,
    :282), `data_ids` is -1 there, and `leaf_node` is sorted by packed leaf id
    (reference: order set by a float atomic counter, :260-269)."""
    ct = _pack_tree(tree)
    _check_indices(indices)
    dev = indices.device
    Q = indices.shape[0]
    N = ct.N
    with _on(dev):
        values = torch.empty((Q, ct.K), dtype=torch.float32, device=dev)
        node_ids = torch.empty((Q,), dtype=torch.int64, device=dev)
        data_ids = torch.empty((Q,), dtype=torch.int64, device=dev)
        mask = torch.zeros((ct.n_internal * N * N * N,), dtype=torch.uint8, device=dev)
        _call("svoxt_query_fwd", ctypes.byref(ct), _ptr(indices), Q, _ptr(values), _ptr(node_ids),
              _ptr(data_ids), _ptr(mask), _stream(dev))
        n_slots = mask.numel()
        cap = min(Q, n_slots)
        leaf_buf = torch.empty((cap, 4), dtype=torch.int64, device=dev)
        count = torch.empty((1,), dtype=torch.int64, device=dev)
        ws = torch.empty((_lib.svoxt_query_leaves_workspace_bytes(n_slots),), dtype=torch.uint8, device=dev)
        _call("svoxt_query_leaves", _ptr(mask), n_slots, N, _ptr(leaf_buf), _ptr(count), _ptr(ws), _stream(dev))
        leaf_node = leaf_buf[:int(count.item())]         # host sync, like the reference's .item() (:312)
    return values, node_ids, data_ids, leaf_node


def query_vertical_backward(tree: TreeSpec, indices: torch.Tensor,
                            grad_output: torch.Tensor) -> torch.Tensor:
    """svox_kernel.cu:380-402."""
    ct = _pack_tree(tree)
    _check_indices(indices)
    _check_input(grad_output, "grad_output")
    if grad_output.dtype != torch.float32 or tuple(grad_output.shape) != (indices.shape[0], ct.K):
        raise RuntimeError("grad_output must be float32 [Q, K]")
    dev = indices.device
    with _on(dev):
        grad = torch.empty((ct.M, ct.K), dtype=torch.float32, device=dev)
        _call("svoxt_query_bwd", ctypes.byref(ct), _ptr(indices), indices.shape[0],
              _ptr(grad_output), _ptr(grad), _stream(dev))
    return grad


def count_forward(tree: TreeSpec, rays: RaysSpec, opt: RenderOptions) -> torch.Tensor:
    """Roofline counters (not in the reference): int64 [5] on the device =
    (rays hitting the cube, leaf crossings, child words read, valid leaves,
    composited samples).  See SURVEY.md 8(d)."""
    ct, cr, co = _pack_tree(tree), _pack_rays(rays), _pack_opts(opt)
    dev = tree.features.device
    with _on(dev):
        counters = torch.zeros((5,), dtype=torch.int64, device=dev)
        _call("svoxt_count_fwd", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co),
              _ptr(counters), _stream(dev))
    return counters


def count_touched(tree: TreeSpec, rays: RaysSpec, opt: RenderOptions):
    """Roofline instrumentation (include/svoxt.h, svoxt_count_touched): what one forward march of
    the batch touches.  Returns a dict of counts: feature rows read by the forward (valid leaves)
    and again by the backward (composited samples), grid cells and (child, data) pairs (or child /
    data words without the grid), and the most leaf crossings of any ray."""
    ct, cr, co = _pack_tree_accel(tree), _pack_rays(rays), _pack_opts(opt)
    dev = tree.features.device
    n_slots = ct.n_internal * ct.N ** 3
    with _on(dev):
        rows = torch.zeros((2 * ct.M,), dtype=torch.uint8, device=dev)
        n_cells = (1 << (3 * (ct.accel_log2 & 0xff))) if ct.accel else 0
        tmask = torch.zeros(((n_cells + n_slots) if ct.accel else 2 * n_slots,), dtype=torch.uint8, device=dev)
        longest = torch.zeros((1,), dtype=torch.int64, device=dev)
        _call("svoxt_count_touched", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co), _ptr(rows), _ptr(tmask),
              _ptr(longest), _stream(dev))
        first = n_cells if ct.accel else n_slots
        res = {"rows_valid": int(rows[:ct.M].sum(dtype=torch.int64)), "rows_composited": int(rows[ct.M:].sum(dtype=torch.int64)),
               "longest_ray_crossings": int(longest.item()), "accel": bool(ct.accel)}
        a, b = int(tmask[:first].sum(dtype=torch.int64)), int(tmask[first:].sum(dtype=torch.int64))
        res.update({"grid_cells": a, "node_pairs": b} if ct.accel else {"child_words": a, "data_words": b})
    return res


class bwd_counters:
    """`with bwd_counters() as c: ...backward...; c.read()` -> (64-byte atomic requests, (tile, pass, row)
    groups) of the one-kernel per-tile backwards run inside (svoxt_set_bwd_counters)."""

    def __init__(self, device):
        self.buf = torch.zeros((2,), dtype=torch.int64, device=device)

    def __enter__(self):
        _call("svoxt_set_bwd_counters", _ptr(self.buf))
        return self

    def __exit__(self, *exc):
        _call("svoxt_set_bwd_counters", None)

    def read(self):
        return tuple(int(v) for v in self.buf.cpu().tolist())


class bwd_check:
    """`with bwd_check(dev) as c: ...backward...; c.read()` -> ({site: violations}, tiles worked on): the per-tile
    backwards run inside take their CHECKED instances -- every LDS / pool / table index compared with its extent
    (svoxt_set_bwd_check; the sites are listed at grad_fused_kernel / grad_wide_kernel)."""

    def __init__(self, device):
        self.buf = torch.zeros((32,), dtype=torch.int64, device=device)

    def __enter__(self):
        _call("svoxt_set_bwd_check", _ptr(self.buf))
        return self

    def __exit__(self, *exc):
        _call("svoxt_set_bwd_check", None)

    def read(self):
        w = [int(v) for v in self.buf.cpu().tolist()]
        return {site: n for site, n in enumerate(w[2:31]) if n}, w[31]


def motion_render(tree: TreeSpec, rays: RaysSpec, opt: RenderOptions):
    """rt_kernel.cu:1480-1504.  Returns (joint distances [Q, J], depth [Q, 1],
    hit_point [Q, 3], data_idx [Q, 1] int64); J = tree.extra_data.shape[0]."""
    ct, cr, co = _pack_tree_accel(tree), _pack_rays(rays), _pack_opts(opt)
    if not _numel(tree.extra_data):
        raise RuntimeError("motion_render needs extra_data [n_joints, >= 3] (the joint positions)")
    dev = tree.features.device
    with _on(dev):
        out = torch.empty((cr.Q, ct.extra_rows), dtype=torch.float32, device=dev)
        depth = torch.empty((cr.Q, 1), dtype=torch.float32, device=dev)
        hit = torch.empty((cr.Q, 3), dtype=torch.float32, device=dev)
        idx = torch.empty((cr.Q, 1), dtype=torch.int64, device=dev)
        _call("svoxt_motion_render", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co),
              _ptr(out), _ptr(depth), _ptr(hit), _ptr(idx), _stream(dev))
    return out, depth, hit, idx


def _pack_motion(tree: TreeSpec, ct: _CTree) -> _CMotion:
    jf, sw, ji = tree.joint_features, tree.skinning_weights, tree.joint_index
    for nm, x in (("joint_features", jf), ("skinning_weights", sw), ("joint_index", ji)):
        if not _numel(x):
            raise RuntimeError(f"motion_feature_render needs {nm}")
        _check_input(x, nm)
    if jf.dtype != torch.float32 or jf.dim() != 2 or sw.dtype != torch.float32 or sw.dim() != 2:
        raise RuntimeError("joint_features / skinning_weights must be float32 and 2-D")
    if ji.dtype != torch.int32 or ji.shape != sw.shape or sw.shape[0] != ct.M:
        raise RuntimeError("joint_index must be int32 with the shape of skinning_weights, [M, n_bind]")
    return _CMotion(jf.data_ptr(), jf.shape[0], jf.shape[1], sw.data_ptr(), ji.data_ptr(), sw.shape[1])


def _motion_workspace(ct: _CTree, cm: _CMotion, dev) -> torch.Tensor:
    nbytes = _lib.svoxt_motion_workspace_bytes(ct.M, cm.feature_dim)
    if nbytes < 0:
        raise RuntimeError("joint feature dim must be in [1, 32] (the reference's tmp_data_dim)")
    return torch.empty((nbytes,), dtype=torch.uint8, device=dev)


def motion_feature_render(tree: TreeSpec, rays: RaysSpec, opt: RenderOptions) -> torch.Tensor:
    """rt_kernel.cu:1525-1543: [Q, joint_features.shape[1]]."""
    ct, cr, co = _pack_tree_accel(tree), _pack_rays(rays), _pack_opts(opt)
    cm = _pack_motion(tree, ct)
    dev = tree.features.device
    with _on(dev):
        out = torch.empty((cr.Q, cm.feature_dim), dtype=torch.float32, device=dev)
        ws = _motion_workspace(ct, cm, dev)
        _call("svoxt_motion_feature_render_fwd", ctypes.byref(ct), ctypes.byref(cm), ctypes.byref(cr),
              ctypes.byref(co), _ptr(out), _ptr(ws), ws.numel(), _stream(dev))
    return out


def motion_feature_render_backward(tree: TreeSpec, rays: RaysSpec, opt: RenderOptions,
                                   grad_output: torch.Tensor) -> torch.Tensor:
    """rt_kernel.cu:1546-1572: gradient wrt joint_features, [n_joints, F] (the
    derivative of the forward; the reference's kernel is defective, include/svoxt.h)."""
    ct, cr, co = _pack_tree_accel(tree), _pack_rays(rays), _pack_opts(opt)
    cm = _pack_motion(tree, ct)
    _check_input(grad_output, "grad_output")
    if grad_output.dtype != torch.float32 or tuple(grad_output.shape) != (cr.Q, cm.feature_dim):
        raise RuntimeError("grad_output must be float32 [Q, joint feature dim]")
    dev = tree.features.device
    with _on(dev):
        grad = torch.empty((cm.n_joints, cm.feature_dim), dtype=torch.float32, device=dev)
        ws = _motion_workspace(ct, cm, dev)
        _call("svoxt_motion_feature_render_bwd", ctypes.byref(ct), ctypes.byref(cm), ctypes.byref(cr),
              ctypes.byref(co), _ptr(grad_output), _ptr(grad), _ptr(ws), ws.numel(), _stream(dev))
    return grad


def _check_warp(matrices, indices, skinning_weights, joint_index):
    _check_indices(indices)
    for nm, x in (("matrices", matrices), ("skinning_weights", skinning_weights), ("joint_index", joint_index)):
        _check_input(x, nm)
    if matrices.dtype != torch.float32 or matrices.dim() != 3 or tuple(matrices.shape[1:]) != (4, 4):
        raise RuntimeError("matrices must be float32 [n_joints, 4, 4]")
    Q = indices.shape[0]
    if skinning_weights.dtype != torch.float32 or skinning_weights.dim() != 2 or skinning_weights.shape[0] != Q:
        raise RuntimeError("skinning_weights must be float32 [Q, n_bind]")
    if joint_index.dtype != torch.int32 or joint_index.shape != skinning_weights.shape:
        raise RuntimeError("joint_index must be int32 with the shape of skinning_weights")
    return Q, matrices.shape[0], skinning_weights.shape[1]


def warp_vertices(matrices: torch.Tensor, indices: torch.Tensor, skinning_weights: torch.Tensor,
                  joint_index: torch.Tensor):
    """svox_kernel.cu:354-378: linear blend skinning of points.  Returns
    (vertices_out [Q, 3], matrix_out [Q, 4, 4])."""
    Q, J, B = _check_warp(matrices, indices, skinning_weights, joint_index)
    dev = indices.device
    with _on(dev):
        vout = torch.empty((Q, 3), dtype=torch.float32, device=dev)
        mout = torch.empty((Q, 4, 4), dtype=torch.float32, device=dev)
        _call("svoxt_warp_vertices", _ptr(matrices), J, _ptr(indices), Q, _ptr(skinning_weights),
              _ptr(joint_index), B, _ptr(vout), _ptr(mout), _stream(dev))
    return [vout, mout]


def warp_vertices_backward(matrices: torch.Tensor, indices: torch.Tensor, skinning_weights: torch.Tensor,
                           joint_index: torch.Tensor, indices_grad_out: torch.Tensor,
                           matrices_grad_out: torch.Tensor):
    """svox_kernel.cu:404-436.  Returns [grad_indices [Q, 3], grad_matrices [n_joints, 4, 4],
    grad_skinning_weights [Q, n_bind]]."""
    Q, J, B = _check_warp(matrices, indices, skinning_weights, joint_index)
    _check_input(indices_grad_out, "indices_grad_out")
    _check_input(matrices_grad_out, "matrices_grad_out")
    if indices_grad_out.dtype != torch.float32 or tuple(indices_grad_out.shape) != (Q, 3) or \
            matrices_grad_out.dtype != torch.float32 or tuple(matrices_grad_out.shape) != (Q, 4, 4):
        raise RuntimeError("gradients must be float32 [Q, 3] and [Q, 4, 4]")
    dev = indices.device
    with _on(dev):
        gi = torch.empty((Q, 3), dtype=torch.float32, device=dev)
        gm = torch.empty((J, 4, 4), dtype=torch.float32, device=dev)
        gs = torch.empty((Q, B), dtype=torch.float32, device=dev)
        _call("svoxt_warp_vertices_bwd", _ptr(matrices), J, _ptr(indices), Q, _ptr(skinning_weights),
              _ptr(joint_index), B, _ptr(indices_grad_out), _ptr(matrices_grad_out), _ptr(gi), _ptr(gm),
              _ptr(gs), _stream(dev))
    return [gi, gm, gs]


def refine_leaves(child: torch.Tensor, data: torch.Tensor, parent_depth: torch.Tensor, filled: int,
                  leaf_node: torch.Tensor, node_id: torch.Tensor = None) -> None:
    """The table updates of N3Tree.refine for the leaves in `leaf_node` [U, 4] int64
    (svox.py:535-546), in place, as one kernel (not an entry of the reference's
    extension, which does this with tensor ops).  The tables must have room for
    filled + U nodes."""
    for nm, x in (("child", child), ("data", data), ("parent_depth", parent_depth), ("leaf_node", leaf_node)):
        _check_input(x, nm)
    if leaf_node.dtype != torch.int64 or leaf_node.dim() != 2 or leaf_node.shape[1] != 4:
        raise RuntimeError("leaf_node must be int64 [U, 4]")
    if child.dtype != torch.int32 or data.dtype != torch.int32 or parent_depth.dtype != torch.int32:
        raise RuntimeError("child / data / parent_depth must be int32")
    if node_id is not None:
        _check_input(node_id, "node_id")
        if node_id.dtype != torch.int32 or node_id.numel() != leaf_node.shape[0]:
            raise RuntimeError("node_id must be int32 [U]")
    dev = child.device
    with _on(dev):
        _call("svoxt_refine", _ptr(leaf_node), leaf_node.shape[0], child.shape[1], int(filled), child.shape[0],
              _ptr(child), _ptr(data), _ptr(parent_depth), _ptr(node_id), _stream(dev))
    for t in (child, data, parent_depth):
        torch.autograd.graph.increment_version(t)
    _drop_accel(child)


def construct_tree(tree: TreeSpec, indices: torch.Tensor) -> None:
    """svox_kernel.cu:341-352: data[leaf containing point i] = i, in place on
    `tree.data`.  Where several points share a leaf the smallest index is kept
    (the reference keeps whichever thread wrote last)."""
    ct = _pack_tree(tree)
    _check_indices(indices)
    dev = indices.device
    with _on(dev):
        _call("svoxt_construct_tree", ctypes.byref(ct), _ptr(indices), indices.shape[0], _stream(dev))
    # tree.data was written behind torch's back: tell the version counter (the
    # acceleration-grid cache keys on it) and drop any grid built from the old words
    torch.autograd.graph.increment_version(tree.data)
    _drop_accel(tree.child)


def build_octree(points: torch.Tensor, offset: torch.Tensor, scaling: torch.Tensor, depth: int,
                 empty_index: int, reserve: int = 0):
    """Octree of a point cloud in one pipeline (not an entry of the reference's
    extension; it stands for `depth - 1` rounds of `tree[points].refine()` on a
    fresh N = 2 tree followed by `construct_tree(points)`, include/svoxt.h).

    Returns (child [n + reserve, 2, 2, 2] int32, data [n + reserve, 2, 2, 2, 1] int32,
    parent_depth [n + reserve, 2] int32, n): the first n rows are the tree, the
    `reserve` rows after them are initialised like unused rows of an N3Tree."""
    _check_indices(points)
    for name, x in (("offset", offset), ("scaling", scaling)):
        _check_input(x, name)
        if x.dtype != torch.float32 or x.numel() != 3:
            raise RuntimeError(f"{name} must be float32 [3]")
    dev = points.device
    P = points.shape[0]
    with _on(dev):
        nbytes = _lib.svoxt_build_workspace_bytes(int(depth))
        if nbytes < 0:
            raise RuntimeError("build_octree: depth must be in [1, 10]")
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        count = torch.empty((1,), dtype=torch.int64, device=dev)
        _call("svoxt_build_count", _ptr(points), P, _ptr(offset), _ptr(scaling), int(depth),
              _ptr(ws), nbytes, _ptr(count), _stream(dev))
        n = int(count.item())                 # the one host read: sizes the tables
        rows = n + int(reserve)
        child = torch.empty((rows, 2, 2, 2), dtype=torch.int32, device=dev)
        data = torch.empty((rows, 2, 2, 2, 1), dtype=torch.int32, device=dev)
        parent_depth = torch.empty((rows, 2), dtype=torch.int32, device=dev)
        if reserve > 0:
            child[n:].zero_()
            data[n:].fill_(int(empty_index))
            parent_depth[n:].zero_()
        _call("svoxt_build_emit", _ptr(points), P, _ptr(offset), _ptr(scaling), int(depth),
              _ptr(ws), nbytes, _ptr(child), _ptr(data), _ptr(parent_depth), n, int(empty_index),
              _stream(dev))
    return child, data, parent_depth, n


# ---------------------------------------------------------------------------
# Entry points of svox_t.csrc that are outside this project's hot path
# (SURVEY.md section 2).  They exist so a caller gets a clear error, not an
# AttributeError.
# ---------------------------------------------------------------------------

def _out_of_scope(name):
    def fn(*_a, **_k):
        raise NotImplementedError(
            f"svox_t_amd.csrc.{name}: outside the accelerated hot path "
            "(volume_render / opacity / depth / query / construct_tree); see SURVEY.md section 2")
    fn.__name__ = name
    return fn


for _n in ("assign_vertical", "p2v", "p2v_backward",
           "calc_corners", "grid_weight_render", "quantize_median_cut"):
    globals()[_n] = _out_of_scope(_n)
