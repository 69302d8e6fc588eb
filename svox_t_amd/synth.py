"""Deterministic synthetic inputs for the volume-render hot path.

The reference ships no scenes, checkpoints or benchmark inputs, so the
workloads BASELINE.json names are generated here (recipe: SURVEY.md 8(d)):

* ``shell_tree(D)``   -- an N=2 tree refined wherever a leaf box meets the
  spherical shell 0.34 <= |x - 0.5| <= 0.36, finest leaf side 2**-D.  The
  arrays have exactly the layout ``N3Tree.refine(sel=...)`` produces
  (reference: svox_t/svox.py:488-560): children are appended in
  ``(node, x, y, z)`` lexicographic order, ``child`` holds *relative* offsets,
  ``parent_depth[:, 0]`` the packed parent slot, ``[:, 1]`` the depth.
* ``shell_features``  -- colour columns N(0,1), sigma = 10**U(0, 2.5) with
  10 % of the rows set to -1 (exercises the sigma <= 0 skip).
* ``pinhole_rays``    -- camera on a circle of radius 1.6 about the cube
  centre; per-pixel directions follow the reference's ``cam2world_ray``
  (svox_t/csrc/rt_kernel.cu:1153-1166), row-major pixel order.

Everything here is host-side numpy/torch on CPU; it feeds the HIP path, it is
not part of it.
"""
from __future__ import annotations

import math
from typing import NamedTuple

import numpy as np
import torch

# int(1e10) wrapped to int32, the reference's "empty leaf" fill value
# (svox_t/svox.py:124).  Any index >= features.size(0) means "empty"
# (svox_t/csrc/rt_kernel.cu:269); this particular value is kept for fidelity.
EMPTY_SENTINEL = 1410065408

SHELL_RMIN = 0.34
SHELL_RMAX = 0.36


class ShellTree(NamedTuple):
    child: np.ndarray         # int32 [n_internal, 2, 2, 2]
    data: np.ndarray          # int32 [n_internal, 2, 2, 2, 1]
    parent_depth: np.ndarray  # int32 [n_internal, 2]
    n_internal: int
    n_features: int           # M: number of occupied finest-level leaves
    depth: int                # D


def _box_hits_shell(lo: np.ndarray, size: float) -> np.ndarray:
    """lo: [n, 3] lower corners (float64); True where the box meets the shell."""
    c = 0.5
    hi = lo + size
    dmin = np.sqrt((np.maximum(np.maximum(lo - c, c - hi), 0.0) ** 2).sum(-1))
    dmax = np.sqrt((np.maximum(np.abs(lo - c), np.abs(hi - c)) ** 2).sum(-1))
    return (dmin <= SHELL_RMAX) & (dmax >= SHELL_RMIN)


_SLOT_OFFS = np.stack(np.meshgrid(np.arange(2), np.arange(2), np.arange(2),
                                  indexing="ij"), -1).reshape(8, 3).astype(np.float64)


def shell_tree(depth: int) -> ShellTree:
    """Build the depth-``depth`` shell tree (finest leaf side 2**-depth)."""
    assert depth >= 1
    # Per level: node lower corners.  Level 0 = root (its 8 slots have side 1/2).
    corners = [np.zeros((1, 3))]
    child_lvls = []
    parent_lvls = [np.array([[-1, -1]], dtype=np.int64)]  # root: parent packed id 0 in the reference, see below
    first_id = [0]
    n_nodes = 1
    last_hits = None
    for lvl in range(1, depth + 1):
        side = 0.5 ** lvl                      # slot side of nodes at level lvl-1
        par = corners[-1]                      # [n, 3]
        n = par.shape[0]
        slot_lo = (par[:, None, :] + _SLOT_OFFS[None] * side).reshape(-1, 3)
        hits = _box_hits_shell(slot_lo, side)  # [n*8] in (node, x, y, z) order
        if lvl == depth:
            last_hits = hits
            child_lvls.append(np.zeros(n * 8, dtype=np.int32))
            break
        new_rel = np.cumsum(hits) - 1                      # rank among refined slots
        new_ids = n_nodes + new_rel                        # absolute ids of the new nodes
        node_of_slot = first_id[-1] + np.arange(n * 8) // 8
        ch = np.where(hits, new_ids - node_of_slot, 0).astype(np.int32)
        child_lvls.append(ch)
        sel = np.nonzero(hits)[0]
        packed_parent = node_of_slot[sel] * 8 + (sel % 8)
        parent_lvls.append(np.stack([packed_parent, np.full(sel.shape, lvl, dtype=np.int64)], -1))
        corners.append(slot_lo[sel])
        first_id.append(n_nodes)
        n_nodes += sel.size
    child = np.concatenate(child_lvls).reshape(-1, 2, 2, 2)
    assert child.shape[0] == n_nodes
    pd = np.concatenate(parent_lvls).astype(np.int32)
    pd[0] = (0, 0)                                         # root row is all zeros in the reference
    data = np.full(n_nodes * 8, EMPTY_SENTINEL, dtype=np.int32)
    n_last = last_hits.size
    feat_idx = np.cumsum(last_hits) - 1
    tail = data[n_nodes * 8 - n_last:]
    tail[last_hits] = feat_idx[last_hits].astype(np.int32)
    M = int(last_hits.sum())
    return ShellTree(child.astype(np.int32), data.reshape(-1, 2, 2, 2, 1), pd,
                     n_nodes, M, depth)


def shell_features(M: int, K: int, seed: int = 0) -> torch.Tensor:
    """Feature table [M, K]: K-1 colour/feature columns ~ N(0,1); the last
    column is sigma = 10**U(0, 2.5), with 10 % of the rows set to -1."""
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(M, K, generator=g, dtype=torch.float32)
    sigma = torch.pow(10.0, torch.rand(M, generator=g, dtype=torch.float32) * 2.5)
    neg = torch.rand(M, generator=g) < 0.1
    sigma[neg] = -1.0
    feats[:, K - 1] = sigma
    return feats


def camera_pose(azimuth_deg: float = 30.0, elevation_deg: float = 20.0,
                radius: float = 1.6, center=(0.5, 0.5, 0.5)) -> np.ndarray:
    """c2w [3, 4] (float64) of a camera on a circle about ``center`` looking at
    it; -z is the viewing direction, +y up (OpenGL convention, as the
    reference's cam2world_ray assumes)."""
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    c = np.asarray(center, dtype=np.float64)
    eye = c + radius * np.array([math.cos(el) * math.cos(az),
                                 math.cos(el) * math.sin(az),
                                 math.sin(el)])
    fwd = c - eye
    fwd /= np.linalg.norm(fwd)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    true_up = np.cross(right, fwd)
    c2w = np.stack([right, true_up, -fwd, eye], axis=1)   # columns x, y, z, origin
    return c2w


def pinhole_rays(width: int, height: int, c2w: np.ndarray | None = None,
                 fx: float | None = None, fy: float | None = None):
    """Row-major [H*W, 3] float32 origins / dirs / viewdirs (= dirs).

    Follows cam2world_ray (svox_t/csrc/rt_kernel.cu:1153-1166):
    x = (ix - W/2)/fx, y = -(iy - H/2)/fy, dir = c2w[:3,:3] @ normalize(x, y, -1).
    """
    if c2w is None:
        c2w = camera_pose()
    if fx is None:
        fx = 1111.111 * width / 800.0          # svox_t/renderer.py:310 default at 800 px
    if fy is None:
        fy = fx
    ix = np.arange(width, dtype=np.float64)
    iy = np.arange(height, dtype=np.float64)
    x = (ix - 0.5 * width) / fx
    y = -(iy - 0.5 * height) / fy
    X, Y = np.meshgrid(x, y, indexing="xy")    # [H, W]
    Z = np.sqrt(X * X + Y * Y + 1.0)
    cam = np.stack([X / Z, Y / Z, -1.0 / Z], -1).reshape(-1, 3)
    R = np.asarray(c2w)[:3, :3]
    dirs = (cam @ R.T).astype(np.float32)
    origins = np.broadcast_to(np.asarray(c2w)[:3, 3].astype(np.float32), dirs.shape).copy()
    return (torch.from_numpy(origins), torch.from_numpy(dirs), torch.from_numpy(dirs.copy()))


def grad_output(Q: int, cols: int, seed: int = 1) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(Q, cols, generator=g, dtype=torch.float32)
