"""CPU restatement of the reference's octree construction from a point cloud.

TEST INFRASTRUCTURE ONLY: imported by tests/,
never by the product.  numpy, float32 arithmetic where the reference's is.

The reference builds a frame's tree with (Artemis' usage of this package; the
pieces are helpers.py:101-109, svox.py:160-161, 488-560):

    tree = N3Tree(N=2, ...)                      # root only
    for _ in range(depth - 1):
        tree[points].refine()                    # query -> unique leaves -> one new node per leaf
    tree.construct_tree(points)                  # data[leaf of point i] = i

Functions, each following the reference lines it cites:

  descend         query_single_from_root           svox_t/csrc/include/common.cuh:38-42, 63-100
                  + transform_coord                common.cuh:45-51
  unique_leaves   the leaf_node list of query_vertical   svox_kernel.cu:240-324 (sorted by
                  packed id here; the reference's order comes from a float atomic counter)
  refine          N3Tree.refine with an explicit selector       svox.py:488-560
  construct       construct_tree_kernel            svox_kernel.cu:110-121 (smallest index wins
                  where the reference keeps an arbitrary writer)
  build_from_points   the loop above

Pinning: `refine` is checked against tables the reference's own N3Tree.refine
produced for the same selectors (tests/golden/topology_points_*.npz, made by
tests/golden/make_golden.py, which drives the reference with this module's
`descend` / `unique_leaves`).
"""
from __future__ import annotations

import numpy as np

EMPTY_INDEX = 1410065408          # int(1e10) wrapped to int32 (svox.py:124)


class Topology:
    """child [cap, N, N, N] int32, data [cap, N, N, N] int32, parent_depth [cap, 2] int32, n."""

    def __init__(self, N=2, reserve=1):
        self.N = N
        self.child = np.zeros((reserve, N, N, N), dtype=np.int32)
        self.data = np.full((reserve, N, N, N), EMPTY_INDEX, dtype=np.int32)
        self.parent_depth = np.zeros((reserve, 2), dtype=np.int32)
        self.n = 1

    def _grow(self, rows):
        N = self.N
        if rows <= self.child.shape[0]:
            return
        add = rows - self.child.shape[0]
        self.child = np.concatenate([self.child, np.zeros((add, N, N, N), np.int32)])
        self.data = np.concatenate([self.data, np.full((add, N, N, N), EMPTY_INDEX, np.int32)])
        self.parent_depth = np.concatenate([self.parent_depth, np.zeros((add, 2), np.int32)])

    def arrays(self):
        n = self.n
        return self.child[:n].copy(), self.data[:n].copy(), self.parent_depth[:n].copy()


def descend(topo: Topology, points, offset, scaling):
    """Packed leaf id node*N^3 + u*N^2 + v*N + w of every point (common.cuh:63-100)."""
    N = topo.N
    f32 = np.float32
    p = (np.asarray(offset, f32)[None, :] + np.asarray(scaling, f32)[None, :] * np.asarray(points, f32)).astype(f32)
    hi = f32(1.0 - 1e-6)                                       # the upper clamp is formed in double (:40)
    p = np.minimum(np.maximum(p, f32(0.0)), hi)
    node = np.zeros(p.shape[0], dtype=np.int64)
    packed = np.zeros(p.shape[0], dtype=np.int64)
    live = np.ones(p.shape[0], dtype=bool)
    child = topo.child.reshape(-1)
    while live.any():
        p[live] = p[live] * f32(N)
        uvw = np.floor(p[live])
        p[live] = p[live] - uvw
        uvw = uvw.astype(np.int64)
        slot = ((node[live] * N + uvw[:, 0]) * N + uvw[:, 1]) * N + uvw[:, 2]
        skip = child[slot].astype(np.int64)
        packed[live] = slot
        idx = np.nonzero(live)[0]
        node[idx] += skip
        live[idx[skip == 0]] = False
    return packed


def unique_leaves(topo: Topology, packed):
    """[U, 4] (node, u, v, w), increasing packed id (svox_kernel.cu:240-259, 304-320)."""
    N = topo.N
    u = np.unique(packed)
    out = np.empty((u.shape[0], 4), dtype=np.int64)
    tmp = u.copy()
    for i in (3, 2, 1):
        out[:, i] = tmp % N
        tmp //= N
    out[:, 0] = tmp
    return out


def refine(topo: Topology, leaf_node):
    """One round of N3Tree.refine(sel=(*leaf_node.T,), leaf_node=leaf_node) (svox.py:520-556)."""
    N = topo.N
    k = leaf_node.shape[0]
    if k == 0:
        return
    filled = topo.n
    topo._grow(filled + k)
    sel = tuple(leaf_node.T)
    new_idx = np.arange(filled, filled + k, dtype=np.int64)
    topo.child[sel] = (new_idx - leaf_node[:, 0]).astype(np.int32)                 # :535-536
    topo.data[filled:filled + k] = topo.data[sel][:, None, None, None]              # :537-538
    packed = ((leaf_node[:, 0] * N + leaf_node[:, 1]) * N + leaf_node[:, 2]) * N + leaf_node[:, 3]
    topo.parent_depth[filled:filled + k, 0] = packed.astype(np.int32)               # :539 (_pack_index)
    topo.parent_depth[filled:filled + k, 1] = topo.parent_depth[leaf_node[:, 0], 1] + 1   # :540-541
    topo.n = filled + k


def construct(topo: Topology, points, offset, scaling):
    """data[leaf of point i] = i (svox_kernel.cu:110-121); smallest i per leaf."""
    packed = descend(topo, points, offset, scaling)
    flat = topo.data.reshape(-1)
    order = np.arange(packed.shape[0], dtype=np.int64)
    best = np.full(flat.shape[0], np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(best, packed, order)
    hit = best != np.iinfo(np.int64).max
    flat[hit] = best[hit].astype(np.int32)


def build_from_points(points, offset, scaling, depth):
    """(child, data, parent_depth) of the depth-`depth` tree of a point cloud."""
    topo = Topology(N=2)
    for _ in range(depth - 1):
        refine(topo, unique_leaves(topo, descend(topo, points, offset, scaling)))
    construct(topo, points, offset, scaling)
    return topo.arrays()
