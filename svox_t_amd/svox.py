"""N3Tree: host-side owner of the sparse N^3-tree the HIP kernels traverse.

Mirrors the part of the reference's `svox_t.N3Tree` (svox_t/svox.py:78-158,
:216-285, :488-560, :829-925) that the volume-render hot path touches: the
buffers and their layout, `refine`, the point query with its autograd bridge,
and `_spec`, which packs the tensors for the operator boundary.

Layout (identical to the reference so trees interchange):
    features      float32 [M, data_dim]          nn.Parameter, the leaf feature table
    data          int32   [cap, N, N, N, 1]      per leaf slot: row of `features`; >= M means empty
    child         int32   [cap, N, N, N]         per slot: offset to the child node, 0 = leaf
    parent_depth  int32   [cap, 2]               packed parent slot, depth
    invradius, offset float32 [3]                world -> [0,1]^3 : p' = offset + invradius * p
"""
from __future__ import annotations


import torch
from torch import autograd, nn

from svox_t_amd.helpers import DataFormat, N3TreeView, _get_c_extension

_C = _get_c_extension()

# int(1e10) as int32: the fill value of `data` in the reference (svox.py:124).
EMPTY_INDEX = 1410065408


class _QueryVerticalFunction(autograd.Function):
    """svox_t/svox.py:38-56: gradient flows to argument 0 (the feature table) only."""

    @staticmethod
    def forward(ctx, data, tree_spec, indices):
        out, node_ids, data_ids, leaf_node = _C.query_vertical(tree_spec, indices)
        ctx.mark_non_differentiable(node_ids, data_ids, leaf_node)
        ctx.tree_spec = tree_spec
        ctx.save_for_backward(indices)
        return out, node_ids, data_ids, leaf_node

    @staticmethod
    def backward(ctx, grad_out, *_unused):
        if ctx.needs_input_grad[0]:
            return _C.query_vertical_backward(ctx.tree_spec, ctx.saved_tensors[0],
                                              grad_out.contiguous()), None, None
        return None, None, None


class _WarpVerticalFunction(autograd.Function):
    """svox_t/svox.py:58-76.  As there, gradients flow only when the
    transformation matrices require one; then all three inputs get theirs."""

    @staticmethod
    def forward(ctx, transformation_matrix, coordinates, skinning_weights, joint_index):
        vertices, matrices = _C.warp_vertices(transformation_matrix, coordinates, skinning_weights, joint_index)
        ctx.save_for_backward(transformation_matrix, coordinates, skinning_weights, joint_index)
        return vertices, matrices

    @staticmethod
    def backward(ctx, vertices_grad_out, matrices_grad_out):
        if ctx.needs_input_grad[0]:
            grad_indices, grad_matrices, grad_skinning_weights = _C.warp_vertices_backward(
                *ctx.saved_tensors, vertices_grad_out.contiguous(), matrices_grad_out.contiguous())
            return grad_matrices, grad_indices, grad_skinning_weights, None
        return None, None, None, None


def get_transformation_matrix(src_pose, tgt_pose):
    """svox_t/svox.py:971-972."""
    return torch.matmul(tgt_pose, torch.inverse(src_pose))


def warp_vertices(transformation_matrix, coordinates, skinning_weights, joint_index):
    """Linear blend skinning of points (svox_t/svox.py:974-976): returns
    (warped points [Q, 3], per-point matrices [Q, 4, 4]); differentiable."""
    return _WarpVerticalFunction.apply(transformation_matrix, coordinates, skinning_weights, joint_index)


def blend_transformation_matrix(transformation_matrix, skinning_weights, joint_index):
    """Per-point blended joint matrices [Q, 4, 4] (svox_t/svox.py:978-981) -- what
    VolumeRenderer.forward takes as `transformation_matrices`."""
    coordinates = torch.zeros((skinning_weights.size(0), 3), device=skinning_weights.device)
    _, matrices = _C.warp_vertices(transformation_matrix, coordinates, skinning_weights, joint_index)
    return matrices


class N3Tree(nn.Module):
    def __init__(self, N=2, data_dim=4, depth_limit=10, init_reserve=1, init_refine=0,
                 geom_resize_fact=1.5, radius=0.5, center=(0.5, 0.5, 0.5),
                 data_format="RGBA", extra_data=None, map_location="cpu"):
        super().__init__()
        assert N >= 2 and depth_limit >= 0
        self.N = int(N)
        self.data_dim = int(data_dim)
        for i in range(1, init_refine + 1):
            init_reserve += (N ** i) ** 3
        dev = map_location
        self.features = nn.Parameter(torch.zeros(init_reserve, data_dim, device=dev))
        self.register_buffer("data", torch.full((init_reserve, N, N, N, 1), EMPTY_INDEX,
                                                dtype=torch.int32, device=dev))
        self.register_buffer("child", torch.zeros(init_reserve, N, N, N, dtype=torch.int32, device=dev))
        self.register_buffer("parent_depth", torch.zeros(init_reserve, 2, dtype=torch.int32, device=dev))
        self.register_buffer("_n_internal", torch.tensor(1, device=dev))
        self.register_buffer("_n_free", torch.tensor(0, device=dev))
        if isinstance(radius, (int, float)):
            radius = [radius] * 3
        radius = torch.tensor(radius, dtype=torch.float32, device=dev)
        center = torch.tensor(center, dtype=torch.float32, device=dev)
        self.register_buffer("invradius", 0.5 / radius)
        self.register_buffer("offset", 0.5 * (1.0 - center / radius))
        self.depth_limit = depth_limit
        self.geom_resize_fact = geom_resize_fact
        self.data_format = DataFormat(data_format) if data_format is not None else None
        if extra_data is not None:
            assert isinstance(extra_data, torch.Tensor)
            self.register_buffer("extra_data", extra_data.to(device=dev))
        else:
            self.extra_data = None
        self._ver = 0
        self._last_all_leaves = None
        self._lock_tree_structure = False
        self._weight_accum = None
        self.filled = 1
        self.refine(repeats=init_refine)

    # ------------------------------------------------------------------ build
    @classmethod
    def from_arrays(cls, child, data, parent_depth, features, data_format="RGBA",
                    radius=0.5, center=(0.5, 0.5, 0.5), depth_limit=10, extra_data=None,
                    device="cpu"):
        """Adopt pre-built topology arrays (e.g. svox_t_amd.synth.shell_tree)."""
        child = torch.as_tensor(child, dtype=torch.int32)
        n, N = child.shape[0], child.shape[1]
        features = torch.as_tensor(features, dtype=torch.float32)
        tree = cls(N=N, data_dim=features.shape[1], depth_limit=depth_limit, init_reserve=1,
                   radius=radius, center=center, data_format=data_format, extra_data=extra_data)
        tree.child = child.contiguous()
        tree.data = torch.as_tensor(data, dtype=torch.int32).reshape(n, N, N, N, 1).contiguous()
        tree.parent_depth = torch.as_tensor(parent_depth, dtype=torch.int32).contiguous()
        tree.features = nn.Parameter(features.contiguous())
        tree._n_internal.fill_(n)
        tree.filled = n
        tree._invalidate()
        return tree.to(device)

    def construct_tree(self, indices):
        """data[leaf containing point i] = i (svox.py:160-161 -> construct_tree_kernel,
        svox_kernel.cu:110-121); the smallest index where points share a leaf."""
        _C.construct_tree(self._spec(self.features), indices.to(self.data.device))
        self._invalidate()

    def build_from_points(self, points, depth, reserve=0):
        """Replace the topology with the octree of a point cloud: what

            for _ in range(depth - 1): tree[points].refine()
            tree.construct_tree(points)

        leaves behind on a fresh tree (helpers.py:101-109, svox.py:488-560, 160-161),
        computed by one HIP pipeline (csrc/svoxt_build.hip) with a single host read
        instead of one query + a dozen tensor ops + a sync per level.  N = 2 only;
        `points` are world coordinates, float32 [P, 3], row i of `features` belongs
        to point i.  `reserve` extra rows are kept free for later refine() calls.
        :return: n_internal"""
        if self._lock_tree_structure:
            raise RuntimeError("Tree locked")
        if self.N != 2:
            raise RuntimeError("build_from_points: N = 2 only; use refine() / construct_tree() for other N")
        if not self.data.is_cuda:
            raise RuntimeError("build_from_points: only the GPU (HIP) path exists; move the tree to a GPU")
        with torch.no_grad():
            points = points.to(device=self.data.device, dtype=torch.float32).contiguous()
            child, data, parent_depth, n = _C.build_octree(points, self.offset, self.invradius, depth,
                                                           EMPTY_INDEX, reserve)
            self.child, self.data, self.parent_depth = child, data, parent_depth
            self._n_internal.fill_(n)
            self.filled = n
            self._invalidate()
        return n

    # ------------------------------------------------------------------ query
    def forward(self, features, indices, cuda=True, want_node_ids=False, world=True,
                want_data_ids=False, want_leaf_node=False):
        """Nearest-leaf feature lookup at `indices` [Q, 3]; differentiable wrt
        `features` (svox.py:216-285).  The reference's non-CUDA branch is broken
        by the index indirection (svox.py:263-264); here `cuda=False` is refused."""
        assert not indices.requires_grad
        assert indices.dim() == 2
        if not cuda or not self.data.is_cuda:
            raise RuntimeError("N3Tree.forward: only the GPU (HIP) path exists; "
                               "move the tree to a GPU and call with cuda=True")
        result, node_ids, data_ids, leaf_node = _QueryVerticalFunction.apply(
            features, self._spec(features, world=world), indices)
        if not (want_node_ids or want_data_ids or want_leaf_node):
            return result
        ret = [result]
        if want_node_ids:
            ret.append(node_ids)
        if want_data_ids:
            ret.append(data_ids)
        if want_leaf_node:
            ret.append(leaf_node)
        return ret

    def __getitem__(self, key):
        return N3TreeView(self, key)

    # ----------------------------------------------------------------- refine
    def refine(self, repeats=1, sel=None, leaf_node=None, node_id=None):
        """Split every selected leaf slot into a new internal node
        (svox.py:488-560).  New nodes are appended in selector order; the child
        word holds the offset from the parent node; the new node's slots inherit
        the parent slot's feature index.  Returns True iff buffers were regrown.

        `sel` = tuple of four index tensors (node, x, y, z); default: all leaves
        shallower than depth_limit.  Unlike the reference, `repeats > 1`
        recomputes the selector each round (the reference reuses a stale
        `leaf_node`, svox.py:521-522)."""
        if self._lock_tree_structure:
            raise RuntimeError("Tree locked")
        resized = False
        with torch.no_grad():
            for rep in range(repeats):
                if sel is None:
                    leaves = self._all_leaves()
                    keep = self.parent_depth[leaves[:, 0].to(self.parent_depth.device), 1].cpu() < self.depth_limit
                    leaf_node = leaves[keep].to(self.data.device)
                    sel = tuple(leaf_node.T)
                elif leaf_node is None:
                    leaf_node = torch.stack(sel, dim=-1).to(self.data.device)
                sel = tuple(s.to(self.data.device).long() for s in sel)
                leaf_node = leaf_node.to(self.data.device)
                n_new = sel[0].shape[0]
                if n_new == 0:
                    return resized
                filled = self.filled
                need = filled + n_new - self.capacity
                if need > 0:
                    self._resize_add_cap(need)
                    resized = True
                if self.data.is_cuda:
                    # one kernel instead of a dozen tensor ops (same tables, bit for bit)
                    nid = None if node_id is None else \
                        torch.as_tensor(node_id, dtype=torch.int32, device=self.data.device).contiguous()
                    _C.refine_leaves(self.child, self.data, self.parent_depth, filled,
                                     leaf_node.long().contiguous(), nid)
                else:
                    new_ids = torch.arange(filled, filled + n_new, device=self.data.device, dtype=torch.int32)
                    self.child[sel] = new_ids - leaf_node[:, 0].to(torch.int32)
                    self.data[filled:filled + n_new] = self.data[sel][:, None, None, None]
                    self.parent_depth[filled:filled + n_new, 0] = \
                        self._pack_index(leaf_node).to(torch.int32) if node_id is None else node_id
                    self.parent_depth[filled:filled + n_new, 1] = self.parent_depth[leaf_node[:, 0].long(), 1] + 1
                self._n_internal += n_new
                self.filled += n_new
                self._invalidate()
                sel = leaf_node = node_id = None
        return resized

    def _resize_add_cap(self, cap_needed):
        """Grow the topology buffers geometrically (svox.py:841-863); `features`
        is owned by the caller in this fork and is not resized."""
        cap_needed = max(cap_needed, int(self.capacity * (self.geom_resize_fact - 1.0)))
        dev = self.data.device
        N = self.N
        self.data = torch.cat((self.data, torch.full((cap_needed, N, N, N, 1), EMPTY_INDEX,
                                                     dtype=torch.int32, device=dev)))
        self.child = torch.cat((self.child, torch.zeros((cap_needed, N, N, N), dtype=torch.int32, device=dev)))
        self.parent_depth = torch.cat((self.parent_depth,
                                       torch.zeros((cap_needed, 2), dtype=torch.int32, device=dev)))

    # ------------------------------------------------------------- properties
    @property
    def n_internal(self):
        return self.filled

    @property
    def capacity(self):
        return self.parent_depth.shape[0]

    @property
    def n_leaves(self):
        return self._all_leaves().shape[0]

    @property
    def max_depth(self):
        return int(self.parent_depth[:self.filled, 1].max().item())

    def _all_leaves(self):
        """[n_leaves, 4] int64 (node, x, y, z) in lexicographic order, on the CPU
        (svox.py:876-880)."""
        if self._last_all_leaves is None:
            self._last_all_leaves = (self.child[:self.filled] == 0).nonzero(as_tuple=False).cpu()
        return self._last_all_leaves

    def _pack_index(self, txyz):
        N = self.N
        return txyz[:, 0] * (N ** 3) + txyz[:, 1] * (N ** 2) + txyz[:, 2] * N + txyz[:, 3]

    def _unpack_index(self, flat):
        N = self.N
        w = flat % N
        v = (flat // N) % N
        u = (flat // (N * N)) % N
        return torch.stack((flat // (N ** 3), u, v, w), dim=-1)

    def _calc_corners(self, nodes):
        """Lower corner in [0,1]^3 of each leaf slot in `nodes` [Q, 4]
        (svox.py:808-826), walking the parent chain with torch ops."""
        nodes = nodes.to(self.parent_depth.device).long()
        corner = torch.zeros(nodes.shape[0], 3, device=nodes.device)
        curr = nodes.clone()
        live = torch.ones(nodes.shape[0], dtype=torch.bool, device=nodes.device)
        while True:
            corner[live] = (corner[live] + curr[:, 1:].float()) / self.N
            up = curr[:, 0] != 0
            if not up.any():
                break
            idx = live.nonzero(as_tuple=False).squeeze(1)[up]
            live = torch.zeros_like(live)
            live[idx] = True
            curr = self._unpack_index(self.parent_depth[curr[up, 0], 0].long())
        return corner

    def world2tree(self, indices):
        return torch.addcmul(self.offset, indices, self.invradius)

    def tree2world(self, indices):
        return (indices - self.offset) / self.invradius

    def _invalidate(self):
        self._ver += 1
        self._last_all_leaves = None

    def accumulate_weights(self):
        """`with tree.accumulate_weights() as accum:` -- per-leaf-slot sum of the
        compositing weights of every render inside the block (svox.py:664-676,
        :948-969).  Accumulated with float atomics (the reference races,
        rt_kernel.cu:310)."""
        return WeightAccumulator(self)

    # ------------------------------------------------------------------- spec
    def _spec(self, features, joint_features=None, skinning_weights=None, joint_index=None,
              transformation_matrices=None, world=True):
        """Pack the tree for the operator boundary (svox.py:899-925)."""
        dev = self.data.device
        spec = _C.TreeSpec()
        spec.features = features
        spec.data = self.data
        spec.child = self.child
        spec.parent_depth = self.parent_depth
        spec.extra_data = self.extra_data if self.extra_data is not None else torch.empty((0, 0), device=dev)
        spec.offset = self.offset if world else torch.zeros(3, device=dev)
        spec.scaling = self.invradius if world else torch.ones(3, device=dev)
        spec.n_internal = self.filled
        spec._weight_accum = self._weight_accum if self._weight_accum is not None \
            else torch.empty(0, device=dev)
        spec.joint_features = joint_features if joint_features is not None else torch.empty((0, 0), device=dev)
        spec.skinning_weights = skinning_weights if skinning_weights is not None else torch.empty((0, 0), device=dev)
        spec.joint_index = joint_index if joint_index is not None \
            else torch.empty((0, 0), device=dev, dtype=torch.int32)
        spec.transformation_matrices = transformation_matrices if transformation_matrices is not None \
            else torch.empty((0, 0, 0), device=dev)
        # (not in the reference) `tree.static_features = True`: the caller's promise that the feature table
        # is not written behind torch's version counter, so what is derived from its content may be cached
        spec.static_features = bool(getattr(self, "static_features", False))
        return spec

    def __repr__(self):
        return (f"svox_t_amd.N3Tree(N={self.N}, data_dim={self.data_dim}, depth_limit={self.depth_limit}, "
                f"capacity:{self.filled}/{self.capacity}, data_format:{self.data_format or 'RGBA'})")


class WeightAccumulator:
    def __init__(self, tree):
        self.tree = tree

    def __enter__(self):
        self.tree._lock_tree_structure = True
        self.tree._weight_accum = torch.zeros(self.tree.child.shape, dtype=torch.float32,
                                              device=self.tree.data.device)
        self.weight_accum = self.tree._weight_accum
        return self

    def __exit__(self, *_exc):
        self.tree._weight_accum = None
        self.tree._lock_tree_structure = False

    @property
    def value(self):
        return self.weight_accum

    def __call__(self):
        """Weights of the leaves, in `_all_leaves()` order."""
        leaves = self.tree._all_leaves().to(self.weight_accum.device)
        return self.weight_accum[tuple(leaves.T)]
