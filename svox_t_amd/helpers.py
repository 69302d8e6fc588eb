"""Small host-side helpers of the hot path: the `data_format` string parser,
the `LocalIndex` marker and the point-keyed tree view used to drive `refine`.

Reference: svox_t/helpers.py:363-420 (`_get_c_extension`, `LocalIndex`,
`DataFormat`) and the point-query branch of `N3TreeView` (:54-63, :101-109,
:143-176).  The tensor-arithmetic conveniences of `N3TreeView` are outside the
hot path.
"""
from __future__ import annotations

import torch


def _get_c_extension():
    """Return the operator module.  The reference swallows a failed import and
    falls back to slow paths (helpers.py:363-376); here a missing HIP library
    is an error -- there is no fallback to fall back to."""
    import svox_t_amd.csrc as _C
    if not hasattr(_C, "query_vertical"):
        raise ImportError("svox_t_amd.csrc lacks query_vertical: stale build?")
    return _C


class LocalIndex:
    """tree[LocalIndex(points)]: query with points already in [0,1]^3."""

    def __init__(self, val):
        self.val = val


class DataFormat:
    """Parses "RGBA", "SH9", "SG25", "ASG8", ... into (format, basis_dim).

    The table must match the reference (helpers.py:386-420): a string with no
    digits is RGBA with basis_dim = -1; otherwise the alphabetic prefix selects
    the family (anything unknown -> RGBA) and the rest is basis_dim.
    """
    RGBA = 0
    SH = 1
    SG = 2
    ASG = 3
    _BY_NAME = {"SH": SH, "SG": SG, "ASG": ASG}
    _NAMES = {RGBA: "RGBA", SH: "SH", SG: "SG", ASG: "ASG"}

    def __init__(self, txt: str):
        split = next((i for i, ch in enumerate(txt) if not ch.isalpha()), None)
        if split is None:
            self.format = DataFormat.RGBA
            self.basis_dim = -1
        else:
            self.basis_dim = int(txt[split:])
            self.format = DataFormat._BY_NAME.get(txt[:split], DataFormat.RGBA)

    def __repr__(self):
        r = DataFormat._NAMES[self.format]
        return r + str(self.basis_dim) if self.basis_dim >= 0 else r


class N3TreeView:
    """View of the leaves hit by a batch of query points: `tree[points]`.

    Supports what the hot path's callers need: the gathered values, the
    per-point packed leaf ids, the unique leaf list and `refine()`.
    """

    def __init__(self, tree, key):
        self.tree = tree
        local = isinstance(key, LocalIndex)
        if local:
            key = key.val
        if not (torch.is_tensor(key) and key.dim() == 2 and key.shape[1] == 3):
            raise NotImplementedError(
                "N3TreeView: only point keys tree[P] with P of shape [B, 3] are supported")
        pts = key if key.dtype == torch.float32 else key.float()
        vals, packed, leaves = tree.forward(tree.features, pts, want_node_ids=True,
                                            world=not local, want_leaf_node=True)
        self._values = vals
        self._packed_ids = packed
        self.leaf_node_id = packed
        self.unique_leaf_node = leaves
        self.key = tuple(leaves.T)
        self._tree_ver = tree._ver

    def _check_ver(self):
        if self.tree._ver > self._tree_ver:
            raise RuntimeError("N3TreeView has been invalidated because tree data layout has changed")

    @property
    def values(self):
        self._check_ver()
        return self._values

    def refine(self, repeats: int = 1):
        """Refine the leaves this view selects (helpers.py:101-109)."""
        self._check_ver()
        return self.tree.refine(repeats, sel=tuple(self.unique_leaf_node.T),
                                leaf_node=self.unique_leaf_node)

    @property
    def depths(self):
        self._check_ver()
        return self.tree.parent_depth[self.key[0], 1]

    @property
    def lengths_local(self):
        # leaf side N^-(depth + 1).  The reference writes 2.0 ** (...) (helpers.py:164,176), which is
        # this for the N = 2 trees it is used with and wrong for any other branching factor.
        return float(self.tree.N) ** (-self.depths.float() - 1.0)

    @property
    def lengths(self):
        return self.lengths_local[:, None] / self.tree.invradius

    @property
    def corners_local(self):
        self._check_ver()
        return self.tree._calc_corners(self.unique_leaf_node)

    @property
    def corners(self):
        return (self.corners_local - self.tree.offset) / self.tree.invradius
