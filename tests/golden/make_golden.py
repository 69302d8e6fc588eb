#!/usr/bin/env python3
"""Generate tests/golden/*.npz / *.json by importing the REFERENCE Python
package on the CPU (this container only; /root/reference never travels).

    python tests/golden/make_golden.py      # needs /root/reference

What can be pinned this way: the reference's render and query
arithmetic is not reachable without its CUDA extension, but its tree-topology
code, SH polynomials and small helpers run on CPU.  The fixtures hold only data
(inputs and the reference's outputs):

  topology_*.npz   child / parent_depth / data / n_internal produced by the
                   reference's N3Tree.refine -- full refinement and selective
                   (shell) refinement driven through refine(sel=..., leaf_node=...)
                   -- plus leaf corners / depths from its CPU _calc_corners
  topology_points_*.npz   the same tables for the reference's per-frame build loop
                   `tree[points].refine()` x (depth-1) (helpers.py:101-109): the
                   reference's refine() driven with the sorted unique-leaf list of a
                   seeded point cloud (its own point query needs the CUDA extension,
                   so the leaves are located by oracle/builder.py's restatement)
  sh_bases.npz     sh.eval_sh_bases(deg, dirs) in float64 for deg 0..4, and
                   sh.eval_sh(deg, coeffs, dirs) (pins the channel-major layout)
  ndc.npz          renderer.convert_to_ndc (renderer.py:140-160) applied in float64 to the
                   pinhole rays of a forward-facing camera (pins maybe_world2ndc)
  helpers.json     DataFormat parse table, offset / invradius / world2tree for a
                   non-unit radius and centre, _pack_index / _unpack_index
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
warnings.simplefilter("ignore")

import svox_t as ref                       # noqa: E402  (the reference, CPU only: _C is None)
from svox_t import sh as ref_sh            # noqa: E402
from svox_t.helpers import DataFormat as RefDataFormat   # noqa: E402

from svox_t_amd import synth               # noqa: E402  (only for the shell predicate)
from oracle import builder as ob           # noqa: E402  (point -> leaf location for topology_points_*)
from oracle import oracle as O             # noqa: E402  (pinhole rays fed to the reference's convert_to_ndc)


def ref_tree_arrays(t):
    n = t.n_internal
    return dict(child=t.child[:n].numpy().copy(), parent_depth=t.parent_depth[:n].numpy().copy(),
                data=t.data[:n].numpy().copy(), n_internal=np.int64(n))


def full_tree(N, levels):
    t = ref.N3Tree(N=N, data_dim=4, init_reserve=200000, data_format="RGBA")
    for _ in range(levels):
        t.refine(1)                         # refine(repeats>1) crashes upstream (svox.py:521-522)
    return t


def shell_tree_via_reference(depth):
    """Drive the reference's selective refine with the shell predicate."""
    t = ref.N3Tree(N=2, data_dim=4, init_reserve=200000, data_format="RGBA")
    for lvl in range(1, depth):
        leaves = t._all_leaves()                                  # [L, 4] (node, x, y, z), lexicographic
        corners = t._calc_corners(leaves, cuda=False).double().numpy()
        depths = t.parent_depth[leaves[:, 0], 1].numpy()
        side = 0.5 ** (depths + 1.0)
        # only the deepest leaves can still meet the shell, but test them all
        hit = np.zeros(len(leaves), dtype=bool)
        for s in np.unique(side):
            m = side == s
            hit[m] = synth._box_hits_shell(corners[m], float(s))
        hit &= side == 0.5 ** lvl
        sel_nodes = leaves[torch.from_numpy(hit)]
        t.refine(1, sel=(*sel_nodes.T,), leaf_node=sel_nodes)
    leaves = t._all_leaves()
    corners = t._calc_corners(leaves, cuda=False).numpy()
    out = ref_tree_arrays(t)
    out.update(leaves=leaves.numpy(), corners=corners,
               depths=t.parent_depth[leaves[:, 0], 1].numpy())
    return out


def point_cloud(n, seed, radius, center):
    """Seeded cloud: a noisy sphere surface inside the cube, a few points outside it, duplicates."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = 0.62 + 0.05 * rng.normal(size=(n, 1))
    pts = (np.asarray(center) + np.asarray(radius) * r * d).astype(np.float32)
    pts[: n // 50] *= 3.0                       # outside the cube: clamped to its faces
    pts[n // 2: n // 2 + n // 20] = pts[: n // 20]   # exact duplicates
    return pts


def points_tree_via_reference(points, depth, radius, center):
    """The reference's N3TreeView.refine loop (helpers.py:101-109) on the CPU."""
    t = ref.N3Tree(N=2, data_dim=4, init_reserve=400000, radius=radius, center=center, data_format="RGBA")
    offset, scaling = t.offset.numpy(), t.invradius.numpy()
    for _ in range(depth - 1):
        topo = ob.Topology(N=2)
        n = t.n_internal
        topo.child, topo.n = t.child[:n].numpy(), n
        leaf_node = torch.from_numpy(ob.unique_leaves(topo, ob.descend(topo, points, offset, scaling)))
        t.refine(1, sel=(*leaf_node.T,), leaf_node=leaf_node)
    out = ref_tree_arrays(t)
    out.update(points=points, radius=np.asarray(radius, np.float32), center=np.asarray(center, np.float32),
               depth=np.int64(depth), offset=offset, scaling=scaling)
    return out


def main():
    # ---- topology ---------------------------------------------------------
    for name, n, depth, radius, center in (("a", 400, 4, [0.5] * 3, [0.5] * 3),
                                           ("b", 3000, 6, [1.0, 1.2, 0.8], [0.1, -0.2, 0.3])):
        np.savez_compressed(os.path.join(HERE, f"topology_points_{name}.npz"),
                            **points_tree_via_reference(point_cloud(n, 11, radius, center), depth, radius, center))
    np.savez_compressed(os.path.join(HERE, "topology_full_n2_l3.npz"), **ref_tree_arrays(full_tree(2, 3)))
    np.savez_compressed(os.path.join(HERE, "topology_full_n3_l2.npz"), **ref_tree_arrays(full_tree(3, 2)))
    for d in (3, 4, 5):
        np.savez_compressed(os.path.join(HERE, f"topology_shell_d{d}.npz"), **shell_tree_via_reference(d))

    # ---- SH ---------------------------------------------------------------
    g = torch.Generator().manual_seed(7)
    dirs = torch.randn(64, 3, generator=g, dtype=torch.float64)
    dirs /= dirs.norm(dim=-1, keepdim=True)
    dirs[0] = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    dirs[1] = torch.tensor([1.0, 0.0, 0.0], dtype=torch.float64)
    dirs[2] = torch.tensor([0.0, -1.0, 0.0], dtype=torch.float64)
    out = {"dirs": dirs.numpy()}
    for deg in range(5):
        out[f"bases_deg{deg}"] = ref_sh.eval_sh_bases(deg, dirs).numpy()
    coeffs = torch.randn(64, 3, 9, generator=g, dtype=torch.float64)
    out["coeffs_deg2"] = coeffs.numpy()
    out["eval_sh_deg2"] = ref_sh.eval_sh(2, coeffs, dirs).numpy()
    np.savez_compressed(os.path.join(HERE, "sh_bases.npz"), **out)

    # ---- NDC ----------------------------------------------------------------
    from svox_t.renderer import convert_to_ndc
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = [[0.9986295, 0.0, 0.0523360], [0.0027390, 0.9986295, -0.0522642], [-0.0522642, 0.0523360, 0.9972609]]
    pose[:3, 3] = [0.3, -0.2, 2.5]
    W, H, fx, fy, focal = 24, 16, 30.0, 33.0, 31.0
    o, d, _ = O.camera_rays(pose, fx, fy, W, H)
    ro, rd = convert_to_ndc(torch.from_numpy(o).double(), torch.from_numpy(d).double(), focal, W, H)
    np.savez_compressed(os.path.join(HERE, "ndc.npz"), pose=pose, W=W, H=H, fx=fx, fy=fy, focal=focal,
                        origins=o, dirs=d, ndc_origins=ro.numpy(), ndc_dirs=rd.numpy())

    # ---- helpers ------------------------------------------------------------
    fmt = {}
    for txt in ["RGBA", "", "SH1", "SH4", "SH9", "SH16", "SH25", "SG25", "ASG8", "XYZ3", "RGBA4"]:
        f = RefDataFormat(txt)
        fmt[txt] = {"format": int(f.format), "basis_dim": int(f.basis_dim), "repr": repr(f)}
    radius, center = [1.0, 1.2, 0.8], [0.1, -0.2, 0.3]
    t = ref.N3Tree(N=2, data_dim=4, init_reserve=10, radius=radius, center=center, data_format="RGBA")
    pts = torch.randn(16, 3, generator=g)
    txyz = torch.tensor([[0, 0, 0, 0], [5, 1, 0, 1], [123, 1, 1, 1], [7, 0, 1, 0]])
    t3 = ref.N3Tree(N=3, data_dim=4, init_reserve=10)
    helpers = {
        "data_format": fmt,
        "radius": radius, "center": center,
        "offset": t.offset.tolist(), "invradius": t.invradius.tolist(),
        "points": pts.tolist(), "world2tree": t.world2tree(pts).tolist(),
        "txyz": txyz.tolist(),
        "pack_n2": t._pack_index(txyz).tolist(),
        "unpack_n2": t._unpack_index(t._pack_index(txyz).clone()).tolist(),
        "pack_n3": t3._pack_index(txyz).tolist(),
        "empty_sentinel": int(t.data.flatten()[0].item()),
    }
    with open(os.path.join(HERE, "helpers.json"), "w") as f:
        json.dump(helpers, f, indent=1)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
