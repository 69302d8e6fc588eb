"""The oracle and the host-side code against fixtures captured from the
reference's own Python (tests/golden/make_golden.py; data only).  No GPU."""
import json
import os

import numpy as np
import pytest
import torch

import svox_t_amd as svox
from oracle import oracle as O
from svox_t_amd import synth

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name))


@pytest.fixture(scope="module")
def helpers():
    with open(os.path.join(G, "helpers.json")) as f:
        return json.load(f)


# ---------------------------------------------------------------- SH basis
@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_sh_basis_matches_reference_sh_py(deg):
    """maybe_precalc_basis (rt_kernel.cu:139-178) == sh.eval_sh_bases (sh.py:114-162)."""
    g = load("sh_bases.npz")
    bd = (deg + 1) ** 2
    got = O.basis(O.FORMAT_SH, bd, g["dirs"].astype(np.float32))
    np.testing.assert_allclose(got, g[f"bases_deg{deg}"], rtol=0, atol=1e-6)


def test_sh_row_layout_matches_eval_sh():
    """sum_i basis_i * row[c*bd + i] with the channel-major row the kernels read
    equals sh.eval_sh(deg, coeffs[..., C, bd], dirs)."""
    g = load("sh_bases.npz")
    basis = O.basis(O.FORMAT_SH, 9, g["dirs"].astype(np.float32)).astype(np.float64)
    rows = g["coeffs_deg2"].reshape(64, 27)                      # [R x 9, G x 9, B x 9]
    got = np.stack([(basis * rows[:, c * 9:(c + 1) * 9]).sum(-1) for c in range(3)], -1)
    np.testing.assert_allclose(got, g["eval_sh_deg2"], rtol=0, atol=1e-5)


# ---------------------------------------------------------------- helpers
def test_data_format_table(helpers):
    for txt, want in helpers["data_format"].items():
        f = svox.DataFormat(txt)
        assert (f.format, f.basis_dim, repr(f)) == (want["format"], want["basis_dim"], want["repr"]), txt


def test_world2tree_offset_invradius(helpers):
    t = svox.N3Tree(N=2, data_dim=4, init_reserve=4, radius=helpers["radius"], center=helpers["center"])
    np.testing.assert_array_equal(t.offset.numpy(), np.float32(helpers["offset"]))
    np.testing.assert_array_equal(t.invradius.numpy(), np.float32(helpers["invradius"]))
    pts = torch.tensor(helpers["points"])
    np.testing.assert_array_equal(t.world2tree(pts).numpy(), np.float32(helpers["world2tree"]))
    np.testing.assert_allclose(t.tree2world(t.world2tree(pts)).numpy(), pts.numpy(), atol=1e-6)


def test_pack_unpack_and_sentinel(helpers):
    txyz = torch.tensor(helpers["txyz"])
    t2 = svox.N3Tree(N=2, data_dim=4, init_reserve=4)
    t3 = svox.N3Tree(N=3, data_dim=4, init_reserve=4)
    assert t2._pack_index(txyz).tolist() == helpers["pack_n2"]
    assert t3._pack_index(txyz).tolist() == helpers["pack_n3"]
    assert t2._unpack_index(torch.tensor(helpers["pack_n2"])).tolist() == helpers["unpack_n2"]
    assert int(t2.data.flatten()[0]) == helpers["empty_sentinel"] == synth.EMPTY_SENTINEL


# ---------------------------------------------------------------- topology
def _arrays(t):
    n = t.n_internal
    return t.child[:n].numpy(), t.parent_depth[:n].numpy(), t.data[:n].numpy()


@pytest.mark.parametrize("N,levels,name", [(2, 3, "topology_full_n2_l3.npz"), (3, 2, "topology_full_n3_l2.npz")])
def test_full_refine_matches_reference(N, levels, name):
    g = load(name)
    t = svox.N3Tree(N=N, data_dim=4, init_reserve=10)          # small reserve: exercises regrowth
    for _ in range(levels):
        t.refine(1)
    child, pd, data = _arrays(t)
    assert t.n_internal == int(g["n_internal"])
    np.testing.assert_array_equal(child, g["child"])
    np.testing.assert_array_equal(pd, g["parent_depth"])
    np.testing.assert_array_equal(data, g["data"])
    # refine(repeats=k) == k x refine(1) (the reference crashes here, svox.py:521-522)
    t2 = svox.N3Tree(N=N, data_dim=4, init_reserve=10)
    t2.refine(levels)
    np.testing.assert_array_equal(_arrays(t2)[0], g["child"])
    t3 = svox.N3Tree(N=N, data_dim=4, init_reserve=1, init_refine=levels)
    np.testing.assert_array_equal(_arrays(t3)[0], g["child"])


@pytest.mark.parametrize("depth", [3, 4, 5])
def test_shell_tree_matches_reference_selective_refine(depth):
    """synth.shell_tree (direct numpy builder) and N3Tree.refine(sel=...) both
    reproduce the arrays the reference's refine produced for the same selection."""
    g = load(f"topology_shell_d{depth}.npz")
    s = synth.shell_tree(depth)
    assert s.n_internal == int(g["n_internal"])
    np.testing.assert_array_equal(s.child, g["child"])
    np.testing.assert_array_equal(s.parent_depth, g["parent_depth"])
    # same selection driven through this package's refine
    t = svox.N3Tree(N=2, data_dim=4, init_reserve=8)
    for lvl in range(1, depth):
        leaves = t._all_leaves()
        corners = t._calc_corners(leaves).double().numpy()
        side = 0.5 ** (t.parent_depth[leaves[:, 0], 1].numpy() + 1.0)
        hit = np.array([synth._box_hits_shell(corners[i:i + 1], float(side[i]))[0] for i in range(len(leaves))])
        hit &= side == 0.5 ** lvl
        sel = leaves[torch.from_numpy(hit)]
        t.refine(1, sel=tuple(sel.T), leaf_node=sel)
    child, pd, _ = _arrays(t)
    np.testing.assert_array_equal(child, g["child"])
    np.testing.assert_array_equal(pd, g["parent_depth"])
    # leaf enumeration order and corners (svox.py:876-880, :808-826)
    np.testing.assert_array_equal(t._all_leaves().numpy(), g["leaves"])
    np.testing.assert_allclose(t._calc_corners(t._all_leaves()).numpy(), g["corners"], atol=1e-7)
    np.testing.assert_array_equal(t.parent_depth[t._all_leaves()[:, 0], 1].numpy(), g["depths"])


def test_shell_tree_sizes_of_the_benchmark_configs():
    """D=5 -> n_internal 921, M 3344 (D=8/9 are checked by the GPU suite)."""
    s = synth.shell_tree(5)
    assert (s.n_internal, s.n_features) == (921, 3344)
    # occupied slots carry consecutive feature indices in (node, x, y, z) order
    idx = s.data.reshape(-1)
    occ = idx[idx != synth.EMPTY_SENTINEL]
    np.testing.assert_array_equal(occ, np.arange(s.n_features))
    # only finest-level slots are occupied and none of them has a child
    assert not s.child.reshape(-1)[idx != synth.EMPTY_SENTINEL].any()


# ---------------------------------------------------------------- camera rays
def test_ndc_warp_matches_reference_convert_to_ndc():
    """The oracle's maybe_world2ndc (rt_kernel.cu:1170-1190) == renderer.convert_to_ndc
    (renderer.py:140-160, float64) followed by the normalisation the kernel adds."""
    g = load("ndc.npz")
    W, H = int(g["W"]), int(g["H"])
    o, d, v = O.camera_rays(g["pose"], float(g["fx"]), float(g["fy"]), W, H, ndc=(W, H, float(g["focal"])))
    np.testing.assert_array_equal(v, g["dirs"])                        # view dirs: before the warp
    np.testing.assert_allclose(o, g["ndc_origins"], rtol=0, atol=2e-6)
    want_d = g["ndc_dirs"] / np.linalg.norm(g["ndc_dirs"], axis=-1, keepdims=True)
    np.testing.assert_allclose(d, want_d, rtol=0, atol=2e-6)
    # without NDC: unit directions, origin = camera position, centre pixel looks down -z of the camera
    o0, d0, v0 = O.camera_rays(g["pose"], float(g["fx"]), float(g["fy"]), W, H)
    np.testing.assert_array_equal(o0, np.broadcast_to(g["pose"][:3, 3], o0.shape))
    np.testing.assert_allclose(np.linalg.norm(d0, axis=-1), 1.0, atol=1e-6)
    np.testing.assert_array_equal(d0, v0)
    centre = d0.reshape(H, W, 3)[H // 2, W // 2]
    np.testing.assert_allclose(centre, -g["pose"][:3, 2], atol=1e-6)
