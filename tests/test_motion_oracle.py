"""Known answers for the oracle's motion variants (rt_kernel.cu:698-778, 886-981)
and a float64 finite-difference check of the joint-feature gradient.  No GPU."""
import math

import numpy as np

from oracle import oracle as O
from svox_t_amd import synth

SENT = synth.EMPTY_SENTINEL


def root_tree(rows, K, extra=None, slot_rows=None, dtype=np.float32):
    child = np.zeros((1, 2, 2, 2), np.int32)
    data = np.zeros((1, 2, 2, 2, 1), np.int32) if slot_rows is None \
        else np.asarray(slot_rows, np.int32).reshape(1, 2, 2, 2, 1)
    return O.Tree(np.asarray(rows, dtype).reshape(-1, K), data, child, extra=extra, dtype=dtype)


def axis_ray(y=0.25, z=0.25):
    o = np.array([[-1.0, y, z]], np.float32)
    d = np.array([[1.0, 0.0, 0.0]], np.float32)
    return o, d, d.copy()


def test_motion_render_first_hit_closed_form():
    """Ray along +x enters the root at t = 1, in slot (0,0,0).  The reference maps the
    LEAF-LOCAL point (0, 0.5, 0.5) back with (p - offset) / scaling -- not the tree-space
    point (0, 0.25, 0.25) -- and measures joint distances from there."""
    joints = np.float32([[0.0, 0.5, 0.5, 9.0], [1.0, 0.5, 0.5, 9.0], [0.0, 0.5, 2.5, 9.0]])
    t = root_tree([[0.1, 0.2, 0.3, 2.0]], 4, extra=joints)
    out, depth, hit, idx = O.motion_render(t, *axis_ray(), O.make_options())
    np.testing.assert_array_equal(hit[0], [0.0, 0.5, 0.5])
    np.testing.assert_allclose(out[0], [0.0, 1.0, 2.0], atol=1e-7)
    assert depth[0, 0] == 1.0 and idx[0, 0] == 0
    # offset / scaling enter the hit point, and the depth through delta_scale
    t2 = O.Tree(t.features, t.data, t.child, offset=(0.25, 0.0, 0.0), scaling=(0.5, 1.0, 1.0), extra=joints)
    out2, depth2, hit2, _ = O.motion_render(t2, np.float32([[-2.0, 0.25, 0.25]]), *axis_ray()[1:], O.make_options())
    # tree-space origin x = 0.25 + 0.5 * (-2) = -0.75, direction (0.5,0,0) normalised, delta_scale = 2
    np.testing.assert_allclose(depth2[0, 0], 0.75 * 2.0, rtol=1e-6)
    np.testing.assert_allclose(hit2[0], [(0.0 - 0.25) / 0.5, 0.5, 0.5], atol=1e-6)
    # second slot along the ray occupied only: first slot empty -> hit in slot (1,0,0), row 1
    t3 = root_tree([[0, 0, 0, 5.0], [0, 0, 0, 7.0]], 4, extra=joints, slot_rows=[SENT, SENT, SENT, SENT, 1, SENT, SENT, SENT])
    _, depth3, _, idx3 = O.motion_render(t3, *axis_ray(), O.make_options())
    assert idx3[0, 0] == 1 and abs(depth3[0, 0] - 1.501) < 1e-6
    # miss, and a ray that only meets sigma <= sigma_thresh: zeros everywhere
    for tree, rays, opt in ((t, (np.float32([[-1, 2, 0.5]]), *axis_ray()[1:]), O.make_options()),
                            (t, axis_ray(), O.make_options(sigma_thresh=3.0))):
        for a in O.motion_render(tree, *rays, opt):
            assert not a.any()


def test_motion_feature_render_closed_form():
    sigma = 3.0
    jf = np.float32([[0.5, -1.0, 2.0], [1.5, 0.25, -0.5], [9.0, 9.0, 9.0]])
    sw = np.float32([[0.75, 0.25, 0.0, -1.0]])                 # weights <= 0 are skipped (:948)
    ji = np.int32([[0, 1, 2, 2]])
    t = root_tree([[0, 0, 0, sigma]], 4)
    opt = O.make_options(background_brightness=0.5)
    out = O.motion_feature_render(t, O.Motion(jf, sw, ji), *axis_ray(), opt)[0]
    alpha = 1.0 - math.exp(-sigma * 1.001)                      # two leaves: 0.501 + 0.5
    p = 0.75 * jf[0] + 0.25 * jf[1]
    np.testing.assert_allclose(out, alpha / (1 + np.exp(-p)) + (1 - alpha) * 0.5, rtol=2e-6)
    # a ray that misses the cube gives zeros, not the background (:913-919)
    miss = O.motion_feature_render(t, O.Motion(jf, sw, ji), np.float32([[-1, 2, 0.5]]), *axis_ray()[1:], opt)
    assert not miss.any()
    # early stop rescales by 1 / (1 - T) and adds no background (:958-963)
    opt_s = O.make_options(background_brightness=0.5, stop_thresh=0.5, sigma_thresh=0.5)
    out_s = O.motion_feature_render(t, O.Motion(jf, sw, ji), *axis_ray(), opt_s)[0]
    np.testing.assert_allclose(out_s, 1 / (1 + np.exp(-p)), rtol=2e-6)   # stops in leaf 1: weight / (1 - T) = 1


def test_motion_feature_gradient_matches_finite_differences():
    st = synth.shell_tree(3)
    M = st.n_features
    rng = np.random.default_rng(0)
    feats = synth.shell_features(M, 4).numpy().astype(np.float64)
    J, F, B = 5, 6, 3
    jf = rng.normal(size=(J, F))
    sw = rng.random((M, B))
    sw[rng.random((M, B)) < 0.3] = 0
    ji = rng.integers(0, J, size=(M, B))
    o, d, v = (a.numpy().astype(np.float64) for a in synth.pinhole_rays(12, 12))
    t = O.Tree(feats, st.data, st.child, dtype=np.float64)
    opt = O.make_options()
    g = rng.normal(size=(144, F))
    grad = O.motion_feature_render_backward(t, O.Motion(jf, sw, ji, np.float64), o, d, v, opt, g)
    assert np.abs(grad).max() > 0.1
    eps = 1e-6
    for a in range(J):
        for b in range(F):
            jp, jm = jf.copy(), jf.copy()
            jp[a, b] += eps
            jm[a, b] -= eps
            fp = O.motion_feature_render(t, O.Motion(jp, sw, ji, np.float64), o, d, v, opt)
            fm = O.motion_feature_render(t, O.Motion(jm, sw, ji, np.float64), o, d, v, opt)
            assert abs(((fp - fm) * g).sum() / (2 * eps) - grad[a, b]) < 1e-7
    # float32 evaluation of the same gradient agrees to float accuracy
    g32, a32 = O.motion_feature_render_backward(t.astype(np.float32), O.Motion(jf, sw, ji), o, d, v, opt, g, want_abs=True)
    assert (np.abs(g32 - grad) <= 2e-5 * a32 + 1e-7).all()
