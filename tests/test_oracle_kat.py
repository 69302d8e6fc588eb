"""Known-answer and self-consistency tests of the CPU oracle:
closed-form cases, fp64 finite differences, an independent PyTorch renderer
and its autograd gradient.  No GPU."""
import math

import numpy as np
import pytest
import torch

from oracle import oracle as O
from oracle import torch_renderer as TR
from svox_t_amd import synth
from tests.util import Case

SENT = synth.EMPTY_SENTINEL


def root_only_tree(K, rows, slot_rows=None):
    """N=2 root with 8 leaf slots; slot i -> feature row slot_rows[i] (default all row 0)."""
    child = np.zeros((1, 2, 2, 2), np.int32)
    data = np.zeros((1, 2, 2, 2, 1), np.int32) if slot_rows is None \
        else np.asarray(slot_rows, np.int32).reshape(1, 2, 2, 2, 1)
    return O.Tree(np.asarray(rows, np.float32).reshape(-1, K), data, child)


def axis_ray(y=0.25, z=0.25):
    o = np.array([[-1.0, y, z]], np.float32)
    d = np.array([[1.0, 0.0, 0.0]], np.float32)
    return o, d, d.copy()


def test_homogeneous_cube_axis_ray_closed_form():
    """Ray along +x through two leaves of an occupied root.  March by hand:
    t0 = 1 (entry); leaf 1: remaining chord 0.5 -> delta 0.501; leaf 2 entered
    0.001 inside: remaining 0.499 -> delta 0.5; t = 2.001 >= tmax = 2.
    alpha = 1 - exp(-sigma * 1.001); colour = alpha*sigmoid(c) + (1-alpha)*bg."""
    sigma, c = 3.0, np.array([0.3, -1.2, 2.0])
    t = root_only_tree(4, [[*c, sigma]])
    out = O.volume_render(t, *axis_ray(), O.make_options())[0]
    alpha = 1.0 - math.exp(-sigma * 1.001)
    want = np.append(alpha / (1 + np.exp(-c)) + (1 - alpha) * 1.0, alpha)
    np.testing.assert_allclose(out, want, rtol=2e-6, atol=1e-7)
    assert O.render_depth(t, *axis_ray(), O.make_options())[0, 0] == pytest.approx(1.0, rel=1e-6)
    assert O.opacity_render(t, *axis_ray(), O.make_options())[0, 0] == pytest.approx(alpha, rel=2e-6)


def test_miss_empty_and_negative_sigma():
    opt = O.make_options(background_brightness=0.25)
    t = root_only_tree(4, [[0.1, 0.2, 0.3, 5.0]])
    o = np.array([[-1.0, 2.0, 0.5]], np.float32)       # passes above the cube
    d = np.array([[1.0, 0.0, 0.0]], np.float32)
    np.testing.assert_array_equal(O.volume_render(t, o, d, d, opt)[0], [0.25, 0.25, 0.25, 0.0])
    # all slots empty (sentinel index >= M)
    te = root_only_tree(4, [[0.1, 0.2, 0.3, 5.0]], slot_rows=[SENT] * 8)
    np.testing.assert_array_equal(O.volume_render(te, *axis_ray(), opt)[0], [0.25, 0.25, 0.25, 0.0])
    # negative sigma behaves as empty (ReLU)
    tn = root_only_tree(4, [[0.1, 0.2, 0.3, -2.0]])
    np.testing.assert_array_equal(O.volume_render(tn, *axis_ray(), opt)[0], [0.25, 0.25, 0.25, 0.0])
    assert O.render_depth(tn, *axis_ray(), opt)[0, 0] == 0.0
    g = O.volume_render_backward(tn, *axis_ray(), opt, np.ones((1, 4), np.float32))
    assert not g.any()


def test_early_stop_rescales():
    """fast mode: T <= 1e-2 stops and divides the colour by (1 - T)."""
    sigma, c = 50.0, 0.7
    t = root_only_tree(4, [[c, c, c, sigma]])
    opt = O.make_options(sigma_thresh=1e-2, stop_thresh=1e-2)
    out = O.volume_render(t, *axis_ray(), opt)[0]
    T = math.exp(-sigma * 0.501)                         # stops after the first leaf
    s = 1 / (1 + math.exp(-c))
    np.testing.assert_allclose(out, [s, s, s, 1 - T], rtol=2e-6)


def test_sh_uses_channel_major_rows():
    """Row layout [R x bd, G x bd, B x bd, sigma] (rt_kernel.cu:294-299): with
    only the DC coefficient set, colour_c = sigmoid(C0 * coeff_c)."""
    row = np.zeros(28, np.float32)
    row[0], row[9], row[18], row[27] = 1.0, -2.0, 0.5, 4.0
    t = root_only_tree(28, [row])
    out = O.volume_render(t, *axis_ray(), O.make_options(format=O.FORMAT_SH, basis_dim=9))[0]
    alpha = 1.0 - math.exp(-4.0 * 1.001)
    C0 = 0.28209479177387814
    want = [alpha / (1 + math.exp(-C0 * v)) + (1 - alpha) for v in (1.0, -2.0, 0.5)]
    np.testing.assert_allclose(out[:3], want, rtol=2e-6)


@pytest.fixture(scope="module")
def small_case():
    return Case(depth=3, K=13, data_format="SH4", width=24, height=24,
                radius=[1.0, 1.2, 0.8], center=[0.1, -0.2, 0.3])


def test_gradient_matches_fp64_finite_differences(small_case):
    """d loss / d features of the oracle's hand-derived backward vs central
    differences of its own forward, everything in float64."""
    c = small_case
    t64 = c.oracle_tree().astype(np.float64)
    opt = c.oracle_opts()
    rays = tuple(a.astype(np.float64) for a in c.rays_np())
    g = synth.grad_output(c.Q, 4).numpy().astype(np.float64)
    grad = O.volume_render_backward(t64, *rays, opt, g)
    touched = np.argwhere(np.abs(grad) > 1e-6)
    rng = np.random.default_rng(0)
    picks = touched[rng.choice(len(touched), size=40, replace=False)]
    eps = 1e-6
    for (i, j) in picks:
        f = t64.features.copy()
        f[i, j] += eps
        lp = (O.volume_render(O.Tree(f, t64.data, t64.child, t64.offset, t64.scaling, dtype=np.float64),
                              *rays, opt) * g).sum()
        f[i, j] -= 2 * eps
        lm = (O.volume_render(O.Tree(f, t64.data, t64.child, t64.offset, t64.scaling, dtype=np.float64),
                              *rays, opt) * g).sum()
        fd = (lp - lm) / (2 * eps)
        assert fd == pytest.approx(grad[i, j], rel=1e-5, abs=1e-8), (i, j)


def test_fp32_backward_close_to_fp64(small_case):
    c = small_case
    opt = c.oracle_opts()
    g = synth.grad_output(c.Q, 4).numpy()
    g32, absum = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), opt, g, want_abs=True)
    t64 = c.oracle_tree().astype(np.float64)
    g64 = O.volume_render_backward(t64, *(a.astype(np.float64) for a in c.rays_np()), opt,
                                   g.astype(np.float64))
    # fp32 transmittance behind several opaque samples carries ~1e-5 relative
    # error of its own (exp amplifies the argument's rounding), so entries that
    # are negligible against the gradient's scale get an absolute floor.
    # (weight = T * (1 - att) also loses relative accuracy when att ~ 1: an
    # fp32 evaluation of the reference formula is itself only ~1e-4 accurate
    # on single-sample entries -- hence 2e-4 here; the exact check of the
    # formula is the fp64 finite-difference test above.)
    assert np.all(np.abs(g32 - g64) <= 2e-4 * absum + 1e-9 * absum.max())


@pytest.mark.parametrize("fmt,K,fast", [("RGBA", 4, False), ("SH9", 28, False), ("RGBA", 4, True)])
def test_torch_renderer_agrees_forward(fmt, K, fast):
    """Independent vectorised PyTorch implementation vs the C++ oracle (BASELINE
    config 1 is the RGBA row: depth 5, data_dim 4, 64x64, forward)."""
    c = Case(depth=5, K=K, data_format=fmt, width=64, height=64)
    opt = c.oracle_opts(fast=fast)
    want = O.volume_render(c.oracle_tree(), *c.rays_np(), opt)
    got = TR.volume_render(c.oracle_tree(), *c.rays_np(), opt).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6)


def test_autograd_gradient_agrees_with_hand_derived_backward(small_case):
    """torch.autograd through the compositing formula vs the reference's
    two-pass backward formula (rt_kernel.cu:331-496) as restated in the oracle."""
    c = small_case
    opt = c.oracle_opts()
    ot = c.oracle_tree()
    feats = torch.from_numpy(ot.features).double().requires_grad_(True)
    out = TR.volume_render(ot, *c.rays_np(), opt, features=feats)
    g = synth.grad_output(c.Q, 4)
    out.backward(g.double())
    want, absum = O.volume_render_backward(ot, *c.rays_np(), opt, g.numpy(), want_abs=True)
    assert np.all(np.abs(feats.grad.numpy() - want) <= 2e-4 * absum + 1e-9 * absum.max())


def test_opacity_backward_is_backward_with_zero_channels():
    """opacity_render_backward launches the generic backward with C = 0
    (rt_kernel.cu:1607): only d/d sigma = delta * delta_scale * g * T_end."""
    c = Case(depth=4, K=4, data_format="RGBA", width=16, height=16)
    g = synth.grad_output(c.Q, 1, seed=5).numpy()
    grad = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g)
    assert not grad[:, :3].any() and grad[:, 3].any()
    # against autograd of the torch renderer's alpha column
    ot = c.oracle_tree()
    feats = torch.from_numpy(ot.features).double().requires_grad_(True)
    out = TR.volume_render(ot, *c.rays_np(), c.oracle_opts(), features=feats)
    out[:, 3:].backward(torch.from_numpy(g).double())
    np.testing.assert_allclose(grad[:, 3], feats.grad[:, 3].numpy(), rtol=1e-4, atol=1e-9)


def test_ray_permutation_and_alpha_range():
    c = Case(depth=5, K=4, data_format="RGBA", width=32, height=32)
    opt = c.oracle_opts()
    o, d, v = c.rays_np()
    out = O.volume_render(c.oracle_tree(), o, d, v, opt)
    perm = np.random.default_rng(1).permutation(c.Q)
    out_p = O.volume_render(c.oracle_tree(), o[perm], d[perm], v[perm], opt)
    np.testing.assert_array_equal(out_p, out[perm])
    assert out[:, 3].min() >= 0.0 and out[:, 3].max() <= 1.0


def test_choice_of_expf_moves_results_only_at_ulp_level():
    """The oracle's fixed-sequence expf vs glibc's expf: <= 1 ulp apart, and
    the rendered outputs move by ~1e-7."""
    x = np.linspace(-87, 88, 200001).astype(np.float32)
    a = O.expf(x)
    ref = np.exp(x.astype(np.float64))
    assert np.max(np.abs(a - ref) / np.spacing(ref.astype(np.float32))) < 1.0
    c = Case(depth=5, K=28, data_format="SH9", width=48, height=48)
    opt = c.oracle_opts()
    out_a = O.volume_render(c.oracle_tree(), *c.rays_np(), opt)
    try:
        O.use_libm_exp(True)
        out_b = O.volume_render(c.oracle_tree(), *c.rays_np(), opt)
    finally:
        O.use_libm_exp(False)
    np.testing.assert_allclose(out_a, out_b, rtol=0, atol=2e-6)


def test_two_one_ulp_exponentials_are_more_than_1e5_of_the_tight_scale_apart():
    """What "gradients within 1e-5" can mean between two implementations that do NOT share their expf
    bits (this repo's tolerance modes against the oracle; equally the oracle against the reference's CUDA
    expf, which is specified to 2 ulp).  The formula weight = T * (1.f - expf(-y)) (rt_kernel.cu:280-281,
    397-398) divides att's last-place error by 1 - att ~ y, and a thin sample has y = delta * sigma *
    delta_scale ~ 1e-3 * sigma: one ulp of att (6e-8) is 6e-5 of such a weight.  Shown here with the oracle
    alone: its own backward with glibc's expf instead of the fixed-sequence one (both <= 1 ulp, equal for
    most arguments) already leaves the 1e-5 x tight-scale band for some entries, and ~1 % of the entries
    move by more than 1e-5 of their own value -- while the OUTPUTS stay within 1e-5 relative + 1e-6.
    The native-math mode of the HIP path (v_exp_f32: 1 ulp, differing from the oracle's bits for most
    arguments) is therefore held to: outputs rtol 1e-5 + atol 1e-6; gradients all within 1e-4 of the tight
    scale -- the a-priori bound ulp(1) / (step_size * min sigma * delta_scale) = 6e-8 / 1e-3 -- and >= 99 %
    within 1e-5 of it (tests/test_gpu_query_and_misc.py)."""
    from svox_t_amd import synth
    c = Case(depth=6, K=28, data_format="SH9", width=160, height=160)
    ot, opt = c.oracle_tree(), c.oracle_opts()
    g = synth.grad_output(c.Q, 4).numpy()
    out0 = O.volume_render(ot, *c.rays_np(), opt)
    w0, ab, tight = O.volume_render_backward(ot, *c.rays_np(), opt, g, want_abs="both")
    try:
        O.use_libm_exp(True)
        out1 = O.volume_render(ot, *c.rays_np(), opt)
        w1 = O.volume_render_backward(ot, *c.rays_np(), opt, g)
    finally:
        O.use_libm_exp(False)
    touched = ab > 0
    err = np.abs(w1 - w0)
    ratio = (err / (tight + 1e-300))[touched]
    assert ratio.max() > 1e-5                                 # the 1e-5 x tight band does not hold across exponentials ...
    assert ratio.max() < 1e-4 and (ratio > 1e-5).mean() < 1e-3      # ... 1e-4 does, and nearly every entry is inside 1e-5
    assert (err > 1e-5 * np.abs(w0))[touched].mean() > 1e-3         # elementwise "1e-5 of its own value": ~1 % miss
    np.testing.assert_allclose(out1, out0, rtol=1e-5, atol=1e-6)    # the outputs hold the north star's figure


def test_counters_and_algorithmic_bytes():
    c = Case(depth=5, K=4, data_format="RGBA", width=64, height=64)
    _, cnt = O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), count=True)
    assert cnt.rays_hit <= c.Q and cnt.active <= cnt.valid <= cnt.steps <= cnt.levels
    assert cnt.levels <= cnt.steps * 5                   # depth-5 tree: at most 5 child reads per step
    bf = O.algorithmic_bytes_forward(cnt, c.Q, 4, 3)
    assert bf == c.Q * 52 + 4 * (cnt.levels + cnt.steps + cnt.valid) + 12 * cnt.active


def test_weight_accum_and_error_scales_of_the_oracle():
    """oracle.volume_render_weights (rt_kernel.cu:266-267, 309-311): every compositing weight lands
    in exactly one leaf slot, so the slot sums add up to the rays' alpha (thresholds 0), internal
    and empty slots stay 0, and the rendered rows are volume_render's.  volume_render_backward's
    two error scales: the tight one (accum priced by the reference's sequential addends) never
    exceeds the sum-of-elementary-magnitudes one, and both cover |grad|."""
    from tests.util import Case
    from svox_t_amd import synth
    c = Case(depth=5, K=13, data_format="SH4", width=48, height=48)
    ot, rays, opt = c.oracle_tree(), c.rays_np(), c.oracle_opts()
    out, w = O.volume_render_weights(ot, *rays, opt)
    np.testing.assert_array_equal(out, O.volume_render(ot, *rays, opt))
    assert w.shape == ot.child.shape and w.min() >= 0
    assert abs(w.sum() - out[:, 3].astype(np.float64).sum()) <= 1e-5 * w.sum()
    assert (w[ot.child != 0] == 0).all()
    assert (w[(ot.data.reshape(ot.child.shape) >= ot.M) | (ot.data.reshape(ot.child.shape) < 0)] == 0).all()
    # early termination drops the weights behind the stopping sample
    _, wf = O.volume_render_weights(ot, *rays, c.oracle_opts(fast=True))
    assert wf.sum() < w.sum() and (wf <= w + 1e-12).all()
    g = synth.grad_output(c.Q, 4).numpy()
    grad, loose, tight = O.volume_render_backward(ot, *rays, opt, g, want_abs="both")
    grad2, loose2 = O.volume_render_backward(ot, *rays, opt, g, want_abs=True)
    np.testing.assert_array_equal(grad, grad2)
    np.testing.assert_allclose(loose, loose2, rtol=1e-12)
    assert (tight <= loose * (1 + 1e-5)).all() and (np.abs(grad) <= tight * (1 + 1e-5) + 1e-30).all()
    assert (tight[:, :-1] == loose[:, :-1]).all() and (tight[:, -1] < loose[:, -1]).any()
