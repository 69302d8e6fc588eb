"""Forward + backward under non-zero thresholds (`fast=True`: sigma_thresh = stop_thresh = 1e-2, svox_t/renderer.py:428-430;
or renderer.sigma_thresh / .stop_thresh set by the user, :435-438) through the RECORDING forward (r05).

The reference's forward skips samples with sigma <= sigma_thresh (rt_kernel.cu:279) and ends a ray once T <= stop_thresh
(:313-319, rescaling by 1 / (1 - T)); its backward ignores both (:382, 456: every sample with sigma > 0, to the end of the
ray).  The lists therefore hold the backward's set and the forward applies its own rules while it composites: the forward
must equal the oracle's thresholded forward BIT FOR BIT and the gradient the oracle's (threshold-free) backward on the
tight scale -- for every payload family, list capacity (tails), the tiled and the sorted walk."""
import numpy as np
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from oracle import oracle as O
from svox_t_amd import synth
from tests.util import Case, assert_grads_close

pytestmark = pytest.mark.gpu

CASES = {
    "d5_rgba4": dict(depth=5, K=4, data_format="RGBA", width=64, height=64),
    "d5_sh9": dict(depth=5, K=28, data_format="SH9", width=64, height=64),
    "d6_rgba32": dict(depth=6, K=32, data_format="RGBA", width=96, height=96),
    "d5_rgba8": dict(depth=5, K=8, data_format="RGBA", width=64, height=64),
    "d5_sh4_world": dict(depth=5, K=13, data_format="SH4", width=64, height=64, radius=[1.0, 1.2, 0.8], center=[0.1, -0.2, 0.3]),
    "d4_sh16": dict(depth=4, K=49, data_format="SH16", width=48, height=48),
    "d4_sh25": dict(depth=4, K=76, data_format="SH25", width=48, height=48),
    "d5_two_channels": dict(depth=5, K=9, data_format="SH4", width=48, height=48),      # (r05: rendered as three channels, lists and all)
}


def _step(case, gpu, image, sigma_thresh, stop_thresh, opacity=False):
    tree = case.tree(gpu)
    r = svox.VolumeRenderer(tree)
    r.sigma_thresh, r.stop_thresh = sigma_thresh, stop_thresh       # (renderer.py:435-438: user attributes override)
    W = int(round(case.Q ** 0.5))
    shape = (case.Q // W, W) if image else None
    f = tree.features
    out = (r.opacity_render if opacity else r)(f, case.rays_gpu(gpu), image_shape=shape)
    g = synth.grad_output(case.Q, out.shape[1], seed=7)
    out.backward(g.to(gpu))
    torch.cuda.synchronize()
    opt = O.make_options(format=case.format, basis_dim=case.basis_dim, sigma_thresh=sigma_thresh, stop_thresh=stop_thresh)
    return out.detach().cpu().numpy(), f.grad.cpu().numpy(), g.numpy(), opt


@pytest.mark.parametrize("image", [True, False])
@pytest.mark.parametrize("name", list(CASES))
def test_fast_forward_backward(gpu, name, image):
    case = Case(**CASES[name])
    got, ggot, g, opt = _step(case, gpu, image, 1e-2, 1e-2)
    want = O.volume_render(case.oracle_tree(), *case.rays_np(), opt)
    np.testing.assert_array_equal(got, want)
    gwant, gabs, tight = O.volume_render_backward(case.oracle_tree(), *case.rays_np(), opt, g, want_abs="both")
    assert_grads_close(ggot, gwant, tight)
    assert "marches" not in _C.LAST_ROUTE["backward"], _C.LAST_ROUTE         # the lists served it
    # the thresholds did something: the thresholded forward differs from the plain one on this workload
    plain = O.volume_render(case.oracle_tree(), *case.rays_np(), O.make_options(format=case.format, basis_dim=case.basis_dim))
    assert not np.array_equal(want, plain)


@pytest.mark.parametrize("thresholds", [(0.5, 0.0), (0.0, 0.3), (3.0, 0.6), (1e-2, 0.999)])
@pytest.mark.parametrize("name", ["d5_sh9", "d6_rgba32", "d5_rgba4"])
def test_thresholds_apart_and_large(gpu, name, thresholds):
    """Each rule alone, and values at which most samples are skipped / most rays stop after a sample or two."""
    case = Case(**CASES[name])
    got, ggot, g, opt = _step(case, gpu, True, *thresholds)
    np.testing.assert_array_equal(got, O.volume_render(case.oracle_tree(), *case.rays_np(), opt))
    gwant, _, tight = O.volume_render_backward(case.oracle_tree(), *case.rays_np(), opt, g, want_abs="both")
    assert_grads_close(ggot, gwant, tight)


@pytest.mark.parametrize("list_samples", [8, 16])
@pytest.mark.parametrize("name", ["d5_sh9", "d6_rgba32", "d5_rgba4", "d4_sh25"])
def test_fast_with_short_lists(gpu, name, list_samples, monkeypatch):
    """Lists that overflow: a ray the stop rule ended keeps its overflow flag for the BACKWARD's tail (which marches
    every sigma > 0 the list could not hold) while the forward's tail leaves it alone (aux.w)."""
    monkeypatch.setattr(_C, "BWD_LIST_SAMPLES", list_samples)
    case = Case(**CASES[name])
    for th in ((1e-2, 1e-2), (0.0, 0.5), (1.0, 0.0)):
        got, ggot, g, opt = _step(case, gpu, True, *th)
        np.testing.assert_array_equal(got, O.volume_render(case.oracle_tree(), *case.rays_np(), opt))
        gwant, _, tight = O.volume_render_backward(case.oracle_tree(), *case.rays_np(), opt, g, want_abs="both")
        assert_grads_close(ggot, gwant, tight)


@pytest.mark.parametrize("image", [True, False])
def test_opacity_with_thresholds(gpu, image):
    case = Case(depth=6, K=28, data_format="SH9", width=96, height=96)
    for th in ((1e-2, 1e-2), (0.0, 0.4), (2.0, 0.0)):
        got, ggot, g, opt = _step(case, gpu, image, *th, opacity=True)
        np.testing.assert_array_equal(got, O.opacity_render(case.oracle_tree(), *case.rays_np(), opt))
        gwant, _, tight = O.volume_render_backward(case.oracle_tree(), *case.rays_np(), opt, g, want_abs="both")
        assert_grads_close(ggot, gwant, tight)
        assert np.all(ggot[:, :-1] == 0)


def test_fast_view_rotations(gpu):
    """Per-leaf view rotations (transformation_matrices) under fast=True: the one-launch XF forward and its fused backward."""
    case = Case(depth=5, K=28, data_format="SH9", width=64, height=64)
    tree = case.tree(gpu)
    gen = torch.Generator().manual_seed(3)
    A = torch.linalg.qr(torch.randn(tree.features.shape[0], 3, 3, generator=gen))[0].contiguous()
    r = svox.VolumeRenderer(tree)
    f = tree.features
    out = r(f, case.rays_gpu(gpu), transformation_matrices=A.to(gpu), image_shape=(64, 64), fast=True)
    g = synth.grad_output(case.Q, 4, seed=7)
    out.backward(g.to(gpu))
    ot, opt = case.oracle_tree(), case.oracle_opts(fast=True)
    with O.transformation_matrices(A.numpy()):
        want = O.volume_render(ot, *case.rays_np(), opt)
        gwant, _, tight = O.volume_render_backward(ot, *case.rays_np(), opt, g.numpy(), want_abs="both")
    np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
    assert_grads_close(f.grad.cpu().numpy(), gwant, tight)
    assert _C.LAST_ROUTE["forward"].startswith("fwd_roles_kernel<XF>"), _C.LAST_ROUTE


def test_single_march_mode_is_not_taken_with_thresholds(gpu, monkeypatch):
    """SVOXT_BWD_EXACT=0 (tolerance mode: accum from the forward's output) differentiates the forward's own sum -- with
    thresholds that is not the sum the reference's backward forms: the exact arithmetic is taken instead."""
    monkeypatch.setattr(_C, "BWD_EXACT", False)
    case = Case(**CASES["d5_sh9"])
    got, ggot, g, opt = _step(case, gpu, True, 1e-2, 1e-2)
    gwant, _, tight = O.volume_render_backward(case.oracle_tree(), *case.rays_np(), opt, g, want_abs="both")
    assert_grads_close(ggot, gwant, tight)


def test_fast_full_size_config3(gpu):
    """BASELINE configs[2] with fast=True: forward bit for bit, gradient on the tight scale, at 800 x 800 / depth 8 / SH9."""
    case = Case(depth=8, K=28, data_format="SH9", width=800, height=800)
    got, ggot, g, opt = _step(case, gpu, True, 1e-2, 1e-2)
    assert "fwd_roles_kernel" in _C.LAST_ROUTE["forward"] and "grad_fused_kernel" in _C.LAST_ROUTE["backward"], _C.LAST_ROUTE
    np.testing.assert_array_equal(got, O.volume_render(case.oracle_tree(), *case.rays_np(), opt))
    gwant, _, tight = O.volume_render_backward(case.oracle_tree(), *case.rays_np(), opt, g, want_abs="both")
    assert_grads_close(ggot, gwant, tight)


@pytest.mark.parametrize("kind,B", [("SG", 9), ("ASG", 4), ("SG", 16)])
@pytest.mark.parametrize("image", [True, False])
def test_fast_lobes_payloads(gpu, kind, B, image):
    """SG / ASG payloads (lists only together with the per-tile backward's hand-over, svoxt_can_record = 2) under
    fast=True: image batches take the one-launch forward and the fused backward, ray batches march."""
    gen = torch.Generator().manual_seed(4)
    if kind == "SG":
        lobes = torch.cat([torch.rand(B, 1, generator=gen) * 4 + 0.5,
                           torch.nn.functional.normalize(torch.randn(B, 3, generator=gen), dim=-1)], -1).contiguous()
        fmt = O.FORMAT_SG
    else:
        fr = torch.linalg.qr(torch.randn(B, 3, 3, generator=gen))[0]
        lobes = torch.cat([torch.rand(B, 2, generator=gen) * 3 + 0.3, fr.reshape(B, 9)], -1).contiguous()
        fmt = O.FORMAT_ASG
    c = Case(depth=5, K=3 * B + 1, data_format=f"{kind}{B}", width=64, height=48)
    tree = svox.N3Tree.from_arrays(c.st.child, c.st.data, c.st.parent_depth, c.features, data_format=f"{kind}{B}",
                                   extra_data=lobes, device=gpu)
    r = svox.VolumeRenderer(tree)
    f = tree.features
    out = r(f, c.rays_gpu(gpu), image_shape=(48, 64) if image else None, fast=True)
    g = synth.grad_output(c.Q, 4, seed=7)
    out.backward(g.to(gpu))
    ot = O.Tree(c.features.numpy(), c.st.data, c.st.child, extra=lobes.numpy())
    opt = O.make_options(format=fmt, basis_dim=B, sigma_thresh=1e-2, stop_thresh=1e-2)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.volume_render(ot, *c.rays_np(), opt))
    gwant, _, tight = O.volume_render_backward(ot, *c.rays_np(), opt, g.numpy(), want_abs="both")
    assert_grads_close(f.grad.cpu().numpy(), gwant, tight)
    if image:
        assert "grad_fused_kernel" in _C.LAST_ROUTE["backward"], _C.LAST_ROUTE


def test_fast_render_persp(gpu):
    """Camera mode (rays generated in the kernels) forward + backward with fast=True, against the oracle on the rays its
    restatement of cam2world_ray (rt_kernel.cu:1153-1166) generates: pixels bit for bit, gradient on the tight scale."""
    case = Case(depth=6, K=28, data_format="SH9", width=96, height=64)
    tree = case.tree(gpu)
    r = svox.VolumeRenderer(tree)
    pose = synth.camera_pose(azimuth_deg=30.0)
    fx = 1111.111 * 96 / 800.0
    g = synth.grad_output(96 * 64, 4, seed=7)
    f = tree.features
    img = r.render_persp(f, torch.from_numpy(pose).float().to(gpu), width=96, height=64, fx=fx, fast=True)
    img.view(-1, 4).backward(g.to(gpu))
    assert "marches" not in _C.LAST_ROUTE["backward"], _C.LAST_ROUTE
    o, d, v = O.camera_rays(pose.astype(np.float32), fx, fx, 96, 64)
    opt = O.make_options(format=case.format, basis_dim=case.basis_dim, sigma_thresh=1e-2, stop_thresh=1e-2)
    np.testing.assert_array_equal(img.view(-1, 4).detach().cpu().numpy(), O.volume_render(case.oracle_tree(), o, d, v, opt))
    gwant, _, tight = O.volume_render_backward(case.oracle_tree(), o, d, v, opt, g.numpy(), want_abs="both")
    assert_grads_close(f.grad.cpu().numpy(), gwant, tight)
