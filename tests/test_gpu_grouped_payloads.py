"""SH payloads of more than three channels are rendered in groups of three, RGBA-style rows of more than 32 floats in groups of
31 channels, each group as the specialised payload (r05; svox_t_amd/csrc/__init__.py, GROUP_PAYLOADS).  The reference is generic in the channel count (rt_kernel.cu:293-306,
410-425, 470-476).  What must hold is what holds for every route: forward bit for bit against the oracle (channels are
independent), gradient within 1e-5 of the tight scale (its sigma entries are sums of per-group terms where the reference
forms one total_color over all channels first), the same as the generic kernels give, through every way of calling."""
import numpy as np
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from oracle import oracle as O
from svox_t_amd import synth
from tests.util import Case, assert_grads_close

pytestmark = pytest.mark.gpu

PAYLOADS = [("SH9", 37), ("SH4", 21), ("SH1", 8), ("SH16", 65), ("SH4", 49),      # 4 x SH9, 5 x SH4, 7 x SH1, 4 x SH16, 12 x SH4
            ("RGBA", 33), ("RGBA", 40), ("RGBA", 64), ("RGBA", 95)]                 # rows of 33 / 40 / 64 / 95 floats: groups of 31 channels


def _lobes(fmt, n):
    if not fmt.startswith(("SG", "ASG")):
        return None
    gen = torch.Generator().manual_seed(4)
    if fmt.startswith("SG"):
        return torch.cat([torch.rand(n, 1, generator=gen) * 4 + 0.5,
                          torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)], -1).contiguous()
    z = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    x = torch.nn.functional.normalize(torch.cross(z, torch.randn(n, 3, generator=gen), dim=-1), dim=-1)
    y = torch.cross(z, x, dim=-1)
    return torch.cat([torch.rand(n, 2, generator=gen) * 3 + 0.5, x, y, z], -1).contiguous()        # (lambda, mu, x, y, z)


@pytest.mark.parametrize("image", [True, False])
@pytest.mark.parametrize("fmt,K", PAYLOADS + [("SG4", 17), ("ASG4", 21)])
def test_grouped_payload_matches_oracle_and_generic_kernels(gpu, fmt, K, image, monkeypatch):
    c = Case(depth=5, K=K, data_format=fmt, width=64, height=48)
    lobes = _lobes(fmt, c.basis_dim)
    ot = c.oracle_tree() if lobes is None else O.Tree(c.features.numpy(), c.st.data, c.st.child, extra=lobes.numpy())
    for th in ((0.0, 0.0), (1e-2, 1e-2)):
        opt = O.make_options(format=c.format, basis_dim=c.basis_dim, sigma_thresh=th[0], stop_thresh=th[1])
        want = O.volume_render(ot, *c.rays_np(), opt)
        cols = want.shape[1]
        assert cols == (K - 1) // max(c.basis_dim, 1) + 1
        g = synth.grad_output(c.Q, cols, seed=5)
        gwant, _, tight = O.volume_render_backward(ot, *c.rays_np(), opt, g.numpy(), want_abs="both")
        routes = {}
        for grouped in (True, False):
            monkeypatch.setattr(_C, "GROUP_PAYLOADS", grouped)
            tree = c.tree(gpu) if lobes is None else svox.N3Tree.from_arrays(c.st.child, c.st.data, c.st.parent_depth, c.features,
                                                                             data_format=fmt, extra_data=lobes, device=gpu)
            r = svox.VolumeRenderer(tree)
            r.sigma_thresh, r.stop_thresh = th
            f = tree.features
            out = r(f, c.rays_gpu(gpu), image_shape=(48, 64) if image else None)
            assert out.shape == (c.Q, cols)
            out.backward(g.to(gpu))
            np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
            assert_grads_close(f.grad.cpu().numpy(), gwant, tight)
            assert f.grad.shape == (c.st.n_features, K)
            routes[grouped] = (_C.LAST_ROUTE["forward"], _C.LAST_ROUTE["backward"])
        assert "generic" in routes[False][0], routes
        assert "generic" not in routes[True][0], routes
        if lobes is None or image:      # (SG / ASG lists serve the per-tile backward alone: a small unsorted ray batch marches -- with the lobes kernel)
            assert "marches" not in routes[True][1], routes


def test_grouped_payload_through_camera_mode_no_grad_and_twice(gpu):
    c = Case(depth=5, K=37, data_format="SH9", width=64, height=48)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    f = tree.features
    ot = c.oracle_tree()
    want = O.volume_render(ot, *c.rays_np(), c.oracle_opts())
    with torch.no_grad():
        np.testing.assert_array_equal(r(f, c.rays_gpu(gpu)).cpu().numpy(), want)
        np.testing.assert_array_equal(r(f, c.rays_gpu(gpu), fast=True).cpu().numpy(), O.volume_render(ot, *c.rays_np(), c.oracle_opts(fast=True)))
    # two forwards on the same rays object, then the backward of the second; then a second backward (marches: the lists served one)
    rays = c.rays_gpu(gpu)
    g = synth.grad_output(c.Q, 5, seed=2)
    gwant, _, tight = O.volume_render_backward(ot, *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs="both")
    r(f, rays, image_shape=(48, 64))
    out = r(f, rays, image_shape=(48, 64))
    out.backward(g.to(gpu), retain_graph=True)
    assert_grads_close(f.grad.cpu().numpy(), gwant, tight)
    f.grad = None
    out.backward(g.to(gpu))
    assert_grads_close(f.grad.cpu().numpy(), gwant, tight)
    # render_persp: the image of a camera, rays generated in the kernels
    pose = synth.camera_pose(azimuth_deg=30.0)
    fx = 1111.111 * 64 / 800.0
    f.grad = None
    img = r.render_persp(f, torch.from_numpy(pose).float().to(gpu), width=64, height=48, fx=fx)
    assert img.shape == (48, 64, 5)
    o, d, v = O.camera_rays(pose.astype(np.float32), fx, fx, 64, 48)
    np.testing.assert_array_equal(img.detach().cpu().numpy().reshape(-1, 5), O.volume_render(ot, o, d, v, c.oracle_opts()))
    img.backward(g.to(gpu).view(48, 64, 5))
    gw2, _, tight2 = O.volume_render_backward(ot, o, d, v, c.oracle_opts(), g.numpy(), want_abs="both")
    assert_grads_close(f.grad.cpu().numpy(), gw2, tight2)


def test_grouped_payload_at_the_headline_size(gpu, capsys):
    """800 x 800 / depth 8, four channels x SH9 (K = 37): the oracle at full size, and the time beside the generic kernels'."""
    import time
    c = Case(depth=8, K=37, data_format="SH9", width=800, height=800)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    f = tree.features
    rays = c.rays_gpu(gpu)
    g = synth.grad_output(c.Q, 5, seed=3)
    gg = g.to(gpu)
    out = r(f, rays, image_shape=(800, 800))
    out.backward(gg)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts()))
    gw, _, tight = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs="both")
    assert_grads_close(f.grad.cpu().numpy(), gw, tight)

    def timed(n=10):
        for _ in range(3):
            f.grad = None
            r(f, rays, image_shape=(800, 800)).backward(gg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            f.grad = None
            r(f, rays, image_shape=(800, 800)).backward(gg)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    t_grouped = timed()
    _C.GROUP_PAYLOADS = False
    try:
        t_generic = timed(3)
    finally:
        _C.GROUP_PAYLOADS = True
    with capsys.disabled():
        print(f"\n[SH9 x 4 channels, 800 x 800 / depth 8] forward+backward in two groups of three {t_grouped:.3f} ms, generic kernels {t_generic:.3f} ms")
    assert t_grouped < 0.5 * t_generic
