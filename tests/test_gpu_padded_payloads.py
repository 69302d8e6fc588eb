"""Payloads one step away from a specialised one are rendered AS the next specialised one with dummy channels (r05;
svox_t_amd/csrc/__init__.py, PAD_PAYLOADS): one or two channels with a basis -> three, RGBA-style rows of other widths
-> 4 / 8 / 16 / 32 floats.  The reference is generic in the channel count (rt_kernel.cu:293-306, 410-425, 470-476); what
must hold is what holds for every route: forward bit for bit against the oracle (and against the generic kernels),
gradient on the tight scale, nothing in the dummy columns leaking out."""
import numpy as np
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from oracle import oracle as O
from svox_t_amd import synth
from tests.util import Case, assert_grads_close

pytestmark = pytest.mark.gpu

PAYLOADS = [
    ("SH4", 9, None),        # two channels x SH4
    ("SH9", 10, None),       # one channel x SH9
    ("SH1", 3, None),        # two channels x SH1
    ("SH16", 33, None),      # two channels x SH16
    ("RGBA", 6, None), ("RGBA", 3, None), ("RGBA", 2, None), ("RGBA", 12, None), ("RGBA", 21, None), ("RGBA", 31, None),
    ("SG4", 9, "SG"),        # two channels x four spherical gaussians
]


def _tree(c, fmt, lobes, gpu):
    return svox.N3Tree.from_arrays(c.st.child, c.st.data, c.st.parent_depth, c.features, data_format=fmt,
                                   extra_data=lobes, device=gpu)


@pytest.mark.parametrize("image", [True, False])
@pytest.mark.parametrize("fmt,K,kind", PAYLOADS)
def test_padded_payload_matches_oracle_and_generic_kernels(gpu, fmt, K, kind, image, monkeypatch):
    c = Case(depth=5, K=K, data_format=fmt, width=64, height=48)
    lobes = None
    if kind == "SG":
        gen = torch.Generator().manual_seed(4)
        lobes = torch.cat([torch.rand(4, 1, generator=gen) * 4 + 0.5,
                           torch.nn.functional.normalize(torch.randn(4, 3, generator=gen), dim=-1)], -1).contiguous()
    ot = O.Tree(c.features.numpy(), c.st.data, c.st.child, extra=None if lobes is None else lobes.numpy())
    for th in ((0.0, 0.0), (1e-2, 1e-2)):
        opt = O.make_options(format=c.format, basis_dim=c.basis_dim, sigma_thresh=th[0], stop_thresh=th[1])
        want = O.volume_render(ot, *c.rays_np(), opt)
        cols = want.shape[1]
        g = synth.grad_output(c.Q, cols, seed=5)
        gwant, _, tight = O.volume_render_backward(ot, *c.rays_np(), opt, g.numpy(), want_abs="both")
        outs = {}
        for pad in (True, False):
            monkeypatch.setattr(_C, "PAD_PAYLOADS", pad)
            tree = _tree(c, fmt, lobes, gpu)
            r = svox.VolumeRenderer(tree)
            r.sigma_thresh, r.stop_thresh = th
            f = tree.features
            out = r(f, c.rays_gpu(gpu), image_shape=(48, 64) if image else None)
            assert out.shape == (c.Q, cols)
            out.backward(g.to(gpu))
            np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
            assert_grads_close(f.grad.cpu().numpy(), gwant, tight)
            assert f.grad.shape == (c.st.n_features, K)
            outs[pad] = (_C.LAST_ROUTE["forward"], _C.LAST_ROUTE["backward"])
        assert "generic" in outs[False][0] or "marches" in outs[False][1], outs
        assert "generic" not in outs[True][0], outs
        if kind is None or image:       # (SG / ASG lists serve the per-tile backward alone: a small unsorted ray batch marches -- with the lobes kernel)
            assert "marches" not in outs[True][1], outs


def test_padded_payload_through_camera_mode_and_plain_calls(gpu):
    """render_persp and the reference-shaped two plain calls take the padded route too (the plan the forward leaves on the
    spec names the padded table; the backward finds it)."""
    c = Case(depth=5, K=9, data_format="SH4", width=64, height=48)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    pose = synth.camera_pose(azimuth_deg=30.0)
    fx = 1111.111 * 64 / 800.0
    f = tree.features
    img = r.render_persp(f, torch.from_numpy(pose).float().to(gpu), width=64, height=48, fx=fx)
    assert img.shape == (48, 64, 3)
    g = synth.grad_output(c.Q, 3, seed=5)
    img.view(-1, 3).backward(g.to(gpu))
    assert "grad_fused_kernel" in _C.LAST_ROUTE["backward"], _C.LAST_ROUTE
    o, d, v = O.camera_rays(pose.astype(np.float32), fx, fx, 64, 48)
    opt = c.oracle_opts()
    np.testing.assert_array_equal(img.view(-1, 3).detach().cpu().numpy(), O.volume_render(c.oracle_tree(), o, d, v, opt))
    gwant, _, tight = O.volume_render_backward(c.oracle_tree(), o, d, v, opt, g.numpy(), want_abs="both")
    assert_grads_close(f.grad.cpu().numpy(), gwant, tight)
    # a backward without its forward's plan (a fresh spec) still gives the reference's gradient
    from svox_t_amd.renderer import _rays_spec_from_rays
    rs = _rays_spec_from_rays(svox.Rays(*(torch.from_numpy(a).to(gpu) for a in (o, d, v))), (48, 64))
    grad = _C.volume_render_backward(tree._spec(tree.features), rs, r._get_options(), g.to(gpu))
    assert_grads_close(grad.cpu().numpy(), gwant, tight)


@pytest.mark.parametrize("fmt,K,lo,hi", [("SH9", 28, 1, 4), ("SH9", 28, 0, 0), ("SH4", 9, 1, 3), ("SH16", 49, 4, 8), ("SH9", 10, 2, 8)])
def test_component_subrange_through_the_fast_kernels(gpu, fmt, K, lo, hi, monkeypatch):
    """min_comp / max_comp (rt_kernel.cu:295-298): coefficients outside the range are zeros in the copy the kernels see;
    gradient columns outside it come back as exact zeros, as the reference leaves them."""
    c = Case(depth=5, K=K, data_format=fmt, width=64, height=48)
    opt = c.oracle_opts(min_comp=lo, max_comp=hi)
    want = O.volume_render(c.oracle_tree(), *c.rays_np(), opt)
    g = synth.grad_output(c.Q, want.shape[1], seed=5)
    gwant, gabs, tight = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), opt, g.numpy(), want_abs="both")
    for pad in (True, False):
        monkeypatch.setattr(_C, "PAD_PAYLOADS", pad)
        tree = c.tree(gpu)
        r = svox.VolumeRenderer(tree, min_comp=lo, max_comp=hi)
        f = tree.features
        out = r(f, c.rays_gpu(gpu), image_shape=(48, 64))
        out.backward(g.to(gpu))
        np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
        assert_grads_close(f.grad.cpu().numpy(), gwant, tight)
        assert ("generic" in _C.LAST_ROUTE["forward"]) == (not pad), _C.LAST_ROUTE
    bd = c.basis_dim
    outside = [ch * bd + i for ch in range((K - 1) // bd) for i in range(bd) if not lo <= i <= hi]
    assert np.all(f.grad.cpu().numpy()[:, outside] == 0) and np.all(gabs[:, outside] == 0)
