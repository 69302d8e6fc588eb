"""A ray batch that is a row-major pinhole image but is not declared one -- what a caller of the reference's API hands
over -- is recognised and walked in 8 x 8 pixel tiles instead of being sorted (svox_t_amd/csrc/__init__.py,
_detect_image).  Results are per ray: the same bits either way; what is tested is that the recognition fires where it
should, does not where it should not, and that nothing depends on it."""
import numpy as np
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from oracle import oracle as O
from svox_t_amd import synth
from svox_t_amd.renderer import _rays_spec_from_rays
from tests.util import Case, assert_grads_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H,expect", [(800, 800, True), (1024, 512, True), (136, 200, True), (100, 164, False), (64, 56, False)])
def test_pinhole_images_are_recognised(gpu, W, H, expect):
    o, d, v = synth.pinhole_rays(W, H, c2w=synth.camera_pose(azimuth_deg=50.0))
    rs = _rays_spec_from_rays(svox.Rays(o.to(gpu), d.to(gpu), v.to(gpu)), None)
    _C._IMAGE_SHAPES.clear()
    got = _C._detect_image(rs)
    assert got == ((H, W) if expect else None)       # (100 x 164: not multiples of 8; 64 x 56: fewer than 4 096 rays)


def test_shuffled_and_multi_camera_batches_are_not(gpu):
    o, d, v = synth.pinhole_rays(256, 256, c2w=synth.camera_pose(azimuth_deg=50.0))
    perm = torch.randperm(o.shape[0], generator=torch.Generator().manual_seed(1))
    _C._IMAGE_SHAPES.clear()
    assert _C._detect_image(_rays_spec_from_rays(svox.Rays(o[perm].to(gpu), d[perm].to(gpu), v[perm].to(gpu)), None)) is None
    o2, d2, v2 = synth.pinhole_rays(256, 256, c2w=synth.camera_pose(azimuth_deg=120.0))
    oo, dd, vv = torch.cat([o[:32768], o2[:32768]]), torch.cat([d[:32768], d2[:32768]]), torch.cat([v[:32768], v2[:32768]])
    assert _C._detect_image(_rays_spec_from_rays(svox.Rays(oo.to(gpu), dd.to(gpu), vv.to(gpu)), None)) is None   # two origins


@pytest.mark.parametrize("detect", [True, False])
def test_undeclared_image_renders_the_same_either_way(gpu, detect, monkeypatch):
    monkeypatch.setattr(_C, "DETECT_IMAGES", detect)
    _C._IMAGE_SHAPES.clear()
    c = Case(depth=6, K=28, data_format="SH9", width=136, height=200)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    f = tree.features
    out = r(f, c.rays_gpu(gpu))                               # no image_shape: the reference's call
    g = synth.grad_output(c.Q, 4, seed=3)
    out.backward(g.to(gpu))
    assert "grad_fused_kernel" in _C.LAST_ROUTE["backward"], _C.LAST_ROUTE      # tiles (detected) or the sorted order: per-tile either way
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts()))
    gw, _, tight = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs="both")
    assert_grads_close(f.grad.cpu().numpy(), gw, tight)
