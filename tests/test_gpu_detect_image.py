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
    _C._IMAGE_SHAPES.clear(); _C._IMAGE_TRUST.clear()
    got = _C._detect_image(rs)
    assert got == ((H, W) if expect else None)       # (100 x 164: not multiples of 8; 64 x 56: fewer than 4 096 rays)


def test_shuffled_and_multi_camera_batches_are_not(gpu):
    o, d, v = synth.pinhole_rays(256, 256, c2w=synth.camera_pose(azimuth_deg=50.0))
    perm = torch.randperm(o.shape[0], generator=torch.Generator().manual_seed(1))
    _C._IMAGE_SHAPES.clear(); _C._IMAGE_TRUST.clear()
    assert _C._detect_image(_rays_spec_from_rays(svox.Rays(o[perm].to(gpu), d[perm].to(gpu), v[perm].to(gpu)), None)) is None
    o2, d2, v2 = synth.pinhole_rays(256, 256, c2w=synth.camera_pose(azimuth_deg=120.0))
    oo, dd, vv = torch.cat([o[:32768], o2[:32768]]), torch.cat([d[:32768], d2[:32768]]), torch.cat([v[:32768], v2[:32768]])
    assert _C._detect_image(_rays_spec_from_rays(svox.Rays(oo.to(gpu), dd.to(gpu), vv.to(gpu)), None)) is None   # two origins


def test_recognition_is_not_remembered_by_address(gpu, monkeypatch):
    """A batch freed and another allocated where it lay (what torch's caching allocator does from step to step): the
    remembered answer belongs to the tensor objects, not to their addresses (r05: a shuffled batch at the address of the
    previous step's image was walked as that image -- same results, 3x the time)."""
    o, d, v = synth.pinhole_rays(256, 256, c2w=synth.camera_pose(azimuth_deg=50.0))
    perm = torch.randperm(o.shape[0], generator=torch.Generator().manual_seed(2))
    _C._IMAGE_SHAPES.clear(); _C._IMAGE_TRUST.clear()
    monkeypatch.setattr(_C, "IMAGE_TRUST_AFTER", 1 << 30)          # every answer read, none taken on trust (tested below)
    img = svox.Rays(o.to(gpu), d.to(gpu), v.to(gpu))
    ptrs = (img.origins.data_ptr(), img.dirs.data_ptr())
    assert _C._detect_image(_rays_spec_from_rays(img, None)) == (256, 256)
    del img
    hits = 0
    for _ in range(4):                                   # (the allocator hands the freed blocks back: usually at once)
        sh = svox.Rays(o[perm].to(gpu), d[perm].to(gpu), v[perm].to(gpu))
        hits += (sh.origins.data_ptr(), sh.dirs.data_ptr()) == ptrs
        assert _C._detect_image(_rays_spec_from_rays(sh, None)) is None
        del sh
    print(f"batches that reused the image's addresses: {hits} of 4")
    # and the other way round: the same tensors, rewritten in place (the version counter says so)
    buf = svox.Rays(o[perm].to(gpu), d[perm].to(gpu), v[perm].to(gpu))
    assert _C._detect_image(_rays_spec_from_rays(buf, None)) is None
    buf.origins.copy_(o.to(gpu)); buf.dirs.copy_(d.to(gpu)); buf.viewdirs.copy_(v.to(gpu))
    assert _C._detect_image(_rays_spec_from_rays(buf, None)) == (256, 256)


def test_answers_taken_on_trust(gpu, monkeypatch):
    """New tensors every step, as a training loop hands them over: after IMAGE_TRUST_AFTER equal answers the next
    batch of that size is taken to be the same without waiting for its probe, which is read one batch later; a probe
    that disagrees ends the trust.  A batch taken for an image it is not renders the same bits (the walk is a
    bijection of the rays whatever they are), only slower."""
    c = Case(depth=6, K=28, data_format="SH9", width=128, height=128)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    o, d, v = (torch.from_numpy(a).to(gpu) for a in c.rays_np())
    Q = o.shape[0]
    perm = torch.randperm(Q, generator=torch.Generator().manual_seed(5)).to(gpu)
    fresh = lambda shuffled: svox.Rays(*((t[perm].contiguous() if shuffled else t.clone()) for t in (o, d, v)))
    probe = lambda rays: _C._detect_image(_rays_spec_from_rays(rays, None))
    _C._IMAGE_SHAPES.clear(); _C._IMAGE_TRUST.clear()
    for i in range(_C.IMAGE_TRUST_AFTER):
        assert probe(fresh(False)) == (128, 128) and _C._IMAGE_TRUST[Q][2] is None      # read, each of them
    assert probe(fresh(False)) == (128, 128) and _C._IMAGE_TRUST[Q][2] is not None      # taken on trust; its probe is under way
    torch.cuda.synchronize()
    sh = fresh(True)
    with torch.no_grad():
        out_trusted = r(tree.features, sh)                    # walked as a 128 x 128 image, which it is not
    assert _C._IMAGE_SHAPES[id(sh.dirs)][3] == (128, 128)
    torch.cuda.synchronize()
    assert probe(fresh(True)) is None                         # the shuffled batch's probe has arrived and disagreed: read again
    assert _C._IMAGE_TRUST[Q][:2] == [None, 2]               # (the disagreement, then the read that confirmed it)
    assert _C._IMAGE_SHAPES[id(sh.dirs)][3] is None and probe(sh) is None     # the batch that was guessed wrong is put right: tensors that stay are not walked wrongly for good
    # ... also when nothing but the wrongly guessed tensors themselves ever comes by again
    _C._IMAGE_TRUST[Q][:] = [(128, 128), _C.IMAGE_TRUST_AFTER, None]
    sh2 = fresh(True)
    assert probe(sh2) == (128, 128)
    torch.cuda.synchronize()
    assert probe(sh2) is None
    monkeypatch.setattr(_C, "DETECT_IMAGES", False)
    with torch.no_grad():
        out_sorted = r(tree.features, svox.Rays(sh.origins, sh.dirs, sh.viewdirs))
    np.testing.assert_array_equal(out_trusted.cpu().numpy(), out_sorted.cpu().numpy())
    want = O.volume_render(c.oracle_tree(), *(a[perm.cpu().numpy()] for a in c.rays_np()), c.oracle_opts())
    np.testing.assert_array_equal(out_trusted.cpu().numpy(), want)


@pytest.mark.parametrize("detect", [True, False])
def test_undeclared_image_renders_the_same_either_way(gpu, detect, monkeypatch):
    monkeypatch.setattr(_C, "DETECT_IMAGES", detect)
    _C._IMAGE_SHAPES.clear(); _C._IMAGE_TRUST.clear()
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
