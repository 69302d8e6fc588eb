"""The acceleration grid in 4 x 4 x 4 bricks (include/svoxt.h, SVOXT_ACCEL_BRICKS; r05): a layout of a cache -- every
result is the same bits in either layout, at every resolution, through every kernel that reads the grid -- chosen per
grid by the operator layer: a recording forward (its march waits for the cell load alone) goes through a bricked grid,
everything else through the row-major one, and a tree rendered both ways keeps both."""
import ctypes

import numpy as np
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from oracle import oracle as O
from svox_t_amd import synth
from tests.util import Case, assert_grads_close

pytestmark = pytest.mark.gpu

CASES = {"d6_sh9": dict(depth=6, K=28, data_format="SH9", width=96, height=96),
         "d5_rgba4": dict(depth=5, K=4, data_format="RGBA", width=64, height=64),
         "d6_rgba8_world": dict(depth=6, K=8, data_format="RGBA", width=64, height=72, radius=(0.7, 0.5, 0.9), center=(0.1, -0.2, 0.3))}


@pytest.mark.parametrize("bricks", [True, False])
@pytest.mark.parametrize("g", [None, 2, 3, 5])
@pytest.mark.parametrize("name", list(CASES))
def test_every_route_gives_the_oracles_bits_in_either_layout(gpu, name, g, bricks, monkeypatch):
    monkeypatch.setattr(_C, "ACCEL_BRICKS", bricks)
    if g is not None:
        monkeypatch.setattr(_C, "ACCEL_LOG2", g)
    c = Case(**CASES[name])
    tree, ot = c.tree(gpu), c.oracle_tree()
    r = svox.VolumeRenderer(tree)
    rays, rays_np = c.rays_gpu(gpu), c.rays_np()
    for fast in (False, True):
        with torch.no_grad():
            out = r(tree.features, rays, fast=fast)                    # forward alone (one kernel, or march + shade for wide rows)
        np.testing.assert_array_equal(out.cpu().numpy(), O.volume_render(ot, *rays_np, c.oracle_opts(fast=fast)))
    with torch.no_grad():
        np.testing.assert_array_equal(r.render_depth(tree.features, rays).cpu().numpy(), O.render_depth(ot, *rays_np, c.oracle_opts()))
    f = tree.features.detach().clone().requires_grad_(True)
    grad = synth.grad_output(c.Q, out.shape[1], seed=7)
    out = r(f, rays, image_shape=(CASES[name]["height"], CASES[name]["width"]))       # recording forward + the backward over its lists
    out.backward(grad.to(gpu))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.volume_render(ot, *rays_np, c.oracle_opts()))
    gw, _, tight = O.volume_render_backward(ot, *rays_np, c.oracle_opts(), grad.numpy(), want_abs="both")
    assert_grads_close(f.grad.cpu().numpy(), gw, tight)
    # ... and the backward that marches by itself (no lists: the reference's two passes, through the grid)
    monkeypatch.setattr(_C, "AUTO_PLAN", False)
    f2 = tree.features.detach().clone().requires_grad_(True)
    r(f2, rays).backward(grad.to(gpu))
    assert_grads_close(f2.grad.cpu().numpy(), gw, tight)
    ents = [k for k in _C._ACCEL_CACHE if k[0] == id(tree.child)]
    assert ents and all(k[1] == bricks for k in ents), ents                           # the forced layout is the one that was built


def test_the_layout_follows_the_kind_of_forward(gpu):
    """Default policy: a training step's grid in bricks, a forward-only view's row-major; both cached side by side; the
    backward takes the grid its forward went through (no third build)."""
    c = Case(depth=6, K=28, data_format="SH9", width=64, height=64)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    rays = c.rays_gpu(gpu)
    key = lambda b: (id(tree.child), b)
    with torch.no_grad():
        a = r(tree.features, rays)
    assert key(False) in _C._ACCEL_CACHE and key(True) not in _C._ACCEL_CACHE
    f = tree.features.detach().clone().requires_grad_(True)
    b = r(f, rays, image_shape=(64, 64))
    assert key(True) in _C._ACCEL_CACHE and rays is not None
    cells = [_C._ACCEL_CACHE[key(x)][6].data_ptr() for x in (False, True)]
    b.backward(torch.ones_like(b))
    assert [_C._ACCEL_CACHE[key(x)][6].data_ptr() for x in (False, True)] == cells      # nothing rebuilt by the backward
    np.testing.assert_array_equal(a.cpu().numpy(), b.detach().cpu().numpy())
    # the two grids hold the same cells, permuted
    lin, bri = (_C._ACCEL_CACHE[key(x)][6].view(-1) for x in (False, True))
    g = _C._ACCEL_CACHE[key(True)][5]
    n = 1 << (3 * g)
    x, y, z = torch.meshgrid(*(torch.arange(1 << g, device=gpu),) * 3, indexing="ij")
    gb = g - 2
    ci = (((((x >> 2) << gb) + (y >> 2)) << gb) + (z >> 2) << 6) | ((x & 3) << 4) | ((y & 3) << 2) | (z & 3)
    assert torch.equal(bri[:n][ci.reshape(-1)], lin[:n]) and torch.equal(bri[n:], lin[n:])
    # rows of 8 / 16 / 32 floats: their forward is march + shade as two kernels with or without a backward behind it -- bricks
    # either way; the weight-accumulating forward is the one-kernel forward -- row-major
    c8 = Case(depth=5, K=8, data_format="RGBA", width=64, height=64)
    t8 = c8.tree(gpu)
    r8 = svox.VolumeRenderer(t8)
    with torch.no_grad():
        r8(t8.features, c8.rays_gpu(gpu))
    assert (id(t8.child), True) in _C._ACCEL_CACHE and (id(t8.child), False) not in _C._ACCEL_CACHE
    f8 = t8.features.detach().clone().requires_grad_(True)
    o8 = r8(f8, c8.rays_gpu(gpu), image_shape=(64, 64))
    o8.backward(torch.ones_like(o8))
    assert (id(t8.child), False) not in _C._ACCEL_CACHE
    with torch.no_grad(), t8.accumulate_weights():
        r8(t8.features, c8.rays_gpu(gpu))
    assert (id(t8.child), False) in _C._ACCEL_CACHE


def test_a_level_finer_for_marching_wavefronts_past_the_cache_rule():
    """Resolution: the cache-fit rule of the one-kernel forward, plus one level (at most 8) where the march is wavefronts of
    its own (depth 9 / 578 MB of features: 7 -> 8)."""
    from svox_t_amd.csrc import _marshal as M
    assert M._accel_log2_for(792753, 2, 578 << 20) == 7 and M._accel_log2_for(792753, 2, 578 << 20, marching=True) == 8
    assert M._accel_log2_for(95000, 2, 71 << 20) == 8 == M._accel_log2_for(95000, 2, 71 << 20, marching=True)      # already finer: fits
    assert M._accel_log2_for(40, 2, 1 << 20, marching=True) == 0 and M._accel_log2_for(10 ** 6, 3, 1 << 20, marching=True) == 0


def test_library_refuses_bricks_too_coarse_for_them(gpu):
    c = Case(depth=5, K=4, data_format="RGBA", width=64, height=64)
    tree = c.tree(gpu)
    ct = _C._pack_tree(tree._spec(tree.features))
    buf = torch.empty((_C._lib.svoxt_accel_bytes(1 | 0x100, ct.n_internal) // 4,), dtype=torch.int32, device=gpu)
    assert _C._lib.svoxt_accel_bytes(1 | 0x100, ct.n_internal) == _C._lib.svoxt_accel_bytes(1, ct.n_internal)
    rc = _C._lib.svoxt_accel_build(ctypes.byref(ct), 1 | 0x100, ctypes.c_void_p(buf.data_ptr()), None)
    assert rc != 0 and b"SVOXT_ACCEL_BRICKS" in _C._lib.svoxt_last_error()
    ct.accel, ct.accel_log2 = buf.data_ptr(), 1 | 0x100
    out = torch.empty((c.Q, 4), device=gpu)
    rs = _C._pack_rays(__import__("svox_t_amd.renderer", fromlist=["_rays_spec_from_rays"])._rays_spec_from_rays(c.rays_gpu(gpu), None))
    rc = _C._lib.svoxt_volume_render_fwd(ctypes.byref(ct), ctypes.byref(rs), ctypes.byref(_C._pack_opts(svox.VolumeRenderer(tree)._get_options(False))),
                                         ctypes.c_void_p(out.data_ptr()), None)
    assert rc != 0 and b"accel_log2" in _C._lib.svoxt_last_error()
