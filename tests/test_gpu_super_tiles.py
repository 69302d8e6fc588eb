"""The super-tile walk of image batches (NOTEBOOK.md step 70; svoxt_set_super_tile_rows): which launch tile renders
which 8 x 8 pixels changes, nothing a caller sees may.  The library switches it on for trees of more than 2^21 feature rows
only -- config 4's full-size tests run under it -- so here the threshold is set to 0 and small images whose tile grid
is ragged against the 8 x 8 tiles of a super-tile (25 x 9, 8 x 8, 3 x 17 tiles) go through every image route."""
import numpy as np
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from oracle import oracle as O
from svox_t_amd import synth
from tests.util import Case, assert_grads_close

pytestmark = pytest.mark.gpu


@pytest.fixture
def super_tiles_everywhere():
    before = _C._lib.svoxt_set_super_tile_rows(0)
    yield
    _C._lib.svoxt_set_super_tile_rows(before)


@pytest.mark.parametrize("fmt,K,depth,W,H", [("SH9", 28, 6, 200, 72), ("RGBA", 4, 5, 64, 64), ("SH4", 13, 6, 24, 136),
                                             ("RGBA", 32, 5, 200, 72), ("RGBA", 8, 5, 72, 200)])
def test_super_tile_walk_changes_nothing(gpu, super_tiles_everywhere, fmt, K, depth, W, H):
    c = Case(depth=depth, K=K, data_format=fmt, width=W, height=H)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    rays = c.rays_gpu(gpu)
    want = O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts())
    Cout = want.shape[1]
    g = synth.grad_output(c.Q, Cout)
    gw, ab = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs=True)
    # the image routes: declared image (ray tensors), no-grad forward, depth and opacity
    out = r(tree.features, rays, image_shape=(H, W))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
    out.backward(g.to(gpu))
    assert_grads_close(tree.features.grad.cpu().numpy(), gw, ab)
    assert "render_bwd_kernel (marches" not in _C.LAST_ROUTE["backward"], _C.LAST_ROUTE      # (the lists were walked)
    with torch.no_grad():
        np.testing.assert_array_equal(r(tree.features, rays, image_shape=(H, W)).cpu().numpy(), want)
        np.testing.assert_array_equal(r.render_depth(tree.features, rays, image_shape=(H, W)).cpu().numpy(),
                                      O.render_depth(c.oracle_tree(), *c.rays_np(), c.oracle_opts()))
        np.testing.assert_array_equal(r.opacity_render(tree.features, rays, image_shape=(H, W)).cpu().numpy(),
                                      O.opacity_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts()))
    # the setter reports what was in force
    assert _C._lib.svoxt_set_super_tile_rows(0) == 0
