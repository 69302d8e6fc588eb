"""Maximum sizes: a 4096 x 4096 image (16.8 M rays, 26 x the headline batch; 262 144 tiles, a 12.9 GB pool of list blocks --
record indices past 2^31) through the default route, declared and undeclared, against the oracle at full size: forward
bit for bit, gradient within 1e-5 of the tight scale."""
import numpy as np
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from oracle import oracle as O
from svox_t_amd import synth
from tests.util import Case, assert_grads_close

pytestmark = pytest.mark.gpu


def test_sixteen_million_rays(gpu):
    W = H = 4096
    c = Case(depth=8, K=28, data_format="SH9", width=W, height=H)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    rays = c.rays_gpu(gpu)
    f = tree.features.detach().clone().requires_grad_(True)
    g = synth.grad_output(c.Q, 4, seed=3)
    want = O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts())
    gw, _, tight = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs="both")
    for kw in (dict(image_shape=(H, W)), {}):
        f.grad = None
        out = r(f, rays, **kw)
        out.backward(g.to(gpu))
        assert "fwd_roles_kernel" in _C.LAST_ROUTE["forward"] and "grad_fused_kernel" in _C.LAST_ROUTE["backward"], _C.LAST_ROUTE
        np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
        assert_grads_close(f.grad.cpu().numpy(), gw, tight)
        del out
    print(f"\n[4096 x 4096] peak device memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
