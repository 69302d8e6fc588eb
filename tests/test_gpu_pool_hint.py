"""The pooled sample lists learn their size from the forward before them (svox_t_amd/csrc/__init__.py, _POOL_HINT).
A no-grad forward between training steps must not teach the recording forwards ITS size: with one shared hint the
16 training steps after a validation render ran through the tail kernels at 2.3x their time (r04)."""
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd import synth
from tests.util import Case

pytestmark = pytest.mark.gpu


def test_pool_hint_is_kept_per_kind_of_forward(gpu):
    c = Case(depth=6, K=13, data_format="SH4", width=128, height=96)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    rays = c.rays_gpu(gpu)
    g = synth.grad_output(c.Q, 4).to(gpu)
    f = tree.features.detach().clone().requires_grad_(True)
    _C._POOL_HINT.clear()

    def step():
        f.grad = None
        out = r(f, rays, image_shape=(96, 128))
        out.backward(g)
        torch.cuda.synchronize()
        return out.detach()

    want = step()
    for _ in range(4):
        step()
    rec = [k for k in _C._POOL_HINT if k[2] == "record"]
    assert len(rec) == 1
    blocks = _C._POOL_HINT[rec[0]][0]
    grad = f.grad.clone()
    with torch.no_grad():
        for _ in range(4):
            assert torch.equal(r(f, rays, image_shape=(96, 128)), want)
            torch.cuda.synchronize()
    assert _C._POOL_HINT[rec[0]][0] == blocks, "a no-grad forward changed the recording forwards' pool size"
    assert torch.equal(step(), want)
    assert (f.grad - grad).abs().max().item() <= 1e-6 * grad.abs().max().item()
    assert not _C._POOL_HINT[rec[0]][3], "the recording forward's pool ran dry"
