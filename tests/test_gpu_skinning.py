"""GPU tests of warp_vertices / blend_transformation_matrix (csrc/svoxt_motion.hip)
and of [M, 4, 4] transformation_matrices in the renderer, against the CPU oracle."""
import numpy as np
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from oracle import oracle as O
from oracle import skinning as S
from tests.test_skinning_oracle import case
from tests.util import Case, assert_grads_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Q,J,B", [(1, 1, 1), (777, 6, 3), (20000, 24, 4), (3000, 5000, 2)])
def test_warp_vertices_forward_and_backward(gpu, Q, J, B):
    mats, p, sw, ji = case(Q=Q, J=J, B=B, seed=Q)
    t = [torch.from_numpy(a).to(gpu) for a in (mats, p, sw, ji.astype(np.int32))]
    t[0].requires_grad_(True)
    t[1].requires_grad_(True)
    t[2].requires_grad_(True)
    v, m = svox.warp_vertices(*t)
    wv, wm = S.warp_vertices(mats, p, sw, ji)
    np.testing.assert_array_equal(v.detach().cpu().numpy(), wv)
    np.testing.assert_array_equal(m.detach().cpu().numpy(), wm)
    g = torch.Generator().manual_seed(1)
    gv = torch.randn(Q, 3, generator=g)
    gm = torch.randn(Q, 4, 4, generator=g)
    torch.autograd.backward([v, m], [gv.to(gpu), gm.to(gpu)])
    gp, gmat, gabs, gsw = S.warp_vertices_backward(mats, p, sw, ji, gv.numpy(), gm.numpy())
    np.testing.assert_array_equal(t[1].grad.cpu().numpy(), gp)               # per point: bit-exact
    np.testing.assert_array_equal(t[2].grad.cpu().numpy(), gsw)
    assert_grads_close(t[0].grad.cpu().numpy(), gmat, gabs)                  # summed over points: atomic order
    # the operator entry points return lists, as the pybind11 module does
    out = _C.warp_vertices(*(x.detach() for x in t))
    assert isinstance(out, list) and torch.equal(out[0], v.detach()) and torch.equal(out[1], m.detach())


def test_no_gradient_without_matrix_gradient(gpu):
    """svox.py:68-76: the backward is skipped unless the joint matrices need a gradient."""
    mats, p, sw, ji = case(Q=50)
    pt = torch.from_numpy(p).to(gpu).requires_grad_(True)
    v, _ = svox.warp_vertices(torch.from_numpy(mats).to(gpu), pt, torch.from_numpy(sw).to(gpu),
                              torch.from_numpy(ji.astype(np.int32)).to(gpu))
    v.sum().backward()
    assert pt.grad is None


def test_blended_matrices_drive_the_view_rotation(gpu):
    """blend_transformation_matrix -> [M, 4, 4] -> VolumeRenderer.forward(transformation_matrices=...):
    the upper-left 3x3 rotates the view direction per leaf (rt_kernel.cu:283-291)."""
    c = Case(depth=5, K=28, data_format="SH9", width=40, height=40)
    tree = c.tree(gpu)
    M = c.st.n_features
    rng = np.random.default_rng(4)
    J, B = 5, 3
    joints = np.zeros((J, 4, 4), np.float32)
    for k in range(J):                                      # rigid joint transforms
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        joints[k, :3, :3] = q
        joints[k, :3, 3] = rng.normal(size=3)
        joints[k, 3, 3] = 1
    sw = rng.random((M, B)).astype(np.float32)
    ji = rng.integers(0, J, size=(M, B)).astype(np.int32)
    mats = svox.blend_transformation_matrix(torch.from_numpy(joints).to(gpu), torch.from_numpy(sw).to(gpu),
                                            torch.from_numpy(ji).to(gpu))
    assert mats.shape == (M, 4, 4)
    _, want_m = S.warp_vertices(joints, np.zeros((M, 3), np.float32), sw, ji)
    np.testing.assert_array_equal(mats.cpu().numpy(), want_m)
    r = svox.VolumeRenderer(tree)
    feats = tree.features.detach().clone().requires_grad_(True)
    out = r(feats, c.rays_gpu(gpu), transformation_matrices=mats)
    with O.transformation_matrices(want_m):
        want = O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts())
        gout = torch.randn(c.Q, 4, generator=torch.Generator().manual_seed(2))
        wg, wabs = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), gout.numpy(), want_abs=True)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
    out.backward(gout.to(gpu))
    assert_grads_close(feats.grad.cpu().numpy(), wg, wabs)
    # the 3x3 block alone gives the same picture
    out3 = r(feats, c.rays_gpu(gpu), transformation_matrices=mats[:, :3, :3].contiguous())
    assert torch.equal(out3, out)


def test_warp_argument_errors(gpu):
    mats, p, sw, ji = case(Q=10)
    a = [torch.from_numpy(x).to(gpu) for x in (mats, p, sw, ji.astype(np.int32))]
    with pytest.raises(RuntimeError, match="4, 4"):
        svox.warp_vertices(a[0][:, :3, :3].contiguous(), a[1], a[2], a[3])
    with pytest.raises(RuntimeError, match="int32"):
        svox.warp_vertices(a[0], a[1], a[2], a[3].long())
    with pytest.raises(RuntimeError, match="CUDA"):
        svox.warp_vertices(a[0].cpu(), a[1], a[2], a[3])
    bad = a[3].clone()
    bad[:, 0] = 99                                          # out-of-range joints are skipped
    v, m = svox.warp_vertices(a[0], a[1], a[2], bad)
    assert torch.isfinite(v).all()
