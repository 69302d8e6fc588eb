"""Randomised parity stress: irregular trees (random selective refinement),
adversarial features (zero / negative / huge sigma, large coefficients) and
rays (origins inside the volume, axis-parallel directions, grazing and missing
rays, non-unit direction lengths, anisotropic world scaling).  Everything except
the gradient sum order must match the oracle bit for bit."""
import os

import numpy as np
import pytest
import torch

import svox_t_amd as svox
from oracle import oracle as O
from svox_t_amd import synth
from tests.util import assert_grads_close

pytestmark = pytest.mark.gpu


def random_tree(seed, N=2, max_depth=6, data_format="SH4", K=13, p_refine=0.45, p_occ=0.6,
                radius=(0.7, 1.3, 0.9), center=(0.2, -0.1, 0.4)):
    g = torch.Generator().manual_seed(seed)
    t = svox.N3Tree(N=N, data_dim=K, init_reserve=4, depth_limit=max_depth, data_format=data_format,
                    radius=list(radius), center=list(center))
    for _ in range(max_depth):
        leaves = t._all_leaves()
        depth_ok = t.parent_depth[leaves[:, 0], 1] < max_depth - 1
        pick = (torch.rand(len(leaves), generator=g) < p_refine) & depth_ok
        if pick.any():
            sel = leaves[pick]
            t.refine(1, sel=tuple(sel.T), leaf_node=sel)
    leaves = t._all_leaves()
    occ = torch.rand(len(leaves), generator=g) < p_occ
    M = int(occ.sum())
    idx = torch.full((len(leaves),), synth.EMPTY_SENTINEL, dtype=torch.int32)
    idx[occ] = torch.randperm(M, generator=g).to(torch.int32)          # not in leaf order
    t.data[tuple(leaves.T)] = idx[:, None]
    feats = torch.randn(M, K, generator=g) * 3.0
    sig = torch.pow(10.0, torch.rand(M, generator=g) * 6.0 - 2.0)      # 1e-2 .. 1e4
    kind = torch.rand(M, generator=g)
    sig[kind < 0.1] = 0.0
    sig[(kind >= 0.1) & (kind < 0.2)] = -5.0
    feats[:, K - 1] = sig
    feats[torch.rand(M, generator=g) < 0.05, :K - 1] = 40.0            # saturated sigmoids
    return t, feats


def random_rays(seed, Q, t):
    g = torch.Generator().manual_seed(seed)
    lo = (t.tree2world(torch.zeros(1, 3)))[0]
    hi = (t.tree2world(torch.ones(1, 3)))[0]
    ext = hi - lo
    o = lo + (torch.rand(Q, 3, generator=g) * 2.0 - 0.5) * ext           # inside and outside the cube
    target = lo + torch.rand(Q, 3, generator=g) * ext
    d = target - o
    d = d * (0.2 + 3.0 * torch.rand(Q, 1, generator=g))                  # not unit length
    # axis-parallel and plane-parallel directions (zero components)
    n = Q // 8
    d[:n] = torch.eye(3)[torch.randint(0, 3, (n,), generator=g)] * torch.where(torch.rand(n, 1, generator=g) < 0.5, -1.0, 1.0)
    d[n:2 * n, torch.randint(0, 3, (1,), generator=g).item()] = 0.0
    # some rays that miss: point away from the cube
    d[2 * n:2 * n + n // 2] = (o[2 * n:2 * n + n // 2] - (lo + 0.5 * ext)) + 1e-3
    v = torch.nn.functional.normalize(torch.randn(Q, 3, generator=g), dim=-1)
    return o.contiguous(), d.contiguous(), v.contiguous()


# SVOXT_STRESS_SEEDS=n: a longer sweep (seeds 0..n-1) than the default four
SEEDS = list(range(int(os.environ.get("SVOXT_STRESS_SEEDS", "4"))))


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("N,fmt,K", [(2, "SH4", 13), (2, "RGBA", 4), (3, "SH1", 4)])
def test_random_tree_and_rays(gpu, seed, N, fmt, K, monkeypatch):
    import svox_t_amd.csrc as _C
    # odd seeds force the two-kernel backward onto these incoherent rays (nearly every record
    # of a tile a different feature row: the merge table runs full), even seeds take the default
    monkeypatch.setattr(_C, "BWD_GATHER", 2 if seed % 2 else 1)
    t, feats = random_tree(seed, N=N, max_depth=6 if N == 2 else 3, data_format=fmt, K=K)
    n = t.n_internal
    o, d, v = random_rays(100 + seed, 6000, t)
    data_np, child_np = t.data[:n].numpy().copy(), t.child[:n].numpy().copy()
    ot = O.Tree(feats.numpy(), data_np, child_np,
                offset=t.offset.numpy().copy(), scaling=t.invradius.numpy().copy())
    df = svox.DataFormat(fmt)
    tg = t.to(gpu)          # nn.Module.to moves the buffers in place: host copies were taken above
    r = svox.VolumeRenderer(tg, step_size=2e-3, background_brightness=0.5)
    rays = svox.Rays(o.to(gpu), d.to(gpu), v.to(gpu))
    rnp = (o.numpy(), d.numpy(), v.numpy())
    for fast in (False, True):
        th = 1e-2 if fast else 0.0
        opt = O.make_options(step_size=2e-3, background_brightness=0.5, format=df.format,
                             basis_dim=df.basis_dim, sigma_thresh=th, stop_thresh=th)
        f = feats.to(gpu).requires_grad_(True)
        out = r(f, rays, fast=fast)
        want, cnt = O.volume_render(ot, *rnp, opt, count=True)
        np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
        assert cnt.active > 0 and cnt.rays_hit < len(o)
        with torch.no_grad():
            np.testing.assert_array_equal(r.render_depth(f, rays, fast=fast).cpu().numpy(), O.render_depth(ot, *rnp, opt))
            np.testing.assert_array_equal(r.opacity_render(f, rays, fast=fast).cpu().numpy(), O.opacity_render(ot, *rnp, opt))
        gout = synth.grad_output(len(o), out.shape[1], seed=seed)
        out.backward(gout.to(gpu))
        # the backward ignores both thresholds (rt_kernel.cu:382,456): same oracle call either way
        opt0 = O.make_options(step_size=2e-3, background_brightness=0.5, format=df.format, basis_dim=df.basis_dim)
        gw, ab = O.volume_render_backward(ot, *rnp, opt0, gout.numpy(), want_abs=True)
        assert_grads_close(f.grad.cpu().numpy(), gw, ab)
        # the same batch in svoxt_ray_order's order: every row unchanged, the same gradient sum
        f2 = feats.to(gpu).requires_grad_(True)
        out2 = r(f2, rays, fast=fast, sort_rays=True)
        np.testing.assert_array_equal(out2.detach().cpu().numpy(), want)
        out2.backward(gout.to(gpu))
        assert_grads_close(f2.grad.cpu().numpy(), gw, ab)
        # opacity backward (from sample lists when the thresholds are 0, marching otherwise)
        f3 = feats.to(gpu).requires_grad_(True)
        g1 = synth.grad_output(len(o), 1, seed=seed + 50)
        r.opacity_render(f3, rays, fast=fast, sort_rays=bool(seed % 2)).backward(g1.to(gpu))
        gw1, ab1 = O.volume_render_backward(ot, *rnp, opt0, g1.numpy(), want_abs=True)
        assert_grads_close(f3.grad.cpu().numpy(), gw1, ab1)
    # point query on the same irregular tree
    pts = torch.rand(4000, 3, generator=torch.Generator().manual_seed(seed)) * 1.2 - 0.1
    vals, nid, did = tg(feats.to(gpu), pts.to(gpu), want_node_ids=True, want_data_ids=True, world=False)
    wv, wn, wd = O.query(O.Tree(feats.numpy(), data_np, child_np), pts.numpy())
    np.testing.assert_array_equal(vals.cpu().numpy(), wv)
    np.testing.assert_array_equal(nid.cpu().numpy(), wn)
    np.testing.assert_array_equal(did.cpu().numpy(), wd)


@pytest.mark.parametrize("seed", [0, 1])
def test_random_tree_view_rotations_and_motion(gpu, seed, monkeypatch):
    """The same adversarial trees and rays through the view-rotation kernels (arbitrary,
    not even orthonormal, 4x4 matrices; with and without thresholds) and the motion
    variants."""
    import svox_t_amd.csrc as _C
    monkeypatch.setattr(_C, "BWD_GATHER", 2 if seed % 2 else 1)     # seed 1: two-kernel backward with rotations
    t, feats = random_tree(seed, N=2, max_depth=6, data_format="SH4", K=13)
    n = t.n_internal
    M = feats.shape[0]
    o, d, v = random_rays(300 + seed, 5000, t)
    g = torch.Generator().manual_seed(seed)
    joints = torch.randn(9, 5, generator=g)
    data_np, child_np = t.data[:n].numpy().copy(), t.child[:n].numpy().copy()
    off, scl = t.offset.numpy().copy(), t.invradius.numpy().copy()
    ot = O.Tree(feats.numpy(), data_np, child_np, offset=off, scaling=scl, extra=joints.numpy())
    xf = torch.randn(M, 4, 4, generator=g)
    tg = svox.N3Tree.from_arrays(child_np, data_np, t.parent_depth[:n].numpy(), feats, data_format="SH4",
                                 radius=[0.7, 1.3, 0.9], center=[0.2, -0.1, 0.4], extra_data=joints, device=gpu)
    np.testing.assert_array_equal(tg.offset.cpu().numpy(), off)
    r = svox.VolumeRenderer(tg, step_size=2e-3, background_brightness=0.5)
    rays = svox.Rays(o.to(gpu), d.to(gpu), v.to(gpu))
    rnp = (o.numpy(), d.numpy(), v.numpy())
    for fast in (False, True):
        th = 1e-2 if fast else 0.0
        opt = O.make_options(step_size=2e-3, background_brightness=0.5, format=O.FORMAT_SH, basis_dim=4,
                             sigma_thresh=th, stop_thresh=th)
        opt0 = O.make_options(step_size=2e-3, background_brightness=0.5, format=O.FORMAT_SH, basis_dim=4)
        f = feats.to(gpu).requires_grad_(True)
        out = r(f, rays, transformation_matrices=xf.to(gpu), fast=fast)
        gout = synth.grad_output(len(o), 4, seed=seed)
        out.backward(gout.to(gpu))
        with O.transformation_matrices(xf.numpy()):
            want = O.volume_render(ot, *rnp, opt)
            gw, ab = O.volume_render_backward(ot, *rnp, opt0, gout.numpy(), want_abs=True)
        np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
        assert_grads_close(f.grad.cpu().numpy(), gw, ab)
        # motion variants
        got = r.motion_render(feats.to(gpu), rays, fast=fast)
        for a, b in zip(got, O.motion_render(ot, *rnp, opt)):
            np.testing.assert_array_equal(a.cpu().numpy(), b)
        jf = torch.randn(9, 7, generator=g)
        sw = torch.rand(M, 3, generator=g)
        sw[torch.rand(M, 3, generator=g) < 0.3] = 0.0
        ji = torch.randint(0, 9, (M, 3), generator=g, dtype=torch.int32)
        jft = jf.to(gpu).requires_grad_(True)
        mf = r.motion_feature_render(feats.to(gpu), jft, sw.to(gpu), ji.to(gpu), rays, fast=fast)
        mo = O.Motion(jf.numpy(), sw.numpy(), ji.numpy())
        np.testing.assert_array_equal(mf.detach().cpu().numpy(), O.motion_feature_render(ot, mo, *rnp, opt))
        gm = synth.grad_output(len(o), 7, seed=seed + 5)
        mf.backward(gm.to(gpu))
        wg, wabs = O.motion_feature_render_backward(ot, mo, *rnp, opt0, gm.numpy(), want_abs=True)
        assert_grads_close(jft.grad.cpu().numpy(), wg, wabs)
