"""GPU tests of the point query, generic-N / SG / component-range paths, edge
cases, and the full-size BASELINE workloads.  Through the Python surface ->
ctypes -> C ABI."""
import numpy as np
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from oracle import oracle as O
from svox_t_amd import synth
from svox_t_amd.renderer import _rays_spec_from_rays
from tests.util import Case, assert_grads_close, assert_outputs_close

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ query
@pytest.mark.parametrize("world", [True, False])
def test_query_vertical_matches_oracle(gpu, world):
    c = Case(depth=5, K=13, data_format="SH4", width=8, height=8,
             radius=[1.0, 1.2, 0.8], center=[0.1, -0.2, 0.3])
    tree = c.tree(gpu)
    g = torch.Generator().manual_seed(11)
    pts = torch.rand(5000, 3, generator=g)
    pts[:200] = torch.rand(200, 3, generator=g) * 3 - 1          # some outside the cube: clamped
    if world:
        pts = tree.tree2world(pts.to(gpu)).cpu()
    vals, node_ids, data_ids, leaf_node = tree(tree.features, pts.to(gpu), want_node_ids=True,
                                               world=world, want_data_ids=True, want_leaf_node=True)
    ot = c.oracle_tree()
    if not world:
        ot = O.Tree(ot.features, ot.data, ot.child)
    wv, wn, wd = O.query(ot, pts.numpy())
    np.testing.assert_array_equal(vals.detach().cpu().numpy(), wv)
    np.testing.assert_array_equal(node_ids.cpu().numpy(), wn)
    np.testing.assert_array_equal(data_ids.cpu().numpy(), wd)
    uniq = np.unique(wn)
    N = 2
    want_leaf = np.stack([uniq // 8, (uniq // 4) % 2, (uniq // 2) % 2, uniq % 2], -1)
    np.testing.assert_array_equal(leaf_node.cpu().numpy(), want_leaf)
    # backward: scatter-add of the upstream rows
    gout = torch.randn(vals.shape, generator=g)
    vals.backward(gout.to(gpu))
    want = O.query_backward(ot, pts.numpy(), gout.numpy())
    absum = O.query_backward(ot, pts.numpy(), gout.abs().numpy())
    assert_grads_close(tree.features.grad.cpu().numpy(), want, absum)


def test_point_keyed_view_refine_and_construct_tree(gpu):
    """tree[points].refine() then construct_tree(points): the dynamic set-up
    sequence of SURVEY.md 3.4, on the GPU."""
    tree = svox.N3Tree(N=2, data_dim=4, init_reserve=64, map_location=gpu)
    g = torch.Generator().manual_seed(2)
    pts = (torch.rand(300, 3, generator=g) * 0.2 + 0.4).to(gpu)     # a blob in the middle
    for _ in range(3):
        tree[svox.LocalIndex(pts)].refine()
    assert tree.max_depth == 3
    # every point now sits in a depth-3 leaf
    view = tree[svox.LocalIndex(pts)]
    assert int(view.depths.min()) == 3
    tree.construct_tree(tree.tree2world(pts))
    feats = torch.randn(300, 4, device=gpu)
    vals, _, data_ids = tree(feats, tree.tree2world(pts), want_node_ids=True, want_data_ids=True)
    # several points may share a leaf: the leaf keeps one of them; that point reads itself back
    ids = data_ids.cpu().numpy()
    assert ids.min() >= 0 and ids.max() < 300
    np.testing.assert_array_equal(vals.cpu().numpy(), feats.cpu().numpy()[ids])


# ------------------------------------------------- generic paths and edge cases
def _random_full_tree(N, levels, K, seed=0):
    t = svox.N3Tree(N=N, data_dim=K, init_reserve=8)
    for _ in range(levels):
        t.refine(1)
    leaves = t._all_leaves()
    g = torch.Generator().manual_seed(seed)
    occupied = torch.rand(len(leaves), generator=g) < 0.35
    M = int(occupied.sum())
    idx = torch.full((len(leaves),), synth.EMPTY_SENTINEL, dtype=torch.int32)
    idx[occupied] = torch.arange(M, dtype=torch.int32)
    t.data[tuple(leaves.T)] = idx[:, None]
    feats = synth.shell_features(M, K, seed=seed)
    return t, feats


@pytest.mark.parametrize("N,levels", [(3, 2), (4, 2)])
def test_branching_factor_other_than_two(gpu, N, levels):
    t, feats = _random_full_tree(N, levels, 4)
    ot = O.Tree(feats.numpy(), t.data[:t.n_internal].numpy(), t.child[:t.n_internal].numpy())
    o, d, v = synth.pinhole_rays(48, 48)
    opt = O.make_options()
    tg = t.to(gpu)
    r = svox.VolumeRenderer(tg)
    f = feats.to(gpu).requires_grad_(True)
    out = r(f, svox.Rays(o.to(gpu), d.to(gpu), v.to(gpu)))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.volume_render(ot, o.numpy(), d.numpy(), v.numpy(), opt))
    gout = synth.grad_output(o.shape[0], 4)
    out.backward(gout.to(gpu))
    want, absum = O.volume_render_backward(ot, o.numpy(), d.numpy(), v.numpy(), opt, gout.numpy(), want_abs=True)
    assert_grads_close(f.grad.cpu().numpy(), want, absum)


def test_component_subrange_and_sg_format(gpu):
    c = Case(depth=4, K=28, data_format="SH9", width=40, height=40)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree, min_comp=1, max_comp=5)
    with torch.no_grad():
        got = r(tree.features, c.rays_gpu(gpu)).cpu().numpy()
    want = O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts(min_comp=1, max_comp=5))
    np.testing.assert_array_equal(got, want)
    # spherical gaussians: basis from extra_data rows (lambda, mu_xyz)
    g = torch.Generator().manual_seed(4)
    lobes = torch.cat([torch.rand(6, 1, generator=g) * 4 + 0.5,
                       torch.nn.functional.normalize(torch.randn(6, 3, generator=g), dim=-1)], -1)
    cs = Case(depth=4, K=19, data_format="SG6", width=40, height=40)
    t = svox.N3Tree.from_arrays(cs.st.child, cs.st.data, cs.st.parent_depth, cs.features,
                                data_format="SG6", extra_data=lobes, device=gpu)
    rs = svox.VolumeRenderer(t)
    f = t.features
    out = rs(f, cs.rays_gpu(gpu))
    ot = O.Tree(cs.features.numpy(), cs.st.data, cs.st.child, extra=lobes.numpy())
    opt = O.make_options(format=O.FORMAT_SG, basis_dim=6)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.volume_render(ot, *cs.rays_np(), opt))
    gout = synth.grad_output(cs.Q, 4)
    out.backward(gout.to(gpu))
    want, absum = O.volume_render_backward(ot, *cs.rays_np(), opt, gout.numpy(), want_abs=True)
    assert_grads_close(f.grad.cpu().numpy(), want, absum)


def test_asg_format(gpu):
    """Anisotropic spherical gaussians (maybe_precalc_basis FORMAT_ASG, rt_kernel.cu:118-130, marked
    untested there): basis_i = S * exp(-lambda dot_x^2 - mu dot_y^2) / basis_dim from extra_data rows
    [lambda, mu, x(3), y(3), z(3)] -- forward bit-exact vs the oracle, gradient within tolerance,
    and the basis values themselves against a float64 evaluation of the formula."""
    g = torch.Generator().manual_seed(5)
    B = 5
    frames = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0]                 # orthonormal lobe frames
    lobes = torch.cat([torch.rand(B, 2, generator=g) * 3 + 0.3, frames.reshape(B, 9)], -1).contiguous()
    cs = Case(depth=4, K=3 * B + 1, data_format=f"ASG{B}", width=40, height=40)
    assert (cs.format, cs.basis_dim) == (O.FORMAT_ASG, B)
    t = svox.N3Tree.from_arrays(cs.st.child, cs.st.data, cs.st.parent_depth, cs.features,
                                data_format=f"ASG{B}", extra_data=lobes, device=gpu)
    rs = svox.VolumeRenderer(t)
    f = t.features
    out = rs(f, cs.rays_gpu(gpu))
    ot = O.Tree(cs.features.numpy(), cs.st.data, cs.st.child, extra=lobes.numpy())
    opt = O.make_options(format=O.FORMAT_ASG, basis_dim=B)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.volume_render(ot, *cs.rays_np(), opt))
    gout = synth.grad_output(cs.Q, 4)
    out.backward(gout.to(gpu))
    want, absum = O.volume_render_backward(ot, *cs.rays_np(), opt, gout.numpy(), want_abs=True)
    assert_grads_close(f.grad.cpu().numpy(), want, absum)
    assert np.abs(want[:, :-1]).max() > 0
    # the oracle's basis against the formula in float64
    d = cs.vdirs.numpy().astype(np.float64)[:200]
    L = lobes.numpy().astype(np.float64)
    ref = (d @ L[:, 8:11].T) * np.exp(-L[:, 0] * (d @ L[:, 2:5].T) ** 2 - L[:, 1] * (d @ L[:, 5:8].T) ** 2) / B
    np.testing.assert_allclose(O.basis(O.FORMAT_ASG, B, cs.vdirs.numpy()[:200], extra=lobes.numpy()), ref, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("kind,B", [("SG", 9), ("ASG", 4), ("SG", 25), ("ASG", 16), ("SG", 1)])
def test_sg_asg_with_sh_sized_lobe_counts_take_the_register_kernels(gpu, kind, B, monkeypatch):
    """SG / ASG payloads with 1 / 4 / 9 / 16 / 25 lobes and three channels are served by the kernels that keep a ray's
    basis values in registers (render_fwd_kernel / render_bwd_kernel<..., LOBES>, r03; other lobe counts stay with the
    generic kernels: the two tests above): forward bit-exact against the oracle, gradient within its tolerance,
    and a gradient that is not trivially zero."""
    g = torch.Generator().manual_seed(40 + B)
    if kind == "SG":
        lobes = torch.cat([torch.rand(B, 1, generator=g) * 4 + 0.5,
                           torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)], -1).contiguous()
        fmt = O.FORMAT_SG
    else:
        frames = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0]
        lobes = torch.cat([torch.rand(B, 2, generator=g) * 3 + 0.3, frames.reshape(B, 9)], -1).contiguous()
        fmt = O.FORMAT_ASG
    cs = Case(depth=4, K=3 * B + 1, data_format=f"{kind}{B}", width=40, height=40)
    assert (cs.format, cs.basis_dim) == (fmt, B)
    t = svox.N3Tree.from_arrays(cs.st.child, cs.st.data, cs.st.parent_depth, cs.features,
                                data_format=f"{kind}{B}", extra_data=lobes, device=gpu)
    rs = svox.VolumeRenderer(t)
    f = t.features
    out = rs(f, cs.rays_gpu(gpu))
    ot = O.Tree(cs.features.numpy(), cs.st.data, cs.st.child, extra=lobes.numpy())
    opt = O.make_options(format=fmt, basis_dim=B)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.volume_render(ot, *cs.rays_np(), opt))
    gout = synth.grad_output(cs.Q, 4)
    out.backward(gout.to(gpu))
    assert _C.LAST_ROUTE["backward"].startswith("render_bwd_kernel (marches"), _C.LAST_ROUTE
    want, absum, tight = O.volume_render_backward(ot, *cs.rays_np(), opt, gout.numpy(), want_abs="both")
    assert_grads_close(f.grad.cpu().numpy(), want, absum)
    assert np.abs(want[:, :-1]).max() > 0
    # declared an image: sample lists with the backward's hand-over (march + tile shade), then the per-tile backward
    # over them (grad_fused_kernel<..., LOBES>) -- the exact route: held to the tight scale; also with lists of 8
    # samples, whose overflowing rays go through the tail launches
    for cap in (None, 8):
        if cap is not None:
            monkeypatch.setattr(_C, "BWD_LIST_SAMPLES", cap)
        f.grad = None
        out = rs(f, cs.rays_gpu(gpu), image_shape=(40, 40))
        np.testing.assert_array_equal(out.detach().cpu().numpy(), O.volume_render(ot, *cs.rays_np(), opt))
        out.backward(gout.to(gpu))
        assert ("fwd_roles_kernel" if B <= 16 else "shade_tile_kernel") in _C.LAST_ROUTE["forward"], _C.LAST_ROUTE
        assert _C.LAST_ROUTE["backward"].startswith("grad_fused_kernel<EXACT>"), _C.LAST_ROUTE
        assert_grads_close(f.grad.cpu().numpy(), want, tight)
    # the knobs that take the per-tile backward away take the lists away with it
    monkeypatch.setattr(_C, "BWD_GATHER", 0)
    f.grad = None
    rs(f, cs.rays_gpu(gpu), image_shape=(40, 40)).backward(gout.to(gpu))
    assert _C.LAST_ROUTE["backward"].startswith("render_bwd_kernel (marches"), _C.LAST_ROUTE
    assert_grads_close(f.grad.cpu().numpy(), want, tight)


@pytest.mark.parametrize("Q", [0, 1, 63, 257])
def test_ragged_and_empty_ray_batches(gpu, Q):
    c = Case(depth=4, K=4, data_format="RGBA", width=32, height=32)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    rays = svox.Rays(*(t[:Q].to(gpu) for t in (c.origins, c.dirs, c.vdirs)))
    out = r(tree.features, rays)
    assert out.shape == (Q, 4)
    o, d, v = (a[:Q] for a in c.rays_np())
    if Q:
        np.testing.assert_array_equal(out.detach().cpu().numpy(), O.volume_render(c.oracle_tree(), o, d, v, c.oracle_opts()))
    out.sum().backward()
    want, absum = O.volume_render_backward(c.oracle_tree(), o, d, v, c.oracle_opts(),
                                           np.ones((Q, 4), np.float32), want_abs=True)
    assert_grads_close(tree.features.grad.cpu().numpy(), want, absum)


def test_rejects_noncontiguous_and_wrong_dtype(gpu):
    c = Case(depth=3, K=4, data_format="RGBA", width=8, height=8)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    rays = c.rays_gpu(gpu)
    bad = svox.Rays(rays.origins.t().contiguous().t(), rays.dirs, rays.viewdirs)
    with pytest.raises(RuntimeError, match="contiguous"):
        r(tree.features, bad)
    with pytest.raises(RuntimeError, match="float"):
        r(tree.features, svox.Rays(rays.origins.double(), rays.dirs.double(), rays.viewdirs.double()))
    with pytest.raises(RuntimeError, match="M, 3, 3"):
        r(tree.features, rays, transformation_matrices=torch.eye(3, device=gpu).repeat(5, 1, 1))
    with pytest.raises(RuntimeError, match="M, 4, 4"):
        r(tree.features, rays, transformation_matrices=torch.zeros(tree.features.shape[0], 3, 4, device=gpu))


def test_weight_accumulation(gpu):
    """Sum over leaf slots of the accumulated weights == sum over rays of alpha
    (every composited weight lands in exactly one slot)."""
    c = Case(depth=5, K=4, data_format="RGBA", width=64, height=64)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    with torch.no_grad(), tree.accumulate_weights() as acc:
        out = r(tree.features, c.rays_gpu(gpu))
        w = acc.value.clone()
        per_leaf = acc()
    assert w.shape == tree.child.shape and per_leaf.shape[0] == tree.n_leaves
    total_alpha = out[:, 3].double().sum().item()
    assert w.double().sum().item() == pytest.approx(total_alpha, rel=1e-4)
    assert (w[tree.child != 0] == 0).all()              # internal slots never receive weight
    # occupied slots only
    empty = (tree.data.squeeze(-1) >= tree.features.shape[0])
    assert (w[empty] == 0).all()


@pytest.mark.parametrize("name", ["d5_rgba4", "d5_rgba8", "d6_sh9", "random2", "random3"])
@pytest.mark.parametrize("accel", ["grid", "plain"])
def test_weight_accumulation_matches_oracle(gpu, name, accel, monkeypatch):
    """tree._weight_accum (rt_kernel.cu:266-267, 309-311; svox.py:948-969) per leaf slot against the
    oracle's double-precision sums: every weight is bit-identical, so the only difference is the
    order of the float atomics -- bounded by 1e-6 of the slot's sum (the weights are positive).
    With the acceleration grid (which resolves coarse leaves without their slot id: the kernel
    recovers it by the root descent) and without; on the shell trees, and on random trees whose
    coarse leaves hold data; with and without early termination; N = 2 and 3."""
    import svox_t_amd.csrc as _C
    from svox_t_amd.renderer import _rays_spec_from_rays
    monkeypatch.setattr(_C, "ACCEL_LOG2", 0 if accel == "plain" else 5)
    if name.startswith("random"):
        from tests.test_gpu_random_stress import random_rays, random_tree
        N = int(name[-1])
        t, feats = random_tree(7, N=N, max_depth=6 if N == 2 else 3, data_format="SH4", K=13)
        n = t.n_internal
        o, d, v = random_rays(107, 6000, t)
        fmt, bd = svox.DataFormat("SH4").format, 4
        ot = O.Tree(feats.numpy(), t.data[:n].numpy().copy(), t.child[:n].numpy().copy(),
                    offset=t.offset.numpy().copy(), scaling=t.invradius.numpy().copy())
        tree = t.to(gpu)           # nn.Module.to moves the buffers in place: host copies were taken above
        features = feats.to(gpu)
        rays = svox.Rays(o.to(gpu), d.to(gpu), v.to(gpu))
        rays_np = (o.numpy(), d.numpy(), v.numpy())
    else:
        cfg = dict(d5_rgba4=dict(depth=5, K=4, data_format="RGBA", width=64, height=64),
                   d5_rgba8=dict(depth=5, K=8, data_format="RGBA", width=64, height=64),    # rows with a two-kernel forward: not taken here
                   d6_sh9=dict(depth=6, K=28, data_format="SH9", width=96, height=96))[name]
        c = Case(**cfg)
        tree, ot = c.tree(gpu), c.oracle_tree()
        features = tree.features
        fmt, bd = c.format, c.basis_dim
        rays, rays_np = c.rays_gpu(gpu), c.rays_np()
    r = svox.VolumeRenderer(tree)
    _C._POOL_HINT.clear()
    for fast in (False, True):
        th = 1e-2 if fast else 0.0
        with torch.no_grad(), tree.accumulate_weights() as acc:
            out = r(features, rays, fast=fast)
            got = acc.value.double().cpu().numpy()
        # (r05) the weight-accumulating forward is the one-kernel forward: no scratch lists were made for it, so none
        # was read back as a pool's use
        assert not _C._POOL_HINT, _C._POOL_HINT
        want_out, want = O.volume_render_weights(ot, *rays_np, O.make_options(format=fmt, basis_dim=bd,
                                                                               sigma_thresh=th, stop_thresh=th))
        np.testing.assert_array_equal(out.cpu().numpy(), want_out)
        want = want[:got.shape[0]]
        assert got.shape == want.shape
        assert (got[want == 0] == 0).all()                      # exactly the slots the oracle touches
        assert want.sum() > 0
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=0)


# ------------------------------------------------------------ full-size configs
@pytest.fixture(scope="module")
def cfg3():
    """BASELINE configs[1]/[2]: depth-8 SH9 tree, 800x800 rays."""
    return Case(depth=8, K=28, data_format="SH9", width=800, height=800)


def test_config2_forward_full_size(cfg3, gpu):
    c = cfg3
    assert (c.st.n_internal, c.st.n_features) == (123841, 668912)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    with torch.no_grad():
        got = r(tree.features, c.rays_gpu(gpu)).cpu().numpy()
    want, cnt = O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), count=True)
    assert_outputs_close(got, want)
    np.testing.assert_array_equal(got, want)
    got_cnt = _C.count_forward(tree._spec(tree.features), _rays_spec_from_rays(c.rays_gpu(gpu)),
                               r._get_options()).cpu().tolist()
    assert tuple(got_cnt) == tuple(cnt) == (610128, 18919396, 113601892, 6545320, 5886409)
    # size-independent properties
    assert got[:, 3].min() >= 0 and got[:, 3].max() <= 1
    miss = got[:, 3] == 0
    assert np.all(got[miss, :3] == 1.0)                 # untouched rays show the white background


def test_config3_backward_full_size(cfg3, gpu):
    c = cfg3
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    out = r(tree.features, c.rays_gpu(gpu), image_shape=(800, 800))     # as bench.py runs it
    gout = synth.grad_output(c.Q, 4)
    out.backward(gout.to(gpu))
    got = tree.features.grad.cpu().numpy()
    want, absum = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), gout.numpy(), want_abs=True)
    assert_grads_close(got, want, absum)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts()))
    # linearity in the upstream gradient: backward(2g) == 2 backward(g) up to atomics order
    tree.features.grad = None
    out2 = r(tree.features, c.rays_gpu(gpu))
    out2.backward(2 * gout.to(gpu))
    assert_grads_close(tree.features.grad.cpu().numpy(), 2 * want, 2 * absum)


def test_sh25_backward_full_size(gpu):
    """The headline geometry with rows of 76 floats (SH25): grad_fused_kernel over the forward's hand-over with
    ceil(76 / 16) rounds of columns in its reduce and passes of 896 records -- dense tiles, whose first pair of
    rounds fills 896 slots, are what a first version with 768 overran (r03; only full size has them)."""
    c = Case(depth=8, K=76, data_format="SH25", width=800, height=800)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    out = r(tree.features, c.rays_gpu(gpu), image_shape=(800, 800))
    gout = synth.grad_output(c.Q, 4)
    out.backward(gout.to(gpu))
    assert _C.LAST_ROUTE["backward"].startswith("grad_fused_kernel<EXACT>"), _C.LAST_ROUTE
    got = tree.features.grad.cpu().numpy()
    want, absum, tight = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), gout.numpy(), want_abs="both")
    assert_grads_close(got, want, tight)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts()))


def test_config3_backward_relative_error_both_routes(cfg3, gpu, monkeypatch, capsys):
    """What "1e-5 relative" holds for at full size, route by route.

    (a) the default, exact route (grad_fused_kernel<..., EXACT>: accum added up sequentially like the
        reference's pass 1, rt_kernel.cu:365-437; every contribution bit-identical to the
        reference's formulas): every entry within 1e-5 of the TIGHT scale -- accum priced by
        |w_j * total_color_j| and |T * bg * sum g|, the scale in force before the single-march
        backward existed -- and the plain elementwise criterion |got - want| <= 1e-5 |want| fails
        for at most 0.5 % of the entries, sigma column and colour columns alike (measured r02:
        0.157 % / 0.157 %).  That remainder is float-atomic reordering: an entry that is the small
        sum of many contributions of both signs has |want| below the rounding error of its own
        sum -- true of the reference's atomics too.
    (b) the opt-in single-march route (SVOXT_BWD_EXACT=0: accum = sum_c g_c * out_c from the
        forward's output, one sweep, 0.10 ms faster): within 1e-5 of the sum-of-elementary-
        magnitudes scale, colour columns as (a) -- but 36 % of the sigma entries miss the
        elementwise criterion (r02), which is why it is not the default.  Reported here."""
    c = cfg3
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    gout = synth.grad_output(c.Q, 4)
    want, absum, tight = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), gout.numpy(),
                                                   want_abs="both")
    assert (tight <= absum * (1 + 1e-5) + 1e-300).all()      # |sum| <= sum|.|, up to the float rounding of total_color
    stats = {}
    for route, exact in (("exact", True), ("single-march", False)):
        monkeypatch.setattr(_C, "BWD_EXACT", exact)
        tree.features.grad = None
        out = r(tree.features, c.rays_gpu(gpu), image_shape=(800, 800))
        out.backward(gout.to(gpu))
        got = tree.features.grad.double().cpu().numpy()
        assert_grads_close(got, want, tight if exact else absum, what=route)
        err = np.abs(got - want)
        touched = absum > 0
        sig = np.zeros_like(touched); sig[:, -1] = True
        off = err > 1e-5 * np.abs(want)
        stats[route] = (float(off[touched & sig].mean()), float(off[touched & ~sig].mean()),
                        float((err / (tight + 1e-300))[touched].max()))
    with capsys.disabled():
        for k, (fs, fc, worst) in stats.items():
            print(f"\n[cfg3 backward, {k}] entries with |err| > 1e-5 |want|: sigma column {fs:.4%}, "
                  f"colour columns {fc:.4%}; worst |err| / tight scale {worst:.2e}")
    fs, fc, worst = stats["exact"]
    assert fs <= 0.005 and fc <= 0.005 and worst <= 1e-5
    assert stats["single-march"][1] <= 0.005


class _Cfg4:
    """BASELINE configs[3] (depth-9, 31 features + sigma, 1024 x 1024) and the oracle's results for it,
    computed once per test module."""

    def __init__(self):
        self.case = Case(depth=9, K=32, data_format="RGBA", width=1024, height=1024)
        self._bwd = None

    def backward(self):
        if self._bwd is None:
            c = self.case
            gout = synth.grad_output(c.Q, 32)
            want, absum, tight = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), gout.numpy(),
                                                           want_abs="both")
            self._bwd = (gout, want, absum, tight)
        return self._bwd


@pytest.fixture(scope="module")
def cfg4():
    return _Cfg4()


def test_config4_depth9_features32_and_depth(cfg4, gpu, monkeypatch):
    """BASELINE configs[3]: depth-9, data_dim 32 (31 features + sigma), 1024x1024:
    volume_render [Q,32] and render_depth [Q,1]; checked against the oracle on a
    deterministic subsample of the rays (the full batch is rendered)."""
    c = cfg4.case
    assert (c.st.n_internal, c.st.n_features) == (792753, 4738568)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    rays = c.rays_gpu(gpu)
    with torch.no_grad():
        out = r(tree.features, rays)
        depth = r.render_depth(tree.features, rays)
    assert out.shape == (c.Q, 32) and depth.shape == (c.Q, 1)
    sel = np.random.default_rng(0).choice(c.Q, size=60000, replace=False)
    o, d, v = (a[sel] for a in c.rays_np())
    ot = c.oracle_tree()
    want = O.volume_render(ot, o, d, v, c.oracle_opts())
    np.testing.assert_array_equal(out[sel].cpu().numpy(), want)
    np.testing.assert_array_equal(depth[sel].cpu().numpy(), O.render_depth(ot, o, d, v, c.oracle_opts()))
    # the same through the one-kernel forward (the default for this payload is march + channel-lane shade)
    monkeypatch.setattr(_C, "FWD_SPLIT", "0")
    with torch.no_grad():
        assert torch.equal(r(tree.features, rays), out)
    assert _C.LAST_ROUTE["forward"].startswith("render_fwd_kernel")
    monkeypatch.setattr(_C, "FWD_SPLIT", "")
    # opt-in tolerance mode (NATIVE_MATH: bit-exact stepping, v_exp_f32 / v_rcp_f32 shading): 1e-5 relative, at full size
    monkeypatch.setattr(_C, "NATIVE_MATH", True)
    with torch.no_grad():
        fast = r(tree.features, rays)
    assert "native" in _C.LAST_ROUTE["forward"]
    monkeypatch.setattr(_C, "NATIVE_MATH", False)
    assert not torch.equal(fast, out)
    assert_outputs_close(fast[sel].cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    # every pixel, against the exact mode's (= the oracle's, bit for bit on the subsample): the same criterion.
    # (No tighter one holds for ANY other exponential, CUDA's own included: a thin sample's weight T (1 - att)
    # carries att's last-place error divided by 1 - att.)
    assert ((fast - out).abs() <= 1e-5 * out.abs() + 1e-6).all()
    # depth is 0 exactly where nothing was hit, else inside the cube's extent
    dn, an = depth.cpu().numpy()[:, 0], out[:, 31].cpu().numpy()
    assert np.all((dn == 0) == (an == 0))
    assert dn.max() < 1.6 + 0.9


def test_config4_backward_full_size(cfg4, gpu, monkeypatch):
    """BASELINE configs[3] forward + backward at full size (1 048 576 rays, 13.4 M samples of 31
    channels): the per-tile backward for wide rows (grad_wide_kernel) and the per-ray one-sigmoid-pass
    form against the oracle's two-pass backward, every entry within 1e-5 of the TIGHT scale (accum
    priced by the reference's own sequential addends)."""
    c = cfg4.case
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    gout, want, _, tight = cfg4.backward()
    for gather, expect in ((1, "grad_wide_kernel"), (0, "render_bwd_kernel<ONEPASS>")):
        monkeypatch.setattr(_C, "BWD_GATHER", gather)
        tree.features.grad = None
        out = r(tree.features, c.rays_gpu(gpu), image_shape=(1024, 1024))
        out.backward(gout.to(gpu))
        assert _C.LAST_ROUTE["backward"].startswith(expect), _C.LAST_ROUTE
        assert_grads_close(tree.features.grad.cpu().numpy(), want, tight, what=expect)
        tree.features.grad = None
    # the device-side count of what the per-tile kernel sends to memory (bench.py prices its roofline from
    # it): one request per 64 bytes of a (tile, window, row) group's gradient row -- two per row of 32 floats
    monkeypatch.setattr(_C, "BWD_GATHER", 1)
    with _C.bwd_counters(gpu) as ctr:
        out = r(tree.features, c.rays_gpu(gpu), image_shape=(1024, 1024))
        out.backward(gout.to(gpu))
        torch.cuda.synchronize()
    requests, groups = ctr.read()
    assert requests == 2 * groups and 1_000_000 < groups < 13_400_000, (requests, groups)
    assert_grads_close(tree.features.grad.cpu().numpy(), want, tight, what="counting instance")


def test_config4_native_math_tolerance_mode_full_size(cfg4, gpu, monkeypatch, capsys):
    """The opt-in tolerance mode for wide rows (NATIVE_MATH / SVOXT_NATIVE_MATH=1) at
    BASELINE configs[3]'s full size, forward AND backward: the stepping is the exact one -- same leaves,
    same lists -- the shading takes its exponentials from v_exp_f32 and its quotients from v_rcp_f32
    where the reference has expf and a double-precision divide (rt_kernel.cu:280, 300-305, 397, 408-425,
    461-476).  Outputs: the north star's figure, |err| <= 1e-5 |want| + 1e-6.  Gradients: what two
    exponentials that differ in the last place can agree to (see below); the fraction of entries off by more
    than 1e-5 of their OWN value is reported for both modes (the exact mode's is float-atomic reordering alone)."""
    c = cfg4.case
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    gout, want, absum, tight = cfg4.backward()
    want_out = O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts())
    touched = absum > 0
    stats = {}
    for mode, native in (("exact", False), ("native", True)):
        monkeypatch.setattr(_C, "NATIVE_MATH", native)
        tree.features.grad = None
        out = r(tree.features, c.rays_gpu(gpu), image_shape=(1024, 1024))
        out.backward(gout.to(gpu))
        assert ("native" in _C.LAST_ROUTE["forward"]) == native and ("native" in _C.LAST_ROUTE["backward"]) == native
        assert _C.LAST_ROUTE["backward"].startswith("grad_wide_kernel")
        got_out = out.detach().cpu().numpy()
        got = tree.features.grad.double().cpu().numpy()
        if native:
            assert_outputs_close(got_out, want_out, rtol=1e-5, atol=1e-6, what="native-math forward")
        else:
            np.testing.assert_array_equal(got_out, want_out)
        err = np.abs(got - want)
        if native:
            # Two exponentials that differ in the last place cannot agree better than ulp(1) / (1 - att) on a
            # thin sample's weight (tests/test_oracle_kat.py shows it with the oracle alone): every entry within
            # 1e-4 of the tight scale (the a-priori bound 6e-8 / (step_size * min sigma)), >= 99 % within 1e-5
            assert_grads_close(got, want, tight, rtol=1e-4, what="native backward")
            assert (err <= 1e-5 * tight + 1e-30)[touched].mean() >= 0.99
        else:
            assert_grads_close(got, want, tight, what="exact backward")
        oerr = np.abs(got_out.astype(np.float64) - want_out)
        stats[mode] = (float((oerr > 1e-5 * np.abs(want_out)).mean()), float((err > 1e-5 * np.abs(want))[touched].mean()),
                       float((err / (tight + 1e-300))[touched].max()))
    with capsys.disabled():
        for k, (fo, fg, worst) in stats.items():
            print(f"\n[cfg4, {k}] outputs with |err| > 1e-5 |want|: {fo:.4%}; gradient entries with |err| > 1e-5 |want|: "
                  f"{fg:.4%}; worst |err| / tight scale {worst:.2e}")
    assert stats["native"][2] <= 1e-4 and stats["exact"][2] <= 1e-5


@pytest.mark.parametrize("width,height,ndc", [(64, 48, False), (50, 37, False), (64, 48, True), (33, 40, True)])
def test_render_persp_generates_the_reference_rays_in_kernel(gpu, width, height, ndc):
    """render_persp -> volume_render_image with a CameraSpec: the kernels generate
    cam2world_ray (+ maybe_world2ndc) rays themselves.  Bit-exact against the oracle
    rendering the rays of its own restatement of rt_kernel.cu:1153-1190, forward
    and (to the gradient tolerance) backward; and equal to the ray-batch render of
    the same rays."""
    c = Case(depth=5, K=28, data_format="SH9", width=8, height=8)
    tree = c.tree(gpu)
    fx = 1111.111 * width / 800.0
    fy = fx * 1.1
    cfg = svox.NDCConfig(width, height, fx) if ndc else None
    r = svox.VolumeRenderer(tree, ndc=cfg)
    if ndc:      # a forward-facing camera in front of the near plane, as NDC scenes have
        pose = np.eye(4, dtype=np.float32)
        pose[:3, 3] = [0.45, 0.55, 2.2]
    else:
        pose = synth.camera_pose(azimuth_deg=70.0, elevation_deg=-15.0).astype(np.float32)
    c2w = torch.from_numpy(pose)
    feats = tree.features.detach().clone().requires_grad_(True)
    img = r.render_persp(feats, c2w.to(gpu), width=width, height=height, fx=fx, fy=fy)
    assert img.shape == (height, width, 4)
    o, d, v = O.camera_rays(pose, fx, fy, width, height, ndc=(width, height, fx) if ndc else None)
    opt = c.oracle_opts()
    if ndc:
        opt.ndc_width, opt.ndc_height, opt.ndc_focal = width, height, fx
    want = O.volume_render(c.oracle_tree(), o, d, v, opt)
    assert (want[:, 3] > 0.05).mean() > 0.05                      # the camera does see the shell
    np.testing.assert_array_equal(img.detach().reshape(-1, 4).cpu().numpy(), want)
    # the same rays through the ray-batch entry give the same image
    with torch.no_grad():
        batch = svox.VolumeRenderer(tree)(feats, svox.Rays(*(torch.from_numpy(a).to(gpu) for a in (o, d, v))))
    assert torch.equal(batch, img.detach().reshape(-1, 4))
    # backward
    g = torch.Generator().manual_seed(4)
    gout = torch.randn(height, width, 4, generator=g)
    img.backward(gout.to(gpu))
    wg, wabs = O.volume_render_backward(c.oracle_tree(), o, d, v, opt, gout.reshape(-1, 4).numpy(), want_abs=True)
    assert_grads_close(feats.grad.cpu().numpy(), wg, wabs)
    # depth / opacity operators accept the camera too
    cam = _C.CameraSpec()
    cam.c2w, cam.fx, cam.fy, cam.width, cam.height = c2w.to(gpu), fx, fy, width, height
    spec = tree._spec(tree.features)
    np.testing.assert_array_equal(_C.render_depth(spec, cam, r._get_options()).cpu().numpy(),
                                  O.render_depth(c.oracle_tree(), o, d, v, opt))
    np.testing.assert_array_equal(_C.opacity_render(spec, cam, r._get_options()).cpu().numpy(),
                                  O.opacity_render(c.oracle_tree(), o, d, v, opt))


@pytest.mark.parametrize("fmt,K", [("SH16", 49), ("SG9", 28), ("ASG16", 49)])
def test_render_persp_with_the_payloads_added_in_r03(gpu, fmt, K):
    """The camera route (rays generated in the kernels) through what round 3 added: SH16's rows of 49 floats and
    SG / ASG lobes in the recording forward and in grad_fused_kernel, whose camera branch forms the view direction
    itself (the lobes' basis values come from it).  Image bit-exact against the oracle on its own camera rays,
    gradient on the tight scale."""
    width = height = 64
    c = Case(depth=5, K=K, data_format=fmt, width=8, height=8)
    B = c.basis_dim
    extra, lobes = None, None
    g = torch.Generator().manual_seed(11)
    if fmt.startswith("SG"):
        lobes = torch.cat([torch.rand(B, 1, generator=g) * 4 + 0.5,
                           torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)], -1).contiguous()
    elif fmt.startswith("ASG"):
        fr = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0]
        lobes = torch.cat([torch.rand(B, 2, generator=g) * 3 + 0.3, fr.reshape(B, 9)], -1).contiguous()
    tree = svox.N3Tree.from_arrays(c.st.child, c.st.data, c.st.parent_depth, c.features, data_format=fmt,
                                   extra_data=lobes, device=gpu)
    ot = O.Tree(c.features.numpy(), c.st.data, c.st.child, extra=None if lobes is None else lobes.numpy())
    opt = O.make_options(format=c.format, basis_dim=B)
    fx = 1111.111 * width / 800.0
    pose = synth.camera_pose(azimuth_deg=40.0, elevation_deg=20.0).astype(np.float32)
    r = svox.VolumeRenderer(tree)
    feats = tree.features.detach().clone().requires_grad_(True)
    img = r.render_persp(feats, torch.from_numpy(pose).to(gpu), width=width, height=height, fx=fx)
    o, d, v = O.camera_rays(pose, fx, fx, width, height)
    want = O.volume_render(ot, o, d, v, opt)
    assert (want[:, 3] > 0.05).mean() > 0.05
    np.testing.assert_array_equal(img.detach().reshape(-1, 4).cpu().numpy(), want)
    gout = torch.randn(height, width, 4, generator=g)
    img.backward(gout.to(gpu))
    assert _C.LAST_ROUTE["backward"].startswith("grad_fused_kernel<EXACT>"), _C.LAST_ROUTE
    wg, wabs, tight = O.volume_render_backward(ot, o, d, v, opt, gout.reshape(-1, 4).numpy(), want_abs="both")
    assert_grads_close(feats.grad.cpu().numpy(), wg, tight)


def test_torch_ray_generator_matches_the_oracle_camera(gpu):
    """renderer.pinhole_rays (a torch utility for callers that want the ray tensors)
    == the oracle's cam2world_ray / maybe_world2ndc restatement."""
    from svox_t_amd.renderer import pinhole_rays
    pose = synth.camera_pose(azimuth_deg=10.0, elevation_deg=35.0).astype(np.float32)
    fx = 91.0
    o, d, v = pinhole_rays(torch.from_numpy(pose).to(gpu), 40, 24, fx, fx)
    wo, wd, wv = O.camera_rays(pose, fx, fx, 40, 24)
    np.testing.assert_array_equal(o.cpu().numpy(), wo)
    np.testing.assert_array_equal(d.cpu().numpy(), wd)
    np.testing.assert_array_equal(v.cpu().numpy(), wv)
    # against the float64 construction used for the benchmark inputs
    _, d2, _ = synth.pinhole_rays(40, 24, c2w=pose.astype(np.float64), fx=fx)
    np.testing.assert_allclose(wd, d2.numpy(), atol=3e-7)


def test_camera_argument_errors(gpu):
    c = Case(depth=3, K=4, data_format="RGBA", width=8, height=8)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    with pytest.raises(RuntimeError):
        r.render_persp(tree.features, torch.eye(4), width=0, height=8)
    cam = _C.CameraSpec()
    cam.c2w, cam.fx, cam.fy, cam.width, cam.height = torch.eye(4), 10.0, 10.0, 8, 8     # not on the GPU
    with pytest.raises(RuntimeError):
        _C.volume_render_image(tree._spec(tree.features), cam, r._get_options())
    cam.c2w = torch.eye(3, device=gpu)                                                   # not [*, 4]
    with pytest.raises(RuntimeError):
        _C.volume_render_image(tree._spec(tree.features), cam, r._get_options())
    cam.c2w, cam.fx = torch.eye(4, device=gpu), 0.0
    with pytest.raises(RuntimeError):
        _C.volume_render_image(tree._spec(tree.features), cam, r._get_options())


def test_acceleration_grid_never_goes_stale(gpu):
    """The grid is a cache of child/data: a new tree at a recycled device address,
    an in-place refine, and an in-place change of `data` must each be seen."""
    import gc
    c = Case(depth=5, K=4, data_format="RGBA", width=40, height=40)
    rays = c.rays_gpu(gpu)
    opt = c.oracle_opts()
    outs = []
    for variant in range(3):          # same shapes, different contents, allocated one after the other
        st = c.st
        data = st.data.copy()
        if variant:
            flat = data.reshape(-1)
            occ = np.flatnonzero(flat != synth.EMPTY_SENTINEL)
            flat[occ] = np.random.default_rng(variant).permutation(flat[occ])
        tree = svox.N3Tree.from_arrays(st.child, data, st.parent_depth, c.features, device=gpu)
        r = svox.VolumeRenderer(tree)
        with torch.no_grad():
            got = r(tree.features, rays).cpu().numpy()
        want = O.volume_render(O.Tree(c.features.numpy(), data, st.child), *c.rays_np(), opt)
        np.testing.assert_array_equal(got, want)
        outs.append(got)
        del tree, r
        gc.collect()
    assert not np.array_equal(outs[0], outs[1])
    # in-place topology and data changes on a live tree
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    with torch.no_grad():
        r(tree.features, rays)                                     # builds the grid
        leaves = tree._all_leaves()
        sel = leaves[tree.parent_depth[leaves[:, 0].to(gpu), 1].cpu() == 4][:50]
        tree.refine(1, sel=tuple(sel.T), leaf_node=sel)            # children inherit the parent's row
        got = r(tree.features, rays).cpu().numpy()
    n = tree.n_internal
    ot = O.Tree(c.features.numpy(), tree.data[:n].cpu().numpy(), tree.child[:n].cpu().numpy())
    np.testing.assert_array_equal(got, O.volume_render(ot, *c.rays_np(), opt))
    with torch.no_grad():
        tree.data[tree.data < tree.features.shape[0]] = 7          # every occupied leaf -> row 7
        got = r(tree.features, rays).cpu().numpy()
    ot = O.Tree(c.features.numpy(), tree.data[:n].cpu().numpy(), tree.child[:n].cpu().numpy())
    np.testing.assert_array_equal(got, O.volume_render(ot, *c.rays_np(), opt))
    # `tensor.data = other` swaps the storage WITHOUT bumping the version counter: the cache is keyed on the
    # data pointer as well
    with torch.no_grad():
        other = tree.data.clone()
        other[other == 7] = 11
        v0 = tree.data._version
        tree.data.data = other
        assert tree.data._version == v0
        got = r(tree.features, rays).cpu().numpy()
    ot = O.Tree(c.features.numpy(), tree.data[:n].cpu().numpy(), tree.child[:n].cpu().numpy())
    np.testing.assert_array_equal(got, O.volume_render(ot, *c.rays_np(), opt))


@pytest.mark.parametrize("fmt,K,depth,mode", [
    ("SH9", 28, 5, "lists"),          # specialised kernels, forward-recorded lists, single-march backward
    ("SH9", 28, 6, "overflow"),       # ... with lists of 4 samples: most rays march their tail
    ("SH9", 28, 5, "exact"),          # ... two list walks (SVOXT_BWD_EXACT)
    ("SH9", 28, 5, "standalone"),     # ... backward without forward lists (own workspace)
    ("SH4", 13, 5, "lists"),
    ("SH9", 28, 5, "two_kernel"),     # ... list walk + per-tile merge (rotated directions travel with the records)
    ("SH4", 13, 6, "two_kernel_overflow"),
    ("SH9", 28, 5, "fused"),          # ... as ONE kernel (grad_fused_kernel<..., XF>: a basis per record, r04) over the forward's hand-over
    ("SH9", 28, 5, "fused_noterms"),  # ... without hand-over: both sweeps gather the rows and form every record's basis
    ("SH9", 28, 6, "fused_overflow"), # ... rays whose list overflowed go whole through the per-ray kernel in front
    ("SH4", 13, 6, "fused"),
    ("SH1", 4, 5, "fused"),
    ("SH16", 49, 4, "lists"),
    ("SG6", 19, 4, "generic"),        # view-dependent format without a specialised kernel
])
def test_transformation_matrices(gpu, monkeypatch, fmt, K, depth, mode):
    """Per-leaf rotation of the view direction (rt_kernel.cu:283-291, :387-395),
    including the reference's quirk that pass 2 of the backward keeps the basis
    of the last sample of pass 1 -- through every route the backward can take."""
    c = Case(depth=depth, K=K, data_format=fmt, width=48, height=48)
    extra = None
    if fmt.startswith("SG"):
        ge = torch.Generator().manual_seed(4)
        extra = torch.cat([torch.rand(6, 1, generator=ge) * 4 + 0.5,
                           torch.nn.functional.normalize(torch.randn(6, 3, generator=ge), dim=-1)], -1)
    tree = svox.N3Tree.from_arrays(c.st.child, c.st.data, c.st.parent_depth, c.features, data_format=fmt,
                                   extra_data=extra, device=gpu)
    M = tree.features.shape[0]
    g = torch.Generator().manual_seed(9)
    A = torch.randn(M, 3, 3, generator=g)
    Qm, _ = torch.linalg.qr(A)                                # random rotations
    if mode in ("overflow", "two_kernel_overflow"):
        monkeypatch.setattr(_C, "BWD_LIST_SAMPLES", 4)
    if mode in ("fused_overflow",):
        monkeypatch.setattr(_C, "BWD_LIST_SAMPLES", 8)
    monkeypatch.setattr(_C, "BWD_GATHER", 2 if mode.startswith(("two_kernel", "fused")) else 0)
    monkeypatch.setattr(_C, "BWD_XF_FUSED", mode.startswith("fused"))
    if mode == "fused_noterms":
        monkeypatch.setattr(_C, "BWD_TERMS", False)
    if mode == "exact":
        monkeypatch.setattr(_C, "BWD_EXACT", True)
    r = svox.VolumeRenderer(tree)
    rays = c.rays_gpu(gpu)
    xf = Qm.to(gpu).contiguous()
    ot = O.Tree(c.features.numpy(), c.st.data, c.st.child, extra=None if extra is None else extra.numpy())
    opt = c.oracle_opts()
    spec = tree._spec(tree.features, transformation_matrices=xf)
    assert _C.can_record(spec, r._get_options()) == (mode != "generic")
    out = r(tree.features, rays, transformation_matrices=xf)
    with O.transformation_matrices(Qm.numpy()):
        want = O.volume_render(ot, *c.rays_np(), opt)
        gout = synth.grad_output(c.Q, 4)
        gw, ab = O.volume_render_backward(ot, *c.rays_np(), opt, gout.numpy(), want_abs=True)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
    plain = O.volume_render(ot, *c.rays_np(), opt)
    assert fmt == "SH1" or np.abs(plain - want).max() > 1e-3   # the rotations do change the image (a constant basis aside)
    if mode == "standalone":
        grad = _C.volume_render_backward(spec, _rays_spec_from_rays(rays), r._get_options(), gout.to(gpu))
        assert_grads_close(grad.cpu().numpy(), gw, ab)
    else:
        out.backward(gout.to(gpu))
        assert_grads_close(tree.features.grad.cpu().numpy(), gw, ab)
        if mode.startswith("fused"):
            assert _C.LAST_ROUTE["backward"].startswith("grad_fused_kernel<EXACT, XF>"), _C.LAST_ROUTE
            assert _C.LAST_ROUTE["forward"].startswith("fwd_roles_kernel<XF>"), _C.LAST_ROUTE
            assert _C.LAST_ROUTE["forward_terms"] == (mode != "fused_noterms")
        elif mode.startswith("two_kernel"):
            assert _C.LAST_ROUTE["backward"].startswith("render_bwd_kernel<GATHER>"), _C.LAST_ROUTE
    if fmt != "SH9" or mode != "lists":
        return
    # identity matrices reproduce the plain render exactly
    eye = torch.eye(3).repeat(M, 1, 1).contiguous()
    with torch.no_grad():
        same = r(tree.features, rays, transformation_matrices=eye.to(gpu)).cpu().numpy()
    np.testing.assert_array_equal(same, plain)


def test_render_sharded_and_render_cameras_single_rank_on_gpu(gpu):
    """parallel.render_sharded / render_cameras with the real renderer (world size 1, gloo):
    the same pixels and gradients as the direct calls."""
    import os
    import torch.distributed as dist
    from svox_t_amd import parallel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        c = Case(depth=5, K=28, data_format="SH9", width=48, height=40)
        tree = c.tree(gpu)
        r = svox.VolumeRenderer(tree)
        rays = c.rays_gpu(gpu)
        g = synth.grad_output(c.Q, 4).to(gpu)
        f1 = tree.features.detach().clone().requires_grad_(True)
        out1 = parallel.render_sharded(r, f1, rays, image_shape=(40, 48))
        out1.backward(g)
        f2 = tree.features.detach().clone().requires_grad_(True)
        out2 = r(f2, rays, image_shape=(40, 48))
        out2.backward(g)
        assert torch.equal(out1, out2)
        want, ab = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.cpu().numpy(), want_abs=True)
        assert_grads_close(f1.grad.cpu().numpy(), want, ab)
        assert_grads_close(f2.grad.cpu().numpy(), want, ab)
        poses = torch.stack([torch.from_numpy(synth.camera_pose(azimuth_deg=a).astype(np.float32)) for a in (20.0, 110.0)])
        imgs = parallel.render_cameras(r, f1, poses.to(gpu), width=32, height=24, fx=40.0)
        assert imgs.shape == (2, 24, 32, 4)
        with torch.no_grad():
            assert torch.equal(imgs[1], r.render_persp(f1, poses[1].to(gpu), width=32, height=24, fx=40.0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fmt,K", [("SH9", 28), ("RGBA", 4)])
def test_ray_order_is_a_permutation_and_changes_no_result(gpu, fmt, K):
    """svoxt_ray_order: a batch that is not an image is rendered in the order of its rays' entry
    points into the cube (coherent wavefronts).  The order is a permutation with the misses
    last; outputs are per ray -- bit-identical with and without it, at the ray's own row --
    and the gradient is the same sum."""
    c = Case(depth=6, K=K, data_format=fmt, width=96, height=96)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    o, d, v = c.rays_np()
    rng = np.random.default_rng(3)
    shuffle = rng.permutation(len(o))
    o, d, v = o[shuffle].copy(), d[shuffle].copy(), v[shuffle].copy()
    o[:50] += 10.0                                           # some rays that miss the cube
    rays = svox.Rays(*(torch.from_numpy(x).to(gpu) for x in (o, d, v)))
    opt = r._get_options()
    perm = _C.ray_order(tree._spec(tree.features), _rays_spec_from_rays(rays), opt).cpu().numpy()
    assert sorted(perm.tolist()) == list(range(len(o)))
    want = O.volume_render(c.oracle_tree(), o, d, v, c.oracle_opts())
    # the rays that miss the unit cube (this tree's world box) are the tail of the order
    dn = d / np.linalg.norm(d, axis=1, keepdims=True)
    inv = 1.0 / (dn.astype(np.float64) + 1e-9)
    t1, t2 = -o * inv, (1.0 - o) * inv
    tmin = np.maximum(np.minimum(t1, t2).max(axis=1), 0.0)
    tmax = np.maximum(t1, t2).min(axis=1)
    miss = tmax < tmin
    assert miss[:50].all() and 50 <= miss.sum() < len(o)
    assert set(perm[len(o) - miss.sum():].tolist()) == set(np.nonzero(miss)[0].tolist())
    from svox_t_amd import synth as S
    g = S.grad_output(len(o), want.shape[1], seed=7)
    outs, grads = [], []
    for sort_rays in (False, True):
        tree.features.grad = None
        out = r(tree.features, rays, sort_rays=sort_rays)
        out.backward(g.to(gpu))
        outs.append(out.detach().cpu().numpy())
        grads.append(tree.features.grad.cpu().numpy())
    np.testing.assert_array_equal(outs[0], want)
    np.testing.assert_array_equal(outs[1], want)
    gw, ab = O.volume_render_backward(c.oracle_tree(), o, d, v, c.oracle_opts(), g.numpy(), want_abs=True)
    assert_grads_close(grads[0], gw, ab)
    assert_grads_close(grads[1], gw, ab)
    # the opacity pair takes the same route
    a0 = r.opacity_render(tree.features, rays, sort_rays=False).detach().cpu().numpy()
    a1 = r.opacity_render(tree.features, rays, sort_rays=True).detach().cpu().numpy()
    np.testing.assert_array_equal(a0, a1)
    # images are walked in tiles already: the ordering entry point refuses them
    with pytest.raises(RuntimeError):
        _C.ray_order(tree._spec(tree.features), _rays_spec_from_rays(rays, image_shape=(96, 96)), opt)


@pytest.mark.parametrize("M,K", [(1, 4), (63, 8), (64, 28), (1000, 32), (70001, 28)])
def test_sigma_mask_build_bits(gpu, M, K):
    """svoxt_sigma_mask_build: bit (row & 31) of 32-bit word (row >> 5) = features[row, K-1] > thresh, for
    row counts that are not multiples of the 1024 rows a workgroup covers; bits past M are clear; NaN
    sigmas and a NaN threshold give no bit (the comparison the march makes is `sigma > thresh`)."""
    import ctypes
    rng = np.random.default_rng(M)
    f = rng.standard_normal((M, K)).astype(np.float32)
    if M > 10:
        f[3, -1] = np.nan
        f[5, -1] = np.inf
        f[7, -1] = 0.25                      # equal to the threshold: not greater
    feats = torch.from_numpy(f).to(gpu)
    ct = _C._CTree(features=feats.data_ptr(), M=M, K=K, N=2, data=feats.data_ptr(), child=feats.data_ptr(), n_internal=1,
                   offset=feats.data_ptr(), scaling=feats.data_ptr())
    nbytes = _C._lib.svoxt_sigma_mask_bytes(M)
    assert nbytes == (M + 63) // 64 * 8
    for thresh in (0.0, 0.25, float("nan")):
        mask = torch.full((nbytes // 8,), -1, dtype=torch.int64, device=gpu)
        _C._call("svoxt_sigma_mask_build", ctypes.byref(ct), ctypes.c_float(thresh), mask.data_ptr(), None)
        torch.cuda.synchronize()
        words = mask.cpu().numpy().view(np.uint32)
        bits = ((words[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).reshape(-1).astype(bool)
        with np.errstate(invalid="ignore"):
            want = f[:, -1] > np.float32(thresh)
        np.testing.assert_array_equal(bits[:M], want)
        assert not bits[M:].any()


@pytest.mark.parametrize("M,K,fill_words", [(1, 4, 4), (1000, 32, 4 * 1024 + 4), (70001, 28, 290_004), (64, 28, 0)])
def test_sigma_mask_build_fill(gpu, M, K, fill_words):
    """svoxt_sigma_mask_build_fill (ABI v18): the same mask as svoxt_sigma_mask_build and, from the same launch,
    exactly fill_bytes bytes of 0xff at `fill` -- nothing in front of them, nothing behind; misaligned or
    ragged fills are refused."""
    import ctypes
    rng = np.random.default_rng(M + 1)
    feats = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(gpu)
    ct = _C._CTree(features=feats.data_ptr(), M=M, K=K, N=2, data=feats.data_ptr(), child=feats.data_ptr(), n_internal=1,
                   offset=feats.data_ptr(), scaling=feats.data_ptr())
    words = _C._lib.svoxt_sigma_mask_bytes(M) // 8
    want = torch.zeros((words,), dtype=torch.int64, device=gpu)
    _C._call("svoxt_sigma_mask_build", ctypes.byref(ct), ctypes.c_float(0.1), want.data_ptr(), None)
    got = torch.zeros((words,), dtype=torch.int64, device=gpu)
    guard = 8
    buf = torch.full((guard + fill_words + guard,), 0x12345678, dtype=torch.int32, device=gpu)
    fill = buf[guard: guard + fill_words] if fill_words else None
    _C._call("svoxt_sigma_mask_build_fill", ctypes.byref(ct), ctypes.c_float(0.1), got.data_ptr(),
             None if fill is None else fill.data_ptr(), fill_words * 4, None)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert (buf[:guard] == 0x12345678).all() and (buf[guard + fill_words:] == 0x12345678).all()
    if fill_words:
        assert (fill == -1).all()
        with pytest.raises(RuntimeError, match="fill"):       # 4 bytes past a 16-byte boundary
            _C._call("svoxt_sigma_mask_build_fill", ctypes.byref(ct), ctypes.c_float(0.1), got.data_ptr(),
                     buf[guard + 1:].data_ptr(), 16, None)
        with pytest.raises(RuntimeError, match="fill"):       # not a multiple of 16 bytes
            _C._call("svoxt_sigma_mask_build_fill", ctypes.byref(ct), ctypes.c_float(0.1), got.data_ptr(),
                     fill.data_ptr(), 20, None)


def test_overflow_word_of_pooled_lists(gpu, monkeypatch):
    """Word 1 of the pool's counter block (ABI v18): -1 after a recording forward none of whose rays filled its list,
    1 after one that did (lists of 8 records on a depth-6 tree) -- what the tail launches read with one scalar load
    instead of every ray's list length.  The results do not depend on it (the oracle holds both)."""
    c = Case(depth=6, K=13, data_format="SH4", width=64, height=48)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    opt = r._get_options()
    spec = tree._spec(tree.features)
    rs = _rays_spec_from_rays(c.rays_gpu(gpu), (48, 64))
    rs.need_grad = False
    want = O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts())
    g = synth.grad_output(c.Q, 4)
    gw, ab = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs=True)
    for cap, word in ((96, -1), (8, 1)):
        monkeypatch.setattr(_C, "BWD_LIST_SAMPLES", cap)
        _C._POOL_HINT.clear()       # (a pool sized by another test's batch of this shape may run dry: that is an overflow too)
        out, lists = _C.volume_render(spec, rs, opt, record=True)
        torch.cuda.synchronize()
        over = int((lists.aux[:, 0] < 0).sum())                    # bit 31 of aux.x: the ray's list overflowed
        assert (over > 0) == (word == 1)
        assert int(lists.pool_next[1]) == word
        np.testing.assert_array_equal(out.cpu().numpy(), want)
        grad = _C.volume_render_backward(spec, rs, opt, g.to(gpu), lists=lists)
        assert_grads_close(grad.cpu().numpy(), gw, ab)
