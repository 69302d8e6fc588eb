"""HIP kernels vs the CPU oracle on identical inputs.
All calls go through svox_t_amd's Python surface -> ctypes -> C ABI."""
import numpy as np
import pytest
import torch

import svox_t_amd as svox
from oracle import oracle as O
from tests.util import Case, assert_grads_close, assert_outputs_close

pytestmark = pytest.mark.gpu

CASES = {
    # BASELINE config 1: depth-5, K=4 RGBA, 64x64
    "d5_rgba4": dict(depth=5, K=4, data_format="RGBA", width=64, height=64),
    "d5_sh9": dict(depth=5, K=28, data_format="SH9", width=64, height=64),
    "d6_rgba32": dict(depth=6, K=32, data_format="RGBA", width=96, height=96),
    # RGBA-style rows of 8 / 16 floats: channel-lane shade kernel, one-kernel forward / backward instances for C = 7 / 15
    "d5_rgba8": dict(depth=5, K=8, data_format="RGBA", width=64, height=64),
    "d5_rgba16": dict(depth=5, K=16, data_format="RGBA", width=56, height=56),
    "d5_sh4_world": dict(depth=5, K=13, data_format="SH4", width=64, height=64,
                         radius=[1.0, 1.2, 0.8], center=[0.1, -0.2, 0.3]),
    "d5_sh1": dict(depth=5, K=4, data_format="SH1", width=48, height=48),
    "d4_sh16": dict(depth=4, K=49, data_format="SH16", width=48, height=48),
    "d4_sh25": dict(depth=4, K=76, data_format="SH25", width=48, height=48),
    # generic kernel: 2 channels x SH4
    "d5_generic": dict(depth=5, K=9, data_format="SH4", width=48, height=48),
}


@pytest.fixture(scope="module", params=list(CASES))
def case(request):
    return Case(**CASES[request.param])


@pytest.mark.parametrize("fast", [False, True])
def test_volume_render_forward(case, gpu, fast):
    tree = case.tree(gpu)
    r = svox.VolumeRenderer(tree)
    with torch.no_grad():
        got = r(tree.features, case.rays_gpu(gpu), fast=fast).cpu().numpy()
    want = O.volume_render(case.oracle_tree(), *case.rays_np(), case.oracle_opts(fast=fast))
    assert_outputs_close(got, want)          # the north-star tolerance: 1e-5 relative
    # ... and in fact bit for bit: the march is exact and both sides evaluate
    # the same correctly rounded operation sequence (svoxt_device.h).
    np.testing.assert_array_equal(got, want)


def test_volume_render_backward(case, gpu):
    tree = case.tree(gpu)
    r = svox.VolumeRenderer(tree)
    feats = tree.features
    out = r(feats, case.rays_gpu(gpu))
    from svox_t_amd import synth
    g = synth.grad_output(case.Q, out.shape[1])
    out.backward(g.to(gpu))
    got = feats.grad.cpu().numpy()
    # the default backward is exact (every contribution the reference's formula, accum added up in the
    # reference's order): held to the TIGHT scale -- accum priced by the reference's own sequential addends
    want, abs_sum, tight = O.volume_render_backward(case.oracle_tree(), *case.rays_np(), case.oracle_opts(),
                                                    g.numpy(), want_abs="both")
    assert_grads_close(got, want, tight)
    assert np.all(got[abs_sum == 0] == 0)


def test_depth_and_opacity(case, gpu):
    tree = case.tree(gpu)
    r = svox.VolumeRenderer(tree)
    rays = case.rays_gpu(gpu)
    ot, opt = case.oracle_tree(), case.oracle_opts()
    with torch.no_grad():
        depth = r.render_depth(tree.features, rays).cpu().numpy()
        alpha = r.opacity_render(tree.features, rays).cpu().numpy()
    # depth = delta_scale * t of a bit-identical march: exact
    np.testing.assert_array_equal(depth, O.render_depth(ot, *case.rays_np(), opt))
    np.testing.assert_array_equal(alpha, O.opacity_render(ot, *case.rays_np(), opt))


def test_opacity_backward(case, gpu):
    tree = case.tree(gpu)
    r = svox.VolumeRenderer(tree)
    feats = tree.features
    out = r.opacity_render(feats, case.rays_gpu(gpu))
    from svox_t_amd import synth
    g = synth.grad_output(case.Q, 1, seed=3)
    out.backward(g.to(gpu))
    got = feats.grad.cpu().numpy()
    want, _, tight = O.volume_render_backward(case.oracle_tree(), *case.rays_np(), case.oracle_opts(),
                                              g.numpy(), want_abs="both")
    assert_grads_close(got, want, tight)
    assert np.all(got[:, :-1] == 0)          # only the sigma column receives gradient


@pytest.mark.parametrize("list_samples", [0, 1, 3, 64])
@pytest.mark.parametrize("tiled", [False, True])
def test_opacity_backward_from_sample_lists(gpu, list_samples, tiled, monkeypatch):
    """opacity_render records sample lists when a backward will follow; the backward walks
    them and sums per tile (rays with more samples than the list holds march their tail).
    Any capacity, tiled or not, gives the reference's gradient; the forward stays bit-exact;
    a second backward over the same graph (lists already rewritten) marches instead."""
    import svox_t_amd.csrc as _C
    from svox_t_amd import synth
    monkeypatch.setattr(_C, "BWD_LIST_SAMPLES", list_samples)
    c = Case(depth=6, K=28, data_format="SH9", width=96, height=96)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    g = synth.grad_output(c.Q, 1, seed=5)
    out = r.opacity_render(tree.features, c.rays_gpu(gpu), image_shape=(96, 96) if tiled else None)
    np.testing.assert_array_equal(out.detach().cpu().numpy(),
                                  O.opacity_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts()))
    want, abs_sum = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.numpy(),
                                             want_abs=True)
    for _ in range(2):
        tree.features.grad = None
        out.backward(g.to(gpu), retain_graph=True)
        got = tree.features.grad.cpu().numpy()
        assert_grads_close(got, want, abs_sum)
        assert np.all(got[:, :-1] == 0)


def test_counters_match_oracle(case, gpu):
    """The march is bit-identical, so the step / level / sample counts are."""
    import svox_t_amd.csrc as _C
    from svox_t_amd.renderer import _rays_spec_from_rays
    tree = case.tree(gpu)
    r = svox.VolumeRenderer(tree)
    cnt = _C.count_forward(tree._spec(tree.features), _rays_spec_from_rays(case.rays_gpu(gpu)),
                           r._get_options()).cpu().tolist()
    _, want = O.volume_render(case.oracle_tree(), *case.rays_np(), case.oracle_opts(), count=True)
    assert tuple(cnt) == tuple(want)


@pytest.mark.parametrize("list_samples", [0, 1, 3, 64])
@pytest.mark.parametrize("from_forward", [True, False, "exact", "one_kernel"])
def test_backward_sample_list_capacity_does_not_change_results(gpu, list_samples, from_forward, monkeypatch):
    """The backward records up to S samples per ray in pass 1 and replays them in
    pass 2; rays with more samples march the rest.  Any S gives the same
    gradient (S = 0: march twice like the reference) -- through the two-kernel
    backward (default with forward lists) and the one-kernel one alike."""
    import svox_t_amd.csrc as _C
    from svox_t_amd import synth
    monkeypatch.setattr(_C, "BWD_LIST_SAMPLES", list_samples)
    monkeypatch.setattr(_C, "BWD_GATHER", 0 if from_forward == "one_kernel" else 2)
    # "exact": lists from the forward, but two list walks instead of using the
    # forward's output for accum (SVOXT_BWD_EXACT=1)
    monkeypatch.setattr(_C, "BWD_EXACT", from_forward == "exact")
    c = Case(depth=6, K=28, data_format="SH9", width=96, height=96)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    g = synth.grad_output(c.Q, 4)
    if from_forward:        # lists recorded by the forward, replayed by the backward
        out = r(tree.features, c.rays_gpu(gpu))
        np.testing.assert_array_equal(out.detach().cpu().numpy(),
                                      O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts()))
        out.backward(g.to(gpu))
    else:                   # stand-alone backward: records in its own first pass
        from svox_t_amd.renderer import _rays_spec_from_rays
        tree.features.grad = _C.volume_render_backward(tree._spec(tree.features), _rays_spec_from_rays(c.rays_gpu(gpu)),
                                                       r._get_options(), g.to(gpu))
    want, abs_sum = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs=True)
    assert_grads_close(tree.features.grad.cpu().numpy(), want, abs_sum)


@pytest.mark.parametrize("name", ["d5_sh9", "d5_rgba4", "d5_sh4_world", "d5_sh1"])
def test_two_kernel_backward_and_repeated_backward(gpu, name, monkeypatch):
    """The two-kernel backward rewrites the forward's lists: a second backward over the
    same graph must still be right (it falls back to marching), and both kernels' routes
    agree with the one-kernel backward to the float-accumulation tolerance."""
    import svox_t_amd.csrc as _C
    from svox_t_amd import synth
    c = Case(**CASES[name])
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    g = synth.grad_output(c.Q, 4)
    want, abs_sum = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs=True)
    monkeypatch.setattr(_C, "BWD_GATHER", 2)                 # also without the image declaration
    out = r(tree.features, c.rays_gpu(gpu))
    out.backward(g.to(gpu), retain_graph=True)
    first = tree.features.grad.clone()
    assert_grads_close(first.cpu().numpy(), want, abs_sum)
    tree.features.grad = None
    out.backward(g.to(gpu))                                   # lists consumed: re-march
    assert_grads_close(tree.features.grad.cpu().numpy(), want, abs_sum)
    # image tiles: the merge works per 8x8 tile when the batch is declared an image
    tree.features.grad = None
    side = int(round(c.Q ** 0.5))
    out = r(tree.features, c.rays_gpu(gpu), image_shape=(side, side))
    out.backward(g.to(gpu))
    assert_grads_close(tree.features.grad.cpu().numpy(), want, abs_sum)
    monkeypatch.setattr(_C, "BWD_GATHER", 0)
    tree.features.grad = None
    r(tree.features, c.rays_gpu(gpu)).backward(g.to(gpu))
    assert_grads_close(tree.features.grad.cpu().numpy(), want, abs_sum)


@pytest.mark.parametrize("fmt,K,depth,side", [("SH16", 49, 5, 64), ("SH25", 76, 5, 56), ("SH16", 49, 6, 96), ("SH25", 76, 6, 96)])
def test_wide_sh_rows_take_the_per_tile_backward(gpu, fmt, K, depth, side, monkeypatch):
    """SH16 / SH25 (rows of 49 / 76 floats) on image batches: grad_fused_kernel over the hand-over the recording
    forward left (r03; per ray these rows cost 49 / 76 float atomics per sample) -- exact arithmetic: held to the
    tight scale like the SH9 route; the same through the kept gradient scratch (stride 64 / 80 floats), and with
    lists of 8 samples so that rays overflow into the tail-only launch."""
    import svox_t_amd.csrc as _C
    from svox_t_amd import synth
    c = Case(depth=depth, K=K, data_format=fmt, width=side, height=side)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    g = synth.grad_output(c.Q, 4)
    want, abs_sum, tight = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs="both")
    fwd_want = O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts())
    for cap in (None, 8):
        if cap is not None:
            monkeypatch.setattr(_C, "BWD_LIST_SAMPLES", cap)
        for rep in range(2):                                   # (the second step runs over the kept scratch)
            tree.features.grad = None
            out = r(tree.features, c.rays_gpu(gpu), image_shape=(side, side))
            np.testing.assert_array_equal(out.detach().cpu().numpy(), fwd_want)
            out.backward(g.to(gpu))
            assert _C.LAST_ROUTE["backward"].startswith("grad_fused_kernel<EXACT>"), _C.LAST_ROUTE
            got = tree.features.grad.cpu().numpy()
            assert_grads_close(got, want, tight)
            assert np.all(got[abs_sum == 0] == 0)
    # not declared an image: the per-ray backward, as before
    tree.features.grad = None
    r(tree.features, c.rays_gpu(gpu)).backward(g.to(gpu))
    assert _C.LAST_ROUTE["backward"].startswith("render_bwd_kernel"), _C.LAST_ROUTE
    assert_grads_close(tree.features.grad.cpu().numpy(), want, tight)


@pytest.mark.parametrize("knob,val", [("FWD_SPLIT", "0"), ("FWD_OVERLAP", False), ("LIST_POOL", False), ("SIGMA_MASK", False),
                                      ("GRAD_SCRATCH", False)])
def test_wide_sh_rows_under_the_forwards_other_arrangements(gpu, knob, val, monkeypatch):
    """SH16's per-tile backward over whatever the recording forward was: one kernel (hand-over in lane-major lines,
    terms_state 2), two launches, dense lists, no sigma bitmask, no kept gradient scratch -- same image bits, same
    gradient on the tight scale."""
    import svox_t_amd.csrc as _C
    from svox_t_amd import synth
    monkeypatch.setattr(_C, knob, val)
    _C.invalidate_caches()
    c = Case(depth=5, K=49, data_format="SH16", width=64, height=64)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    g = synth.grad_output(c.Q, 4)
    out = r(tree.features, c.rays_gpu(gpu), image_shape=(64, 64))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts()))
    out.backward(g.to(gpu))
    assert _C.LAST_ROUTE["backward"].startswith("grad_fused_kernel<EXACT>"), _C.LAST_ROUTE
    assert ("fwd_roles_kernel" in _C.LAST_ROUTE["forward"]) == (knob == "GRAD_SCRATCH"), _C.LAST_ROUTE
    want, abs_sum, tight = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs="both")
    assert_grads_close(tree.features.grad.cpu().numpy(), want, tight)


@pytest.mark.parametrize("fmt,K", [("SH9", 28), ("SH16", 49), ("RGBA", 4), ("RGBA", 16), ("SG9", 28)])
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_views_through_the_image_routes(gpu, fmt, K, seed):
    """Seeded views around (and partly off) the cube -- azimuth, elevation, distance, image size with ragged last
    tiles, feature seed -- through the routes an image takes (one-launch forward with tiles finished by their march,
    per-tile backward that leaves empty tiles at once): pixels bit-exact, gradient on the tight scale, for every
    payload family with a fast kernel."""
    import svox_t_amd.csrc as _C
    from svox_t_amd import synth
    rng = np.random.default_rng(100 * seed + K)
    W, H = int(rng.integers(5, 12)) * 8, int(rng.integers(5, 12)) * 8
    c = Case(depth=5, K=K, data_format=fmt, width=W, height=H, seed=seed)
    pose = synth.camera_pose(azimuth_deg=float(rng.uniform(0, 360)), elevation_deg=float(rng.uniform(-60, 60)),
                             radius=float(rng.uniform(1.1, 2.6)), center=np.array([0.5, 0.5, 0.5]) + rng.uniform(-0.25, 0.25, 3))
    o, d, v = (t.numpy() for t in synth.pinhole_rays(W, H, c2w=pose))
    lobes = None
    if fmt.startswith("SG"):
        g = torch.Generator().manual_seed(seed)
        lobes = torch.cat([torch.rand(c.basis_dim, 1, generator=g) * 4 + 0.5,
                           torch.nn.functional.normalize(torch.randn(c.basis_dim, 3, generator=g), dim=-1)], -1).contiguous()
    tree = svox.N3Tree.from_arrays(c.st.child, c.st.data, c.st.parent_depth, c.features, data_format=fmt,
                                   extra_data=lobes, device=gpu)
    ot = O.Tree(c.features.numpy(), c.st.data, c.st.child, offset=tree.offset.cpu().numpy(), scaling=tree.invradius.cpu().numpy(),
                extra=None if lobes is None else lobes.numpy())
    opt = O.make_options(format=c.format, basis_dim=c.basis_dim)
    r = svox.VolumeRenderer(tree)
    rays = svox.Rays(*(torch.from_numpy(a).to(gpu) for a in (o, d, v)))
    out = r(tree.features, rays, image_shape=(H, W))
    want = O.volume_render(ot, o, d, v, opt)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
    g = synth.grad_output(W * H, want.shape[1], seed=seed)
    out.backward(g.to(gpu))
    gw, ab, tight = O.volume_render_backward(ot, o, d, v, opt, g.numpy(), want_abs="both")
    assert_grads_close(tree.features.grad.cpu().numpy(), gw, tight)
    assert np.all(tree.features.grad.cpu().numpy()[ab == 0] == 0)


def test_wide_sh_rows_in_ray_order(gpu):
    """SH16 on a shuffled batch that is no image, rendered in svoxt_ray_order's order (sort_rays=True): the lists are
    walked per tile of that order -- grad_fused_kernel over the forward's hand-over, rays.order in both kernels -- and
    every ray's pixel and every gradient entry is what the oracle says."""
    import svox_t_amd.csrc as _C
    from svox_t_amd import synth
    c = Case(depth=5, K=49, data_format="SH16", width=72, height=64)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    o, d, v = c.rays_np()
    shuffle = np.random.default_rng(3).permutation(len(o))[: len(o) - 21]            # (a ragged last tile)
    o, d, v = o[shuffle].copy(), d[shuffle].copy(), v[shuffle].copy()
    rays = svox.Rays(*(torch.from_numpy(x).to(gpu) for x in (o, d, v)))
    g = synth.grad_output(len(o), 4, seed=5)
    out = r(tree.features, rays, sort_rays=True)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.volume_render(c.oracle_tree(), o, d, v, c.oracle_opts()))
    out.backward(g.to(gpu))
    assert _C.LAST_ROUTE["backward"].startswith("grad_fused_kernel<EXACT>"), _C.LAST_ROUTE
    want, abs_sum, tight = O.volume_render_backward(c.oracle_tree(), o, d, v, c.oracle_opts(), g.numpy(), want_abs="both")
    assert_grads_close(tree.features.grad.cpu().numpy(), want, tight)


@pytest.mark.parametrize("name", ["d5_sh9", "d5_sh4_world"])
def test_kept_gradient_scratch_changes_nothing(gpu, name, monkeypatch):
    """The padded gradient buffer kept between steps (svoxt_compact_rows_clear leaves it zeroed, the next backward
    skips its fill: SVOXT_LISTS_GRAD_ZEROED): step after step the gradient of a backward that fills its own buffer;
    another upstream gradient in between leaves nothing behind; a scratch left dirty (a call that failed between the
    atomics and the compaction) is filled again; a buffer of another shape gets its own scratch."""
    import svox_t_amd.csrc as _C
    from svox_t_amd import synth
    c = Case(**CASES[name])
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    side = int(round(c.Q ** 0.5))
    g = synth.grad_output(c.Q, 4).to(gpu)
    want, abs_sum = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.cpu().numpy(), want_abs=True)

    def step(gg):
        tree.features.grad = None
        r(tree.features, c.rays_gpu(gpu), image_shape=(side, side)).backward(gg)
        return tree.features.grad.clone()

    monkeypatch.setattr(_C, "GRAD_SCRATCH", False)
    _C.invalidate_caches()
    plain = step(g)
    assert not _C._GRAD_SCRATCH
    assert_grads_close(plain.cpu().numpy(), want, abs_sum)
    monkeypatch.setattr(_C, "GRAD_SCRATCH", True)
    scale = plain.abs().max().item()
    for it in range(3):
        if it == 1:
            step(5.0 * g + 1.0)                                  # something else through the same scratch
        got = step(g)
        assert len(_C._GRAD_SCRATCH) == 1
        (buf, clean), = _C._GRAD_SCRATCH.values()
        assert clean and buf.shape[1] % 16 == 0 and not buf.any().item()        # left zeroed, pad columns included
        assert (got - plain).abs().max().item() <= 1e-6 * scale                  # float atomics' order aside
    ent = next(iter(_C._GRAD_SCRATCH.values()))
    ent[0].fill_(3.0)
    ent[1] = False                                               # as a failed call leaves it
    assert (step(g) - plain).abs().max().item() <= 1e-6 * scale
    _C.invalidate_caches()
    assert not _C._GRAD_SCRATCH


@pytest.mark.parametrize("shape", [(64, 64), (40, 72), (60, 64)])      # last: H not a multiple of 8 -> hint ignored
def test_image_tile_hint_changes_nothing_but_the_lane_assignment(gpu, shape):
    H, W = shape
    c = Case(depth=5, K=28, data_format="SH9", width=W, height=H)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    rays = c.rays_gpu(gpu)
    want = O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts())
    out = r(tree.features, rays, image_shape=(H, W))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
    from svox_t_amd import synth
    g = synth.grad_output(c.Q, 4)
    out.backward(g.to(gpu))
    gw, ab = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs=True)
    assert_grads_close(tree.features.grad.cpu().numpy(), gw, ab)
    with torch.no_grad():
        d = r.render_depth(tree.features, rays, image_shape=(H, W)).cpu().numpy()
        a = r.opacity_render(tree.features, rays, image_shape=(H, W)).cpu().numpy()
    np.testing.assert_array_equal(d, O.render_depth(c.oracle_tree(), *c.rays_np(), c.oracle_opts()))
    np.testing.assert_array_equal(a, O.opacity_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts()))


@pytest.mark.parametrize("kind", ["identical", "zoomed", "wide"])
def test_two_kernel_backward_extreme_coherence(gpu, kind):
    """The per-tile merge at its extremes: every ray of a tile hitting the same leaves
    (64 records per feature row), a camera zoomed into a few leaves, and a wide view in
    which most records of a tile are different rows."""
    from svox_t_amd import synth
    c = Case(depth=6, K=28, data_format="SH9", width=64, height=64)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    if kind == "identical":
        o = c.origins[2080:2081].repeat(4096, 1)
        d = c.dirs[2080:2081].repeat(4096, 1)
    elif kind == "zoomed":
        o, d, _ = synth.pinhole_rays(64, 64, fx=1111.111 * 64 / 800.0 * 40.0)
    else:
        o, d, _ = synth.pinhole_rays(64, 64, fx=1111.111 * 64 / 800.0 * 0.35)
    v = d.clone()
    rays = svox.Rays(o.contiguous().to(gpu), d.contiguous().to(gpu), v.contiguous().to(gpu))
    g = synth.grad_output(4096, 4)
    out = r(tree.features, rays, image_shape=(64, 64))
    want = O.volume_render(c.oracle_tree(), o.numpy(), d.numpy(), v.numpy(), c.oracle_opts())
    np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
    assert (want[:, 3] > 0.1).mean() > (0.02 if kind == "wide" else 0.2)
    out.backward(g.to(gpu))
    gw, ab = O.volume_render_backward(c.oracle_tree(), o.numpy(), d.numpy(), v.numpy(), c.oracle_opts(), g.numpy(), want_abs=True)
    assert_grads_close(tree.features.grad.cpu().numpy(), gw, ab)



@pytest.mark.parametrize("name", ["d5_rgba4", "d5_sh9", "d5_sh4_world", "d4_sh16", "d5_rgba8", "d5_rgba16", "d6_rgba32"])
def test_two_kernel_forward_equals_one_kernel_forward(name, gpu, monkeypatch):
    """FWD_SPLIT = "1" (SVOXT_FWD_SPLIT; march_rec_kernel + shade_tile_kernel + tail launch, through
    svoxt_volume_render_fwd_ws and svoxt_volume_render_fwd_record) against the one-kernel forward:
    outputs bit-identical with thresholds 0 and 1e-2, with lists long enough and too short (the tail
    launch finishes the rays), as an image and as a plain batch; lists recorded either way give the
    backward the same gradient."""
    import svox_t_amd.csrc as _C
    from svox_t_amd.renderer import _rays_spec_from_rays
    c = Case(**CASES[name])
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    spec = tree._spec(tree.features)
    rays = c.rays_gpu(gpu)
    W = CASES[name]["width"]
    for rs in (_rays_spec_from_rays(rays), _rays_spec_from_rays(rays, (c.Q // W, W))):
        for fast in (False, True):
            opt = r._get_options(fast=fast)
            monkeypatch.setattr(_C, "FWD_SPLIT", "0")
            rs.need_grad = False                 # forwards nobody differentiates
            want = _C.volume_render(spec, rs, opt)
            monkeypatch.setattr(_C, "FWD_SPLIT", "1")
            for S in (96, 8):
                monkeypatch.setattr(_C, "FWD_LIST_SAMPLES", S)
                assert torch.equal(_C.volume_render(spec, rs, opt), want), (name, fast, S)
        opt = r._get_options()
        monkeypatch.setattr(_C, "FWD_SPLIT", "0")
        rs.need_grad = False
        want = _C.volume_render(spec, rs, opt)
        g = torch.randn_like(want)
        grads = {}
        for split in ("0", "1"):
            monkeypatch.setattr(_C, "FWD_SPLIT", split)
            for S in (96, 8):
                monkeypatch.setattr(_C, "BWD_LIST_SAMPLES", S)
                out, lists = _C.volume_render(spec, rs, opt, record=True)
                assert lists is not None and torch.equal(out, want)
                grads[(split, S)] = _C.volume_render_backward(spec, rs, opt, g, lists=lists, fwd_output=out)
        ref = grads[("0", 96)]
        scale = ref.abs().max().item()
        for k, v in grads.items():
            assert (v - ref).abs().max().item() <= 1e-5 * scale, k


@pytest.mark.parametrize("name", ["d5_rgba4", "d5_sh9", "d5_sh4_world", "d5_sh1"])
@pytest.mark.parametrize("shape", ["image", "ragged"])
def test_march_and_shade_in_one_launch_change_nothing(name, shape, gpu, monkeypatch):
    """fwd_roles_kernel (include/svoxt.h, svoxt_sample_lists.tile_state): one grid whose first workgroups
    march and whose others shade each tile as soon as its march has published it.  Against the two
    launches it replaces: pixels, lists (through the backward that replays them) and hand-over bit for bit
    -- recording forward and scratch forward, lists long enough and too short (tail launch), an image and
    a batch whose last tile is ragged -- and against the oracle."""
    import svox_t_amd.csrc as _C
    from svox_t_amd import synth
    from svox_t_amd.renderer import _rays_spec_from_rays
    c = Case(**CASES[name])
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    opt = r._get_options()
    spec = tree._spec(tree.features)
    rays = c.rays_gpu(gpu)
    W, H = CASES[name]["width"], CASES[name]["height"]
    if shape == "ragged":
        n = c.Q - 37                                  # not a multiple of 64, not an image
        rays = svox.Rays(*(t[:n].contiguous() for t in rays))
        rs = _rays_spec_from_rays(rays)
        o, d, v = (a[:n] for a in c.rays_np())
    else:
        n = c.Q
        rs = _rays_spec_from_rays(rays, (H, W))
        o, d, v = c.rays_np()
    want = O.volume_render(c.oracle_tree(), o, d, v, c.oracle_opts())
    g = synth.grad_output(n, want.shape[1]).to(gpu)
    monkeypatch.setattr(_C, "FWD_SPLIT", "1")          # the march + shade forward also where it is not the default
    res = {}
    for overlap in (False, True):
        monkeypatch.setattr(_C, "FWD_OVERLAP", overlap)
        for S in (96, 8):
            monkeypatch.setattr(_C, "BWD_LIST_SAMPLES", S)
            monkeypatch.setattr(_C, "FWD_LIST_SAMPLES", S)
            rs.need_grad = False
            plain = _C.volume_render(spec, rs, opt)                      # scratch lists
            assert ("fwd_roles_kernel" in _C.LAST_ROUTE["forward"]) == overlap, _C.LAST_ROUTE
            out, lists = _C.volume_render(spec, rs, opt, record=True)    # recorded lists (+ the backward's hand-over)
            assert ("fwd_roles_kernel" in _C.LAST_ROUTE["forward"]) == overlap, _C.LAST_ROUTE
            np.testing.assert_array_equal(plain.cpu().numpy(), want)
            np.testing.assert_array_equal(out.cpu().numpy(), want)
            grad = _C.volume_render_backward(spec, rs, opt, g, lists=lists)
            res[(overlap, S)] = (lists.aux.clone(), None if lists.terms is None else lists.terms.clone(), lists.terms_state, grad)
    for S in (96, 8):
        a, b = res[(False, S)], res[(True, S)]
        assert torch.equal(a[0][:, :3], b[0][:, :3])               # counts / overflow, resume points, final transmittance
        assert a[2] == b[2]
        scale = a[3].abs().max().item()
        assert (a[3] - b[3]).abs().max().item() <= 1e-6 * scale      # the same contributions, float atomics' order aside


class _ReferenceShapedFunction(torch.autograd.Function):
    """What the reference's own svox_t/renderer.py:60-77 does with whatever module it found as
    `svox_t.csrc` -- two calls, the same spec objects, nothing else (route B of INTEGRATION.md)."""

    @staticmethod
    def forward(ctx, data, C, tree, rays, opt):
        out = C.volume_render(tree, rays, opt)
        ctx.C, ctx.tree, ctx.rays, ctx.opt = C, tree, rays, opt
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return ctx.C.volume_render_backward(ctx.tree, ctx.rays, ctx.opt, grad_out.contiguous()), None, None, None, None


@pytest.mark.parametrize("name", ["d5_rgba4", "d5_sh9"])
def test_plain_calls_hand_the_forwards_lists_to_the_backward(name, gpu, monkeypatch):
    """The two plain calls of the reference's autograd function (no record=, no lists=): the forward
    leaves its sample lists -- and, for a batch that is not an image, the coherent ray order -- on
    the rays spec; the backward of the same spec objects picks them up.  Same output (bit for bit)
    and same gradient as the explicit route and as the oracle; a backward that finds nothing
    (second backward, another rays spec, features edited in between) marches and agrees too."""
    import svox_t_amd.csrc as _C
    from svox_t_amd.renderer import _rays_spec_from_rays
    c = Case(**CASES[name])
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    rays = c.rays_gpu(gpu)
    opt = r._get_options()
    from svox_t_amd import synth
    g = synth.grad_output(c.Q, 4).to(gpu)
    want_out = O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts())
    want, absum = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.cpu().numpy(), want_abs=True)
    monkeypatch.setattr(_C, "SORT_RAYS_MIN", 1024)          # these small batches take the sorted route too
    monkeypatch.setattr(_C, "DETECT_IMAGES", False)         # (the batch IS a 64 x 64 pinhole image: r05 would recognise it and walk it in tiles)
    for sort in (None, False):
        f = tree.features.detach().clone().requires_grad_(True)
        rs = _C.RaysSpec()                                   # as the reference fills it: three tensors
        rs.origins, rs.dirs, rs.vdirs = rays.origins, rays.dirs, rays.viewdirs
        rs.sort = sort
        spec = tree._spec(f)
        out = _ReferenceShapedFunction.apply(f, _C, spec, rs, opt)
        np.testing.assert_array_equal(out.detach().cpu().numpy(), want_out)
        plan = rs._svoxt_plan
        # (the coherent order travels as RaysSpec.order: the kernels read it, nothing is gathered or scattered)
        assert plan is not None and plan.lists is not None and plan.perm is None
        assert (getattr(plan.rays, "order", None) is not None) == (sort is None)
        out.backward(g, retain_graph=True)
        assert rs._svoxt_plan is None                        # taken
        assert_grads_close(f.grad.cpu().numpy(), want, absum)
        first = f.grad.clone()
        f.grad = None
        out.backward(g)                                      # nothing left behind: the backward marches
        assert_grads_close(f.grad.cpu().numpy(), want, absum)
        assert (f.grad - first).abs().max().item() <= 1e-5 * first.abs().max().item()
    # features edited in place between forward and backward: the plan is refused (the march reads the new values)
    f = tree.features.detach().clone().requires_grad_(True)
    rs = _rays_spec_from_rays(rays)
    spec = tree._spec(f)
    out = _ReferenceShapedFunction.apply(f, _C, spec, rs, opt)
    with torch.no_grad():
        f.mul_(1.0)
    assert _C._take_plan("volume", spec, rs, opt) is None
    # a forward nobody differentiates leaves nothing
    rs2 = _rays_spec_from_rays(rays)
    with torch.no_grad():
        out2 = r(tree.features, rays)
    np.testing.assert_array_equal(out2.cpu().numpy(), want_out)
    _C.volume_render(tree._spec(tree.features.detach()), rs2, opt)
    assert rs2._svoxt_plan is None


@pytest.mark.parametrize("K", [8, 16, 32])
@pytest.mark.parametrize("cap", [None, 8, 24])     # 8 / 24: most lists overflow -> the tails by the tail-only launch
def test_wide_rows_backward_routes(gpu, K, cap, monkeypatch):
    """RGBA-style rows of 8 / 16 / 32 floats: the per-tile backward of an image (grad_wide_kernel:
    sigmoids once per record in sweep 1 and once per distinct row of a tile window in sweep 2, merged
    atomic rows), the per-ray one-sigmoid-pass backward (render_bwd_kernel<ONEPASS>) and the plain
    list replay all give the reference's contributions: every entry within 1e-5 of the TIGHT scale
    (accum priced by the reference's own sequential addends)."""
    import svox_t_amd.csrc as _C
    from svox_t_amd import synth
    if cap is not None:
        monkeypatch.setattr(_C, "BWD_LIST_SAMPLES", cap)
    c = Case(depth=5, K=K, data_format="RGBA", width=64, height=56)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    g = synth.grad_output(c.Q, K)
    want, _, tight = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs="both")
    routes = []
    for gather, terms, expect in ((1, True, "grad_wide_kernel"), (0, True, "render_bwd_kernel<ONEPASS>"),
                                  (0, False, "render_bwd_kernel<REPLAY>")):
        monkeypatch.setattr(_C, "BWD_GATHER", gather)
        monkeypatch.setattr(_C, "BWD_TERMS", terms)
        tree.features.grad = None
        out = r(tree.features, c.rays_gpu(gpu), image_shape=(56, 64))
        out.backward(g.to(gpu))
        assert _C.LAST_ROUTE["backward"].startswith(expect), _C.LAST_ROUTE
        assert_grads_close(tree.features.grad.cpu().numpy(), want, tight, what=expect)
        routes.append(tree.features.grad.clone())
    # a second backward over the same forward (the lists are not rewritten by these routes)
    monkeypatch.setattr(_C, "BWD_GATHER", 1)
    monkeypatch.setattr(_C, "BWD_TERMS", True)
    tree.features.grad = None
    out = r(tree.features, c.rays_gpu(gpu), image_shape=(56, 64))
    out.backward(g.to(gpu), retain_graph=True)
    tree.features.grad = None
    out.backward(g.to(gpu))
    assert_grads_close(tree.features.grad.cpu().numpy(), want, tight, what="second backward")


@pytest.mark.parametrize("name", ["d5_rgba8", "d5_rgba16", "d5_sh9"])
def test_sigma_bitmask_changes_nothing(name, gpu, monkeypatch):
    """svoxt_tree.sigma_mask (one bit per feature row: sigma > sigma_thresh) lets the two-kernel forward's
    march skip its sigma gather.  A pure cache: forwards with it -- scratch lists (no stop rule needed at
    stop_thresh = 0), recorded lists and the backward that replays them, a non-zero sigma_thresh -- equal
    the oracle bit for bit, and it follows an in-place change of the features."""
    import svox_t_amd.csrc as _C
    from svox_t_amd import synth
    from svox_t_amd.renderer import _rays_spec_from_rays
    monkeypatch.setattr(_C, "SIGMA_MASK", True)
    monkeypatch.setattr(_C, "FWD_SPLIT", "1")               # the march + shade forward for every payload that has one
    c = Case(**CASES[name])
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    rays = c.rays_gpu(gpu)
    shape = (CASES[name]["height"], CASES[name]["width"])
    ot = c.oracle_tree()
    assert (ot.features[:, -1] <= 0).any() and (ot.features[:, -1] > 0).any()      # rows on both sides of the threshold
    want = O.volume_render(ot, *c.rays_np(), c.oracle_opts())
    _C._SIGMA_CACHE.clear()
    with torch.no_grad():
        got = r(tree.features, rays, image_shape=shape)
    assert "march" in _C.LAST_ROUTE["forward"]
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    assert len(_C._SIGMA_CACHE) == 0             # nothing derived from the features' content is kept by default ...
    tree.static_features = True                  # ... only for a table its owner declares static
    with torch.no_grad():
        got = r(tree.features, rays, image_shape=shape)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    assert len(_C._SIGMA_CACHE) == 1
    # recording forward + backward (the mask is rebuilt, not taken from the cache)
    g = synth.grad_output(c.Q, want.shape[1])
    out = r(tree.features, rays, image_shape=shape)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
    out.backward(g.to(gpu))
    gw, _, tight = O.volume_render_backward(ot, *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs="both")
    assert_grads_close(tree.features.grad.cpu().numpy(), gw, tight)
    # sigma_thresh > 0 with stop_thresh = 0: the bits are taken against that threshold
    opt = r._get_options()
    opt.sigma_thresh = 0.5
    oo = O.make_options(format=c.format, basis_dim=c.basis_dim, sigma_thresh=0.5, stop_thresh=0.0)
    with torch.no_grad():
        got = _C.volume_render(tree._spec(tree.features), _rays_spec_from_rays(rays, shape), opt)
    np.testing.assert_array_equal(got.cpu().numpy(), O.volume_render(ot, *c.rays_np(), oo))
    # an in-place change of the features: the cached mask must not survive it
    with torch.no_grad():
        tree.features[:, -1] *= -1.0
        got = r(tree.features, rays, image_shape=shape)
    f2 = ot.features.copy()
    f2[:, -1] *= -1.0
    want2 = O.volume_render(O.Tree(f2, ot.data, ot.child, offset=ot.offset, scaling=ot.scaling), *c.rays_np(), c.oracle_opts())
    np.testing.assert_array_equal(got.cpu().numpy(), want2)
    assert not np.array_equal(want2, want)
    # ... nor a storage swap, which leaves the version counter alone
    with torch.no_grad():
        v0 = tree.features._version
        tree.features.data = torch.from_numpy(ot.features.copy()).to(gpu)        # back to the original values
        assert tree.features._version == v0
        got = r(tree.features, rays, image_shape=shape)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    # a write through `.data` -- the reference's own idiom, and what an optimizer's `p.data.add_()` does --
    # bumps no version counter: a tree that is NOT declared static gets a fresh mask every forward
    tree.static_features = False
    with torch.no_grad():
        v0 = tree.features._version
        tree.features.data[:, -1] *= -1.0
        assert tree.features._version == v0
        got = r(tree.features, rays, image_shape=shape)
    np.testing.assert_array_equal(got.cpu().numpy(), want2)
    # ... and for a static one, invalidate_caches() is the way to say so
    tree.static_features = True
    with torch.no_grad():
        got = r(tree.features, rays, image_shape=shape)              # caches the mask of the flipped table
        tree.features.data[:, -1] *= -1.0                            # back to the original values
        _C.invalidate_caches(tree.features)
        got = r(tree.features, rays, image_shape=shape)
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_plan_is_not_replayed_for_other_rays_trees_or_operators(gpu):
    """The forward of a plain call leaves its sample lists on the rays spec (svox_t_amd.csrc._Plan).  The
    reference's backward re-marches whatever the spec objects hold when it runs (renderer.py:64-72), so a
    plan must be dropped -- and the backward march -- when anything the lists were recorded for has changed:
    the rays (in-place edit between forward and backward), the operator, the topology."""
    import svox_t_amd.csrc as _C
    from svox_t_amd import synth
    from svox_t_amd.renderer import _rays_spec_from_rays
    c = Case(depth=5, K=28, data_format="SH9", width=64, height=64)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    opt = r._get_options()
    g = synth.grad_output(c.Q, 4)
    rays = c.rays_gpu(gpu)
    rs = _rays_spec_from_rays(rays, (64, 64))
    rs.need_grad = True
    spec = tree._spec(tree.features)
    _C.volume_render(spec, rs, opt)
    assert rs._svoxt_plan is not None and rs._svoxt_plan.lists is not None
    # the caller moves the camera in place between forward and backward
    shift = torch.tensor([0.02, -0.01, 0.015], device=gpu)
    rays.origins.add_(shift)
    got = _C.volume_render_backward(spec, rs, opt, g.to(gpu))
    assert _C.LAST_ROUTE["backward"].startswith("render_bwd_kernel (marches")
    o2 = c.origins.numpy() + shift.cpu().numpy()
    want, _, tight = O.volume_render_backward(c.oracle_tree(), o2, c.dirs.numpy(), c.vdirs.numpy(), c.oracle_opts(),
                                              g.numpy(), want_abs="both")
    assert_grads_close(got.cpu().numpy(), want, tight, what="re-marched with the edited rays")
    # a plan left by volume_render is not taken by opacity_render_backward (and is consumed by the attempt)
    _C.volume_render(spec, rs, opt)
    g1 = synth.grad_output(c.Q, 1, seed=3)
    got = _C.opacity_render_backward(spec, rs, opt, g1.to(gpu))
    assert rs._svoxt_plan is None
    want1, ab1 = O.volume_render_backward(c.oracle_tree(), o2, c.dirs.numpy(), c.vdirs.numpy(), c.oracle_opts(),
                                          g1.numpy(), want_abs=True)
    assert_grads_close(got.cpu().numpy(), want1, ab1, what="opacity backward after a volume forward")
    # topology edited between forward and backward (refine bumps child / data): the lists name stale rows
    _C.volume_render(spec, rs, opt)
    p = rs._svoxt_plan
    torch.autograd.graph.increment_version(tree.child)
    assert _C._take_plan("volume", spec, rs, opt) is None
    # unchanged everything: the plan is taken
    _C.volume_render(spec, rs, opt)
    assert _C._take_plan("volume", spec, rs, opt) is not None


def test_a_forward_without_a_backward_frees_its_lists_by_refcount(gpu):
    """the plan must not close a reference cycle through the autograd node (ctx -> rays spec ->
    plan -> out -> grad_fn -> ctx), or a forward that never gets a backward keeps its sample lists, its
    hand-over buffer and its output on the GPU until the cyclic collector happens to run."""
    import gc
    import weakref
    import svox_t_amd.csrc as _C
    from svox_t_amd.renderer import _VolumeRenderFunction, _rays_spec_from_rays
    c = Case(depth=5, K=28, data_format="SH9", width=64, height=64)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    gc.collect()
    gc.disable()
    try:
        rs = _rays_spec_from_rays(c.rays_gpu(gpu), (64, 64))
        rs.need_grad = True
        out = _VolumeRenderFunction.apply(tree.features, tree._spec(tree.features), rs, r._get_options())
        assert out.grad_fn is not None and rs._svoxt_plan is not None
        refs = [weakref.ref(rs._svoxt_plan.lists.rec), weakref.ref(rs._svoxt_plan.lists.terms), weakref.ref(rs._svoxt_plan)]
        del out, rs
        assert all(w() is None for w in refs), [w() is None for w in refs]
    finally:
        gc.enable()


@pytest.mark.parametrize("name", ["d5_sh9", "d5_rgba8", "d5_rgba4"])
@pytest.mark.parametrize("pool", ["dense", "dry", "tiny"])
def test_sample_list_pool_runs_dry_or_is_dense(name, pool, gpu, monkeypatch):
    """Sample lists are blocks of 8 positions x 64 rays handed out from a pool sized by what earlier
    forwards used.  A pool that is too small -- here 32 blocks (one per sub-pool: most tiles get none) or
    64 -- only means that rays stop recording early, exactly as at the per-ray cap: the tail launches take
    over, the forward stays bit-identical and the backward within tolerance.  Dense lists
    (SVOXT_LIST_POOL=0: every ray owns its slots) give the same."""
    import svox_t_amd.csrc as _C
    from svox_t_amd import synth
    if pool == "dense":
        monkeypatch.setattr(_C, "LIST_POOL", False)
    else:
        monkeypatch.setattr(_C, "_pool_blocks_for", lambda tiles, S, kind="record": 32 if pool == "dry" else 64)
    c = Case(**CASES[name])
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    shape = (CASES[name]["height"], CASES[name]["width"])
    want = O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts())
    g = synth.grad_output(c.Q, want.shape[1])
    gw, _, tight = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs="both")
    for image in (True, False):
        tree.features.grad = None
        out = r(tree.features, c.rays_gpu(gpu), image_shape=shape if image else None)
        np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
        out.backward(g.to(gpu))
        assert_grads_close(tree.features.grad.cpu().numpy(), gw, tight, what=f"{pool}, image={image}")
    # scratch lists of a forward nobody differentiates (the two-kernel forward where it is the default)
    with torch.no_grad():
        np.testing.assert_array_equal(r(tree.features, c.rays_gpu(gpu), image_shape=shape).cpu().numpy(), want)
