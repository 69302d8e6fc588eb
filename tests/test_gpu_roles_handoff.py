"""The march -> shade hand-over inside fwd_roles_kernel.

The forward of a 3-channel payload runs its march and its shade in ONE launch: marching wavefronts publish a tile
into the ready queue of their XCD, shading workgroups take tiles from the queue of the XCD they run on and read the
lists with sc1 loads.  That rests on measured cache behaviour (same L2, acknowledged stores, loads that bypass the
vector cache), not on the memory model, and on a dispatch order HIP does not promise.  What makes it safe whatever
the hardware does: every queue entry carries a checksum of the tile's lists, the consumer folds what it LOADED the
same way and leaves a tile whose checksum does not match -- or that it never got -- to the fallback launch.  These
tests make that rescue path do real work (consumers that drop tiles, give up at once, or read "stale" records) and
check every word it is responsible for; and run the production hand-over under uneven load with a second stream
hogging bandwidth, list buffers reused from step to step, inputs that change every step."""
import numpy as np
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from oracle import oracle as O
from svox_t_amd import synth
from svox_t_amd.renderer import _rays_spec_from_rays
from tests.util import Case

pytestmark = pytest.mark.gpu


def _ray_of(tiles, shape, Q, dev):
    """[tiles, 64] ray index of (tile, lane) -- ray_of_thread of svoxt_device.h -- or -1 past the batch.  (Row-major
    tile order: the cases of this file have feature tables that fit the Infinity Cache; larger ones walk their tiles
    in super-tiles: RaysDev.super_tiles.)"""
    t = torch.arange(tiles, device=dev)[:, None]
    lane = torch.arange(64, device=dev)[None, :]
    if shape is not None and shape[0] % 8 == 0 and shape[1] % 8 == 0 and shape[0] * shape[1] == Q:
        H, W = shape
        q = ((t // (W // 8)) * 8 + lane // 8) * W + (t % (W // 8)) * 8 + lane % 8
    else:
        q = t * 64 + lane
    return torch.where(q < Q, q, torch.full_like(q, -1))


def _canonical(lists, shape, Q):
    """What the lists SAY, independent of which pool block holds what: aux words 0..2 per ray, and per (tile, list
    position, lane) the record and the hand-over entry -- zero where the ray has no such record."""
    dev = lists.aux.device
    tiles, S = lists.tiles, lists.S
    aux = lists.aux[:, :3].clone()
    q = _ray_of(tiles, shape, Q, dev)
    n = torch.where(q >= 0, lists.aux[q.clamp(min=0), 0] & 0x7fffffff, torch.zeros_like(q))       # [tiles, 64]
    tab = lists.blocktab.view(tiles, S // 8).long()
    rec = lists.rec.view(-1, 64, 8, 2)[tab.clamp(min=0)]                      # [tiles, S/8, 64, 8, 2]
    rec = rec.permute(0, 1, 3, 2, 4).reshape(tiles, S, 64, 2)                 # [tiles, k, lane, 2]
    k = torch.arange(S, device=dev)[None, :, None]
    have = (k < n[:, None, :]) & (tab.repeat_interleave(8, dim=1)[:, :, None] >= 0)
    rec = torch.where(have[..., None], rec, torch.zeros_like(rec))
    terms = None
    if lists.terms is not None and lists.terms_state == 3:
        t4 = lists.terms.view(-1, 8, 64, 4)[tab.clamp(min=0)].reshape(tiles, S, 64, 4)          # position-major
        terms = torch.where(have[..., None], t4.view(torch.int32), torch.zeros_like(t4, dtype=torch.int32))
    return aux, rec, terms, int((n.max(dim=1).values > 0).sum())


def _counters(lists):
    """(shaded in the launch, checksum mismatches, workgroups that gave up, dropped by the test flag, shaded by the fallback)"""
    ts = lists.tile_state
    base = (lists.tiles + 1) // 2 * 2
    return tuple(int(v) + 1 for v in ts[base + 256: base + 261].cpu().tolist())


def _forward(spec, rs, opt):
    out, lists = _C.volume_render(spec, rs, opt, record=True)
    assert lists is not None and lists.terms is not None
    return out, lists


CASES = {
    "cfg3": (dict(depth=8, K=28, data_format="SH9", width=800, height=800), "image"),
    "d6_sh4_ragged": (dict(depth=6, K=13, data_format="SH4", width=203, height=77), "ragged"),
    "d5_rgba4_image": (dict(depth=5, K=4, data_format="RGBA", width=64, height=48), "image"),
}


@pytest.mark.parametrize("name", list(CASES))
def test_rescue_path_reproduces_pixels_lists_and_handover(name, gpu, monkeypatch, capsys):
    kw, kind = CASES[name]
    c = Case(**kw)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    opt = r._get_options()
    spec = tree._spec(tree.features)
    shape = (kw["height"], kw["width"]) if kind == "image" else None
    rs = _rays_spec_from_rays(c.rays_gpu(gpu), shape)
    rs.need_grad = False
    g = synth.grad_output(c.Q, 4).to(gpu)
    monkeypatch.setattr(_C, "FWD_SPLIT", "1")
    # the reference: march and shade as two launches (no queue, no hand-over inside a launch)
    monkeypatch.setattr(_C, "FWD_OVERLAP", False)
    out0, l0 = _forward(spec, rs, opt)
    assert "fwd_roles_kernel" not in _C.LAST_ROUTE["forward"]
    want = _canonical(l0, shape, c.Q)
    grad0 = _C.volume_render_backward(spec, rs, opt, g, lists=l0)
    np.testing.assert_array_equal(out0.cpu().numpy(), O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts()))
    busy = want[3]
    assert busy > 0
    monkeypatch.setattr(_C, "FWD_OVERLAP", True)
    D, N, S, A = _C.LISTS_TEST_DROP, _C.LISTS_TEST_NOPOLL, _C.LISTS_TEST_STALE, _C.LISTS_FWD_AGENT_FENCE
    report = []
    for flags in (0, D, N, S, D | S, D | N | S, A):
        monkeypatch.setattr(_C, "ROLES_FLAGS", flags)
        out, l1 = _forward(spec, rs, opt)
        assert "fwd_roles_kernel" in _C.LAST_ROUTE["forward"], _C.LAST_ROUTE
        assert l1.flags & 0xf00 == flags
        shaded, mismatch, gaveup, dropped, fallback = ctr = _counters(l1)
        report.append((flags, ctr))
        got = _canonical(l1, shape, c.Q)
        assert torch.equal(out, out0), f"flags {flags:#x}: pixels"
        assert torch.equal(got[0], want[0]), f"flags {flags:#x}: aux"
        assert torch.equal(got[1], want[1]), f"flags {flags:#x}: records"
        assert torch.equal(got[2], want[2]), f"flags {flags:#x}: hand-over"
        grad = _C.volume_render_backward(spec, rs, opt, g, lists=l1)
        assert (grad - grad0).abs().max().item() <= 1e-6 * grad0.abs().max().item(), f"flags {flags:#x}: gradient"
        # every tile with a sample is shaded exactly once that counts: in the launch (checksum equal) or by the fallback
        assert shaded + fallback == busy, (flags, ctr, busy)
        if flags & D and not flags & N:                         # (with N a small grid's consumers all give up before they take a tile)
            assert dropped > 0 and fallback >= dropped, ctr
        if flags & S and not flags & N:
            assert mismatch > 0 and fallback >= mismatch, ctr
        assert fallback >= dropped + mismatch, ctr              # (+ what consumers that gave up, or never came, left behind; an
                                                                #  entry nobody waited for may also name a tile without samples)
        if not flags & (D | N | S):
            assert mismatch == 0, f"flags {flags:#x}: a hand-over read something its march had not written: {ctr}"
    with capsys.disabled():
        for flags, ctr in report:
            print(f"\n[roles hand-over, {name}] flags {flags:#06x}: shaded in launch {ctr[0]}, checksum mismatches {ctr[1]}, "
                  f"gave up {ctr[2]}, dropped {ctr[3]}, fallback shaded {ctr[4]} (tiles with samples: {busy})")


def test_handoff_under_uneven_load_every_word(gpu, monkeypatch, capsys):
    """The production hand-over (no test flag) while a second stream hogs the memory system, the list buffers reused
    from step to step (the caching allocator hands the same addresses back, still holding the previous step's records:
    what a stale line would show), inputs that change every step, the consumers' vector caches warm from the step
    before: every word of pixels, list headers, records and hand-over against the two-launch forward of the same
    inputs; the checksum must never have fired."""
    c = Case(depth=8, K=28, data_format="SH9", width=800, height=800)
    shape = (800, 800)
    opt = svox.VolumeRenderer(c.tree(gpu))._get_options()
    # four inputs: two cameras x two feature tables (the second with other signs of sigma: other lists)
    f2 = c.features.clone()
    f2[:, -1] = torch.where(torch.rand(f2.shape[0], generator=torch.Generator().manual_seed(7)) < 0.3, -f2[:, -1].abs(), f2[:, -1].abs())
    trees = [c.tree(gpu), svox.N3Tree.from_arrays(c.st.child, c.st.data, c.st.parent_depth, f2, data_format="SH9", device=gpu)]
    cams = []
    for az in (30.0, 75.0):
        o, d, v = synth.pinhole_rays(800, 800, c2w=synth.camera_pose(azimuth_deg=az))
        cams.append(svox.Rays(o.to(gpu), d.to(gpu), v.to(gpu)))
    monkeypatch.setattr(_C, "FWD_SPLIT", "1")
    inputs, refs = [], []
    monkeypatch.setattr(_C, "FWD_OVERLAP", False)
    for tr in trees:
        for rays in cams:
            spec = tr._spec(tr.features)
            rs = _rays_spec_from_rays(rays, shape)
            rs.need_grad = False
            out, l0 = _forward(spec, rs, opt)
            refs.append((out.clone(),) + _canonical(l0, shape, c.Q)[:3])
            inputs.append((spec, rs))
            del l0
    monkeypatch.setattr(_C, "FWD_OVERLAP", True)
    monkeypatch.setattr(_C, "ROLES_FLAGS", 0)
    hog = torch.cuda.Stream()
    a = torch.empty((256 << 20,), dtype=torch.float32, device=gpu)            # 1 GiB each: past the Infinity Cache
    b = torch.empty_like(a)
    main = torch.cuda.current_stream()
    totals = np.zeros(5, dtype=np.int64)
    steps = 24
    for it in range(steps):
        with torch.cuda.stream(hog):
            for _ in range(3):
                b.copy_(a, non_blocking=True)                                 # ~0.4 ms of streaming each, beside the forward
        spec, rs = inputs[it % 4]
        out, l1 = _forward(spec, rs, opt)
        assert "fwd_roles_kernel" in _C.LAST_ROUTE["forward"]
        ref = refs[it % 4]
        got = _canonical(l1, shape, c.Q)
        assert torch.equal(out, ref[0]), f"step {it}: pixels"
        assert torch.equal(got[0], ref[1]), f"step {it}: aux"
        assert torch.equal(got[1], ref[2]), f"step {it}: records"
        assert torch.equal(got[2], ref[3]), f"step {it}: hand-over"
        totals += np.array(_counters(l1))
        del l1, out                                                           # (the next step's lists take these addresses)
    hog.synchronize()
    main.synchronize()
    with capsys.disabled():
        print(f"\n[roles hand-over under load] {steps} steps beside a streaming copy: shaded in launch {totals[0]}, checksum "
              f"mismatches {totals[1]}, gave up {totals[2]}, fallback shaded {totals[4]}")
    assert totals[1] == 0, "a shading workgroup read lists its march had not written (checksum mismatch)"
