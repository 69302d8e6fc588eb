#!/usr/bin/env python3
"""GPU box: the two-kernel forward (march + per-tile shade) against the one-kernel forward.

Bit-equality of outputs and of the recorded sample lists, then timings of both, on the
headline workload (D=8 SH9 800x800), with and without the image hint, with thresholds
(`fast`), and on shuffled rays.  Usage: python scripts/fwd_split_check.py [d8|d9]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd import synth
from svox_t_amd.renderer import _rays_spec_from_rays

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "d8"
depth, K, fmt, W, H = (8, 28, "SH9", 800, 800) if which == "d8" else (9, 32, "RGBA", 1024, 1024)
st = synth.shell_tree(depth)
feats = synth.shell_features(st.n_features, K)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format=fmt, device=dev)
r = svox.VolumeRenderer(tree)
o, d, v = synth.pinhole_rays(W, H)
rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
spec = tree._spec(tree.features)


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def thread_of_ray(rs, Q):
    """launch thread that handles ray q (svoxt_device.h ray_of_thread inverted): 8x8 tiles with the hint"""
    w, h = rs.image_width, rs.image_height
    if w * h != Q or w % 8 or h % 8:
        return torch.arange(Q)
    q = torch.arange(Q)
    y, x = q // w, q % w
    return ((y // 8) * (w // 8) + x // 8) * 64 + (y % 8) * 8 + x % 8


def both(name, rs, opt, record):
    res = {}
    for split in ("0", "1"):
        _C.FWD_SPLIT = split
        _C.FWD_LIST_SAMPLES = 96 if split == "1" else 0
        rs.need_grad = False          # record=False means a forward nobody differentiates
        x = _C.volume_render(spec, rs, opt, record=record)
        torch.cuda.synchronize()
        res[split] = x
        ms = timeit(lambda: _C.volume_render(spec, rs, opt, record=record))
        print(f"{name:28s} split={split} record={int(record)}  {ms:.4f} ms", flush=True)
    a, b = res["0"], res["1"]
    if record:
        (oa, la), (ob, lb) = a, b
        assert (la is None) == (lb is None)
        if la is not None:
            auxa, auxb = la.aux.cpu(), lb.aux.cpu()
            assert torch.equal(auxa[:, :3], auxb[:, :3]), "aux differs"
            n = (auxa[:, 0] & 0x7fffffff).long()
            S = la.S
            # logical records [thread, k] through the block table (rec[block][lane][8]; dense: block = tile * S/8 + b)
            def logical(l):
                tiles, nb = l.tiles, l.S // 8
                tab = l.blocktab.view(tiles, nb).long() if l.pooled else torch.arange(tiles * nb, device=dev).view(tiles, nb)
                r = l.rec.view(-1, 64, 8, 2)[tab.clamp(min=0)]                 # [tiles, nb, 64, 8, 2]
                return r.permute(0, 2, 1, 3, 4).reshape(tiles * 64, l.S, 2), (tab >= 0)
            ra, oka = logical(la)
            rb, okb = logical(lb)
            nt = torch.zeros(la.tiles * 64, dtype=torch.long, device=dev)
            nt[thread_of_ray(rs, n.numel()).to(dev)] = n.to(dev)
            mask = torch.arange(S, device=dev)[None, :] < nt[:, None]
            assert torch.equal(ra[mask], rb[mask]), "recorded samples differ"
            del ra, rb
            print(f"{'':28s} lists equal: {int(n.sum())} records, {int((auxa[:, 0] < 0).sum())} overflowed rays, max {int(n.max())}")
    else:
        oa, ob = a, b
    same = torch.equal(oa, ob)
    print(f"{'':28s} outputs bit-equal: {same}; max |diff| {float((oa - ob).abs().max()):.3e}", flush=True)
    assert same


opt = r._get_options()
fast = r._get_options(fast=True)
rs = _rays_spec_from_rays(rays)
rsh = _rays_spec_from_rays(rays, (H, W))
both("image hint", rsh, opt, False)
both("image hint", rsh, opt, True)
both("no hint", rs, opt, False)
both("image hint, fast", rsh, fast, False)
p = torch.randperm(W * H, device=dev)
rp = svox.Rays(rays.origins[p].contiguous(), rays.dirs[p].contiguous(), rays.viewdirs[p].contiguous())
both("shuffled", _rays_spec_from_rays(rp), opt, False)
# short lists: most rays overflow and finish in the tail launch
_C.BWD_LIST_SAMPLES = 8
both("image hint, S=8", rsh, opt, True)
_C.BWD_LIST_SAMPLES = 96
print("OK")
