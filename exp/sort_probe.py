#!/usr/bin/env python3
"""Probe: how much does ordering a shuffled ray batch by cube-entry Morton code recover?
(torch-only ordering, existing kernels; decides whether a native ray-ordering step pays)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd import synth

dev = torch.device("cuda:0")
st = synth.shell_tree(8)
feats = synth.shell_features(st.n_features, 28).to(dev).requires_grad_(True)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats.detach(), data_format="SH9", device=dev)
r = svox.VolumeRenderer(tree)
W = H = 800


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def part1by2(x):
    x = x & 0x3ff
    x = (x | (x << 16)) & 0x30000ff
    x = (x | (x << 8)) & 0x300f00f
    x = (x | (x << 4)) & 0x30c30c3
    x = (x | (x << 2)) & 0x9249249
    return x


def entry_key(o, d, bits=10):
    inv = 1.0 / d
    t1 = (0.0 - o) * inv
    t2 = (1.0 - o) * inv
    tmin = torch.minimum(t1, t2).amax(dim=1).clamp_min(0.0)
    tmax = torch.maximum(t1, t2).amin(dim=1)
    p = (o + tmin[:, None] * d).clamp(0, 1 - 1e-6)
    c = (p * (1 << bits)).to(torch.int64)
    key = part1by2(c[:, 0]) << 2 | part1by2(c[:, 1]) << 1 | part1by2(c[:, 2])
    key[tmax <= tmin] = 1 << 31
    return key


def _entry(o, d):
    inv = 1.0 / d
    t1 = (0.0 - o) * inv
    t2 = (1.0 - o) * inv
    tmin = torch.minimum(t1, t2).amax(dim=1).clamp_min(0.0)
    tmax = torch.maximum(t1, t2).amin(dim=1)
    p = (o + tmin[:, None] * d).clamp(0, 1 - 1e-6)
    return p, tmax <= tmin


def mixed_key(o, d, ebits, dbits, dir_major):
    """dir_major: [dir 3 x dbits][entry morton 3 x ebits]; else round-robin from the MSB"""
    p, miss = _entry(o, d)
    u = (d / d.norm(dim=1, keepdim=True) * 0.5 + 0.5).clamp(0, 1 - 1e-6)
    e = (p * (1 << ebits)).to(torch.int64)
    q = (u * (1 << dbits)).to(torch.int64)
    key = torch.zeros_like(e[:, 0])
    if dir_major:
        for b in range(dbits - 1, -1, -1):
            for a in range(3):
                key = key << 1 | ((q[:, a] >> b) & 1)
        for b in range(ebits - 1, -1, -1):
            for a in range(3):
                key = key << 1 | ((e[:, a] >> b) & 1)
    else:
        eb, db = ebits - 1, dbits - 1
        while eb >= 0 or db >= 0:
            if eb >= 0:
                for a in range(3):
                    key = key << 1 | ((e[:, a] >> eb) & 1)
                eb -= 1
            if db >= 0:
                for a in range(3):
                    key = key << 1 | ((q[:, a] >> db) & 1)
                db -= 1
    key[miss] = 1 << 40
    return key


KEYS = {
    "entry morton": lambda o, d: entry_key(o, d),
    "rr e6 d4": lambda o, d: mixed_key(o, d, 6, 4, False),
    "rr e7 d3": lambda o, d: mixed_key(o, d, 7, 3, False),
    "dir2|e8": lambda o, d: mixed_key(o, d, 8, 2, True),
    "dir3|e7": lambda o, d: mixed_key(o, d, 7, 3, True),
    "dir4|e6": lambda o, d: mixed_key(o, d, 6, 4, True),
}


def batch(kind):
    if kind == "one camera":
        o, d, v = synth.pinhole_rays(W, H)
    else:
        parts = [synth.pinhole_rays(W, H, c2w=synth.camera_pose(azimuth_deg=30.0 + 45.0 * k)) for k in range(8)]
        o, d, v = (torch.cat([p[i] for p in parts]) for i in range(3))
        sel = torch.randperm(o.shape[0])[: W * H]
        o, d, v = o[sel], d[sel], v[sel]
    perm = torch.randperm(o.shape[0])
    return o[perm].contiguous().to(dev), d[perm].contiguous().to(dev), v[perm].contiguous().to(dev)


for kind in ("one camera", "8 cameras"):
    o, d, v = batch(kind)
    gout = torch.randn(o.shape[0], 4, device=dev)
    for order in ["as given"] + list(KEYS):
        if order != "as given":
            idx = torch.sort(KEYS[order](o, d))[1]
            oo, dd, vv, gg = o[idx].contiguous(), d[idx].contiguous(), v[idx].contiguous(), gout[idx].contiguous()
        else:
            oo, dd, vv, gg = o, d, v, gout
        rays = svox.Rays(oo, dd, vv)
        res = []
        for g in (0, 2):
            _C.BWD_GATHER = g

            def fwd():
                with torch.no_grad():
                    r(feats, rays)

            def step():
                feats.grad = None
                r(feats, rays).backward(gg)
            res.append(f"gather={g}: fwd {timed(fwd):.3f} fwd+bwd {timed(step):.3f}")
        print(f"{kind:11s} {order:13s} " + "   ".join(res), flush=True)
