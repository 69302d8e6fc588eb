#!/usr/bin/env python3
"""opacity_render forward / backward on the headline tree (800x800 pinhole image,
depth-8 shell): the list-walk + per-tile merge backward next to the marching one.

    python scripts/opacity_timing.py [--reps 10]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox          # noqa: E402
import svox_t_amd.csrc as _C       # noqa: E402
from svox_t_amd import synth       # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    st = synth.shell_tree(8)
    feats = synth.shell_features(st.n_features, 28).to(dev).requires_grad_(True)
    tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats.detach(), data_format="SH9", device=dev)
    r = svox.VolumeRenderer(tree)
    W = H = 800
    o, d, v = synth.pinhole_rays(W, H)
    rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
    perm = torch.randperm(W * H, device=dev)
    shuffled = svox.Rays(rays.origins[perm].contiguous(), rays.dirs[perm].contiguous(), rays.viewdirs[perm].contiguous())
    gout = torch.randn(W * H, 1, device=dev)

    def fwd(rr, shape):
        with torch.no_grad():
            return r.opacity_render(feats, rr, image_shape=shape)

    def step(rr, shape):
        feats.grad = None
        r.opacity_render(feats, rr, image_shape=shape).backward(gout)

    for name, rr, shape in (("image (8x8 tiles)", rays, (H, W)), ("shuffled rays", shuffled, None)):
        f = timed(lambda: fwd(rr, shape), a.reps)
        _C.BWD_LIST_SAMPLES = 64
        s1 = timed(lambda: step(rr, shape), a.reps)
        _C.BWD_LIST_SAMPLES = 0
        s0 = timed(lambda: step(rr, shape), a.reps)
        _C.BWD_LIST_SAMPLES = 64
        print(f"{name:20s} fwd {f:.3f} ms   fwd+bwd lists {s1:.3f} ms   fwd+bwd marching {s0:.3f} ms")


if __name__ == "__main__":
    main()
