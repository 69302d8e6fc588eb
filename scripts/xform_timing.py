#!/usr/bin/env python3
"""Render with per-leaf view rotations (transformation_matrices) on the headline
tree: cost of the generic kernels that serve it, next to the plain render.

    python scripts/xform_timing.py [--reps 10]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
import svox_t_amd.csrc as _C  # noqa: E402          # noqa: E402
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
    M = st.n_features
    feats = synth.shell_features(M, 28).to(dev).requires_grad_(True)
    tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats.detach(), data_format="SH9", device=dev)
    r = svox.VolumeRenderer(tree)
    W = H = 800
    o, d, v = synth.pinhole_rays(W, H)
    rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
    rng = np.random.default_rng(0)
    J, B = 24, 4
    joints = torch.eye(4).repeat(J, 1, 1)
    for k in range(J):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        joints[k, :3, :3] = torch.from_numpy(q.astype(np.float32))
    sw = torch.from_numpy(rng.random((M, B)).astype(np.float32)).to(dev)
    ji = torch.from_numpy(rng.integers(0, J, size=(M, B)).astype(np.int32)).to(dev)
    mats = svox.blend_transformation_matrix(joints.to(dev), sw, ji)
    gout = torch.randn(W * H, 4, device=dev)

    def fwd(x):
        with torch.no_grad():
            return r(feats, rays, transformation_matrices=x, image_shape=(H, W))

    def step(x):
        feats.grad = None
        r(feats, rays, transformation_matrices=x, image_shape=(H, W)).backward(gout)

    Q = W * H
    print(f"blend_transformation_matrix ({M} rows, {J} joints, {B} bound): "
          f"{timed(lambda: svox.blend_transformation_matrix(joints.to(dev), sw, ji), a.reps):.3f} ms")
    for name, fn in (("plain forward", lambda: fwd(None)), ("forward with view rotations", lambda: fwd(mats)),
                     ("plain forward + backward", lambda: step(None)),
                     ("forward + backward with view rotations", lambda: step(mats))):
        ms = timed(fn, a.reps)
        print(f"{name:42s} {ms:8.3f} ms  {Q / ms / 1e3:7.1f} Mrays/s")
        if "backward" in name:
            print(f"    route: {_C.LAST_ROUTE['forward']} | {_C.LAST_ROUTE['backward']}")


if __name__ == "__main__":
    main()
