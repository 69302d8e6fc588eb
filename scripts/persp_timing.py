#!/usr/bin/env python3
"""Image-mode render (rays generated in the kernels) against the ray-batch render
of the same camera, on the headline tree (depth 8, SH9, 800 x 800).

    python scripts/persp_timing.py [--reps 30]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox          # noqa: E402
import svox_t_amd.csrc as _C       # noqa: E402
from svox_t_amd import synth       # noqa: E402
from svox_t_amd.renderer import pinhole_rays   # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--size", type=int, default=800)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    st = synth.shell_tree(8)
    feats = synth.shell_features(st.n_features, 28).to(dev)
    tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format="SH9", device=dev)
    r = svox.VolumeRenderer(tree)
    W = H = a.size
    fx = 1111.111 * W / 800.0
    c2w = torch.from_numpy(synth.camera_pose().astype(np.float32)).to(dev)
    o, d, v = pinhole_rays(c2w, W, H, fx, fx)
    rays = svox.Rays(o, d, v)
    gout = torch.randn(H, W, 4, device=dev)
    f = feats.clone().requires_grad_(True)

    def img_fwd():
        with torch.no_grad():
            return r.render_persp(feats, c2w, width=W, height=H, fx=fx)

    def batch_fwd():
        with torch.no_grad():
            return r(feats, rays, image_shape=(H, W))

    def gen_and_batch_fwd():
        with torch.no_grad():
            oo, dd, vv = pinhole_rays(c2w, W, H, fx, fx)
            return r(feats, svox.Rays(oo, dd, vv), image_shape=(H, W))

    def img_step():
        f.grad = None
        r.render_persp(f, c2w, width=W, height=H, fx=fx).backward(gout)

    def batch_step():
        f.grad = None
        r(f, rays, image_shape=(H, W)).backward(gout.view(-1, 4))

    assert torch.equal(img_fwd().view(-1, 4), batch_fwd())
    Q = W * H
    for name, fn in (("image mode forward (rays generated in-kernel)", img_fwd),
                     ("ray-batch forward, rays resident", batch_fwd),
                     ("torch ray generation + ray-batch forward", gen_and_batch_fwd),
                     ("image mode forward + backward", img_step),
                     ("ray-batch forward + backward, rays resident", batch_step)):
        ms = timed(fn, a.reps)
        print(f"{name:48s} {ms:7.3f} ms  {Q / ms / 1e3:7.1f} Mrays/s")
        if "backward" in name:
            print(f"    route: {_C.LAST_ROUTE['forward']} | {_C.LAST_ROUTE['backward']}")


if __name__ == "__main__":
    main()
