#!/usr/bin/env python3
"""render_depth (rt_kernel.cu:782-834) and the weight-accumulating forward on the two benchmark geometries: 800 x 800 rays on the depth-8 SH9 tree and
BASELINE configs[3]'s second output, 1024 x 1024 rays on the depth-9 tree with 32-float rows.

    python scripts/depth_timing.py [--reps 30]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox          # noqa: E402
from svox_t_amd import synth       # noqa: E402


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
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for depth, K, fmt, size in ((8, 28, "SH9", 800), (9, 32, "RGBA", 1024)):
        st = synth.shell_tree(depth)
        feats = synth.shell_features(st.n_features, K).to(dev)
        tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format=fmt, device=dev)
        r = svox.VolumeRenderer(tree)
        o, d, v = synth.pinhole_rays(size, size)
        rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
        Q = size * size
        for name, fn in (("render_depth, ray batch as given", lambda: r.render_depth(feats, rays)),
                         ("render_depth, declared an image (8 x 8 tiles)", lambda: r.render_depth(feats, rays, image_shape=(size, size)))):
            try:
                ms = timed(fn, a.reps)
            except TypeError:
                continue
            print(f"depth {depth}, K {K}, {size} x {size}: {name:46s} {ms:7.3f} ms  {Q / ms / 1e3:8.1f} Mrays/s", flush=True)
        # the forward that also adds every sample's compositing weight to its leaf slot (tree._weight_accum,
        # rt_kernel.cu:266-267, 309-311; svox.py:948-969): SURVEY 8 f3, one float atomic per sample
        with torch.no_grad(), tree.accumulate_weights() as acc:
            ms = timed(lambda: r(feats, rays), a.reps)
            total = float(acc.value.double().sum())
        with torch.no_grad():
            ms0 = timed(lambda: r(feats, rays), a.reps)
        print(f"depth {depth}, K {K}, {size} x {size}: {'volume_render + per-leaf weight accumulation':46s} {ms:7.3f} ms  {Q / ms / 1e3:8.1f} Mrays/s"
              f"   (the same call without: {ms0:.3f} ms; sum of weights over {a.reps + 3} calls {total:.1f})", flush=True)
        if os.environ.get("SVOXT_TIMING_ROUTES"):
            import svox_t_amd.csrc as _C
            print("   route:", _C.LAST_ROUTE["forward"], flush=True)


if __name__ == "__main__":
    main()
