#!/usr/bin/env python3
"""Ray batches that are not images: cost of rendering them as given and in svoxt_ray_order's
order (sort + gathers included), 640 000 rays on the headline tree.

    python scripts/ray_order_timing.py [--reps 10]
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
from svox_t_amd.renderer import _rays_spec_from_rays  # noqa: E402


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

    def batch(kind):
        if kind == "one camera, shuffled":
            o, d, v = synth.pinhole_rays(W, H)
        elif kind == "one camera, row-major":
            o, d, v = synth.pinhole_rays(W, H)
            return svox.Rays(o.to(dev), d.to(dev), v.to(dev))
        else:
            parts = [synth.pinhole_rays(W, H, c2w=synth.camera_pose(azimuth_deg=30.0 + 45.0 * k)) for k in range(8)]
            o, d, v = (torch.cat([p[i] for p in parts]) for i in range(3))
            sel = torch.randperm(o.shape[0])[: W * H]
            o, d, v = o[sel], d[sel], v[sel]
        p = torch.randperm(o.shape[0])
        return svox.Rays(o[p].contiguous().to(dev), d[p].contiguous().to(dev), v[p].contiguous().to(dev))

    for kind in ("one camera, shuffled", "8 cameras, random rays", "one camera, row-major"):
        rays = batch(kind)
        gout = torch.randn(W * H, 4, device=dev)
        t_sort = timed(lambda: _C.ray_order(tree._spec(feats), _rays_spec_from_rays(rays), r._get_options()), a.reps)
        print(f"{kind}: svoxt_ray_order {t_sort:.3f} ms")
        for sort_rays in (False, True):
            for gather in (0, 2):
                _C.BWD_GATHER = gather

                def fwd():
                    with torch.no_grad():
                        r(feats, rays, sort_rays=sort_rays)

                def step():
                    feats.grad = None
                    r(feats, rays, sort_rays=sort_rays).backward(gout)
                print(f"  sort_rays={sort_rays!s:5s} two-kernel backward={'on ' if gather == 2 else 'off'}  "
                      f"fwd {timed(fwd, a.reps):.3f} ms   fwd+bwd {timed(step, a.reps):.3f} ms", flush=True)
        _C.BWD_GATHER = 1


if __name__ == "__main__":
    main()
