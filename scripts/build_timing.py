#!/usr/bin/env python3
"""Time the octree build of a point cloud: the fused HIP pipeline
(N3Tree.build_from_points) against the reference's call sequence through this
package (tree[points].refine() x (depth-1) + construct_tree).

    python scripts/build_timing.py [--points 500000] [--depth 8] [--reps 20]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox          # noqa: E402


def cloud(n, seed=0):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return (0.5 + 0.35 * d * (1 + 0.01 * rng.normal(size=(n, 1)))).astype(np.float32)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=500000)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    p = torch.from_numpy(cloud(a.points)).to(dev)
    tree = svox.N3Tree(N=2, data_dim=4, map_location=dev)

    def fused():
        tree.build_from_points(p, a.depth)

    def stepwise():
        t = svox.N3Tree(N=2, data_dim=4, map_location=dev)
        for _ in range(a.depth - 1):
            t[p].refine()
        t.construct_tree(p)
        return t

    ms_f = timed(fused, a.reps)
    n = tree.n_internal
    ms_s = timed(stepwise, max(3, a.reps // 4))
    print(f"points {a.points} depth {a.depth}: n_internal {n}")
    print(f"  fused HIP pipeline   {ms_f:8.3f} ms  ({a.points / ms_f / 1e3:.1f} Mpoints/s)")
    print(f"  step-by-step route   {ms_s:8.3f} ms  ({a.points / ms_s / 1e3:.1f} Mpoints/s)  x{ms_s / ms_f:.1f}")


if __name__ == "__main__":
    main()
