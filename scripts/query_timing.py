#!/usr/bin/env python3
"""query_vertical and its backward (svox_kernel.cu:45-94, 240-324, 380-402) on the two benchmark trees: 1 M points drawn
uniformly from the cube (most land in empty leaves of the shell trees) and 1 M points on the shell (every one in a
leaf with features).

    python scripts/query_timing.py [--reps 20]
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
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    P = 1 << 20
    g = torch.Generator().manual_seed(0)
    uniform = torch.rand(P, 3, generator=g)
    u = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    shell = 0.5 + 0.35 * u                                       # the shell is 0.34 <= |x - 0.5| <= 0.36
    for depth, K, fmt in ((8, 28, "SH9"), (9, 32, "RGBA")):
        st = synth.shell_tree(depth)
        feats = synth.shell_features(st.n_features, K).to(dev).requires_grad_(True)
        tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats.detach(), data_format=fmt, device=dev)
        for name, pts in (("uniform in the cube", uniform), ("on the shell", shell)):
            p = pts.to(dev)
            gout = torch.randn(P, K, device=dev)

            def fwd():
                with torch.no_grad():
                    return tree(feats, p, world=False)

            def step():
                feats.grad = None
                tree(feats, p, world=False).backward(gout)

            f, s = timed(fwd, a.reps), timed(step, a.reps)
            print(f"depth {depth}, K {K}, {P} points {name:20s} query {f:7.3f} ms ({P / f / 1e3:8.1f} Mpoints/s)   "
                  f"query + backward {s:7.3f} ms ({P / s / 1e3:8.1f} Mpoints/s)", flush=True)


if __name__ == "__main__":
    main()
