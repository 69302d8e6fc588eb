#!/usr/bin/env python3
"""Per-step times of the plain (no image hint) route: looks for steps that are far off the median."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svox_t_amd as svox, svox_t_amd.csrc as _C
from svox_t_amd import synth
dev = torch.device("cuda:0")
st = synth.shell_tree(8)
feats = synth.shell_features(st.n_features, 28)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format="SH9", device=dev)
r = svox.VolumeRenderer(tree)
o, d, v = synth.pinhole_rays(800, 800)
rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
g = synth.grad_output(640000, 4).to(dev)
for mode in ("forward", "forward+backward"):
    ts = []
    for i in range(200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "forward":
            with torch.no_grad():
                r(tree.features, rays)
        else:
            tree.features.grad = None
            r(tree.features, rays).backward(g)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    s = sorted(ts[20:])
    print(mode, "median %.3f ms  p90 %.3f  max %.3f  mean %.3f" % (s[len(s) // 2], s[int(len(s) * 0.9)], s[-1], sum(s) / len(s)),
          " slow steps (> 2x median):", [(i, round(t, 2)) for i, t in enumerate(ts) if i >= 20 and t > 2 * s[len(s) // 2]][:12])
