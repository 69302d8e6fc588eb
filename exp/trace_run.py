#!/usr/bin/env python3
"""Run with SVOXT_LIB=exp/libsvoxt_trace.so: timelines of the workgroups of the forward,
list-walk and merge kernels of one headline step."""
import ctypes, os, sys
import numpy as np
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
o, d, v = synth.pinhole_rays(W, H)
rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
gout = synth.grad_output(W * H, 4).to(dev)
lib = _C._lib
buf = np.zeros((65536, 3), dtype=np.uint64)


def step():
    feats.grad = None
    r(feats, rays, image_shape=(H, W)).backward(gout)


def report(name, slot):
    lib.svoxt_trace_read(slot, buf.ctypes.data_as(ctypes.c_void_p), 1)
    t = buf.astype(np.int64); t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / 100.0          # us
    end = (t[:, 1] - t0) / 100.0
    dur = end - start
    print(f"--- {name}: kernel span {end.max():.1f} us; last start {start.max():.1f} us; sum of durations {dur.sum() / 1e3:.1f} ms")
    print("duration percentiles (us): " + "  ".join(f"p{p}={np.percentile(dur, p):.1f}" for p in (10, 50, 90, 99, 100)))
    grid = np.arange(0, end.max(), 10.0)
    print("resident workgroups every 10 us: " + " ".join(str(np.sum((start <= g) & (end > g))) for g in grid))
    for i in np.argsort(-dur)[:4]:
        print(f"  wg {i}: start {start[i]:.1f} dur {dur[i]:.1f} end {end[i]:.1f}")
    np.save(f"gpurun_out/trace_{name}.npy", t)


os.makedirs("gpurun_out", exist_ok=True)
for _ in range(3):
    step()
torch.cuda.synchronize()
for s in range(3):
    lib.svoxt_trace_read(s, buf.ctypes.data_as(ctypes.c_void_p), 1)
step()
torch.cuda.synchronize()
report("fwd", 0)
report("tail", 1)
report("fused", 2)
