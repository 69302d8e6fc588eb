#!/usr/bin/env python3
"""One headline training step (forward + backward through VolumeRenderer / autograd, 800 x 800, depth-8 SH9, image
hint) captured into a HIP graph (torch.cuda.CUDAGraph) and replayed, against the same step launched eagerly: what the
launch gaps between the step's dozen kernels are worth.  Static inputs as a graph needs them (a trainer would copy its
batch into them before every replay).  Not used by bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svox_t_amd as svox
from svox_t_amd import synth

dev = torch.device("cuda:0")
W = H = 800
st = synth.shell_tree(8)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, synth.shell_features(st.n_features, 28), data_format="SH9", device=dev)
r = svox.VolumeRenderer(tree)
o, d, v = synth.pinhole_rays(W, H, c2w=synth.camera_pose(azimuth_deg=30.0))
rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
gout = synth.grad_output(W * H, 4).to(dev)
f = tree.features

def step():
    f.grad = None
    out = r(f, rays, image_shape=(H, W))
    out.backward(gout)
    return out

def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for _ in range(35):
    step()
eager = timed(step)
ref = f.grad.clone()
print(f"eager  : {eager:.4f} ms per step = {W * H / eager / 1e3:.1f} Mrays/s", flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
f.grad = None
with torch.cuda.graph(g):
    out = r(f, rays, image_shape=(H, W))
    out.backward(gout)
replay = timed(g.replay)
err = (f.grad - ref).abs().max().item() / ref.abs().max().item()
print(f"graph  : {replay:.4f} ms per step = {W * H / replay / 1e3:.1f} Mrays/s; max |grad - eager grad| / max |grad| = {err:.2e}", flush=True)
