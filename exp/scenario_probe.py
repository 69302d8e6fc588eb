"""Sequences of calls a training script makes, timed per step against the plain loop: alternating trees, alternating image sizes,
camera route with a new pose every step, an evaluation forward between training steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd import synth
dev = torch.device("cuda:0")
def make(depth, K, fmt):
    st = synth.shell_tree(depth)
    feats = synth.shell_features(st.n_features, K).to(dev)
    tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format=fmt, device=dev)
    return tree, svox.VolumeRenderer(tree), feats.clone().requires_grad_(True)
def rays_of(size, az=30.0):
    o, d, v = synth.pinhole_rays(size, size, c2w=synth.camera_pose(azimuth_deg=az))
    return svox.Rays(o.to(dev), d.to(dev), v.to(dev))
def loop(fn, reps=60, warm=20):
    for i in range(warm): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
t8, r8, p8 = make(8, 28, "SH9")
t7, r7, p7 = make(7, 28, "SH9")
R800, R400, R800b = rays_of(800), rays_of(400), rays_of(800, 75.0)
def train(r, p, rays, **kw):
    out = r(p, rays, **kw); out.backward(torch.ones_like(out)); p.grad = None
base8 = loop(lambda i: train(r8, p8, R800))
base7 = loop(lambda i: train(r7, p7, R800))
base400 = loop(lambda i: train(r8, p8, R400))
print(f"plain loops: depth 8 / 800^2 {base8:.3f} ms, depth 7 / 800^2 {base7:.3f}, depth 8 / 400^2 {base400:.3f}", flush=True)
x = loop(lambda i: train(*((r8, p8) if i % 2 == 0 else (r7, p7)), R800))
print(f"two trees alternating: {x:.3f} ms per step (mean of the two plain loops {0.5 * (base8 + base7):.3f})", flush=True)
x = loop(lambda i: train(r8, p8, R800 if i % 2 == 0 else R400))
print(f"two image sizes alternating: {x:.3f} (mean {0.5 * (base8 + base400):.3f})", flush=True)
x = loop(lambda i: train(r8, p8, R800 if i % 2 == 0 else R800b))
print(f"two cameras alternating: {x:.3f} (plain {base8:.3f})", flush=True)
def with_eval(i):
    train(r8, p8, R800)
    if i % 4 == 3:
        with torch.no_grad(): r8(p8, R800b, fast=True)
ev = loop(lambda i: (lambda: [r8(p8, R800b, fast=True)])() if False else None, reps=1, warm=0)
with torch.no_grad():
    e = loop(lambda i: r8(p8, R800b, fast=True))
x = loop(with_eval)
print(f"an evaluation forward (fast=True, no grad) after every 4th step: {x:.3f} per step (plain {base8:.3f} + a quarter of {e:.3f} = {base8 + e / 4:.3f})", flush=True)
pose = torch.from_numpy(synth.camera_pose(azimuth_deg=40.0)).float().to(dev)
def cam(i):
    c2w = pose.clone()
    out = r8.render_persp(p8, c2w, width=800, height=800, fx=1111.111); out.backward(torch.ones_like(out)); p8.grad = None
x = loop(cam)
print(f"render_persp with a new pose tensor every step: {x:.3f} (plain ray route {base8:.3f})", flush=True)
x = loop(lambda i: train(r8, p8, R800, fast=True))
print(f"fast=True training: {x:.3f}", flush=True)
print("pool hints:", {k: (v[0], v[2], v[3]) for k, v in _C._POOL_HINT.items()})
