#!/usr/bin/env python3
"""Forward / forward+backward time of an SG25 payload (K = 76) on the depth-8 shell tree at 800 x 800: the kernels
that keep a ray's 25 basis values in registers (render_fwd_kernel / render_bwd_kernel<..., LOBES>) -- r03; SG / ASG
payloads used to take the generic kernels, which SVOXT_LIB pointed at an older build still shows.  No oracle here."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svox_t_amd as svox
from svox_t_amd import synth

dev = torch.device("cuda:0")
B = int(os.environ.get("LOBES", "25"))
kind = os.environ.get("KIND", "SG")
depth, W, H = 8, 800, 800
K = 3 * B + 1
st = synth.shell_tree(depth)
feats = synth.shell_features(st.n_features, K)
g = torch.Generator().manual_seed(7)
if kind == "SG":
    lobes = torch.cat([torch.rand(B, 1, generator=g) * 4 + 0.5,
                       torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)], -1).contiguous()
else:
    fr = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0]
    lobes = torch.cat([torch.rand(B, 2, generator=g) * 3 + 0.3, fr.reshape(B, 9)], -1).contiguous()
if kind == "SH":                 # (for comparison: an SH payload of the same width)
    lobes = None
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format=f"{kind}{B}", extra_data=lobes, device=dev)
r = svox.VolumeRenderer(tree)
o, d, v = synth.pinhole_rays(W, H, c2w=synth.camera_pose(azimuth_deg=30.0))
rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
gout = synth.grad_output(W * H, 4).to(dev)
f = tree.features

def fwd():
    with torch.no_grad():
        return r(f, rays, image_shape=(H, W))

def both():
    f.grad = None
    r(f, rays, image_shape=(H, W)).backward(gout)

for name, fn in (("forward", fwd), ("forward+backward", both)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{kind}{B} (K = {K}) 800x800 depth-8 {name}: {ms:.3f} ms = {W * H / ms / 1e3:.1f} Mrays/s", flush=True)
import svox_t_amd.csrc as _C
print("   routes:", _C.LAST_ROUTE.get("forward"), "|", _C.LAST_ROUTE.get("backward"), flush=True)
