"""Small random-ray batches (what a NeRF-style loop draws): forward + backward per step through the reference's API."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
from svox_t_amd import synth
import svox_t_amd.csrc as _C

dev = torch.device("cuda:0")
st = synth.shell_tree(8)
feats = synth.shell_features(st.n_features, 28).to(dev)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format="SH9", device=dev)
r = svox.VolumeRenderer(tree)
cams = [[t.to(dev) for t in synth.pinhole_rays(800, 800, c2w=synth.camera_pose(azimuth_deg=a))] for a in (20.0, 50.0, 110.0, 200.0, 260.0, 300.0, 330.0, 80.0)]
O_, D_, V_ = (torch.cat([c[i] for c in cams]) for i in range(3))
p = feats.clone().requires_grad_(True)
g = torch.Generator(device=dev).manual_seed(1)
for smin, Q in [(m, q) for m in (16384, 512) for q in (1024, 4096, 8192, 16384, 65536, 262144) if m == 16384 or q < 16384]:
    _C.SORT_RAYS_MIN = smin
    def step(fresh=True):
        idx = torch.randint(0, O_.shape[0], (Q,), device=dev, generator=g)
        rays = svox.Rays(O_[idx], D_[idx], V_[idx])
        out = r(p, rays); out.backward(torch.ones_like(out)); p.grad = None
    for _ in range(6): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 40 * 1e3
    # GPU time alone of the two calls (events around them)
    idx = torch.randint(0, O_.shape[0], (Q,), device=dev, generator=g)
    rays = svox.Rays(O_[idx], D_[idx], V_[idx]); go = torch.ones((Q, 4), device=dev)
    for _ in range(3):
        out = r(p, rays); out.backward(go); p.grad = None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        out = r(p, rays); out.backward(go); p.grad = None
    e1.record(); torch.cuda.synchronize()
    print(f"sort from {smin:5d} rays on, Q {Q:7d}: loop {ms:.3f} ms/step ({Q / ms / 1e3:7.1f} Mrays/s)   the two calls alone, same batch: {e0.elapsed_time(e1) / 20:.3f} ms   "
          f"{_C.LAST_ROUTE['forward'][:40]} | {_C.LAST_ROUTE['backward'][:40]}", flush=True)
