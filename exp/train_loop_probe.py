"""A training loop as a user of the reference's API writes it: a Parameter updated in place every step, a new image batch (new tensors,
one of several cameras) every step, the loss's gradient from torch.  Per-step time against the bench's kernels-only step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
from svox_t_amd import synth
import svox_t_amd.csrc as _C

dev = torch.device("cuda:0")
depth, K, fmt, size = 8, 28, "SH9", 800
st = synth.shell_tree(depth)
feats = synth.shell_features(st.n_features, K).to(dev)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format=fmt, device=dev)
r = svox.VolumeRenderer(tree)
cams = [[t.to(dev) for t in synth.pinhole_rays(size, size, c2w=synth.camera_pose(azimuth_deg=a))] for a in (20.0, 50.0, 110.0, 200.0)]
target = torch.rand((size * size, 3), device=dev)
p = torch.nn.Parameter(feats.clone())

def run(opt_kind, fresh, reps=60, every_sync=False):
    opt = torch.optim.SGD([p], lr=1e-3) if opt_kind == "sgd" else torch.optim.Adam([p], lr=1e-3, fused=True) if opt_kind == "adam" else None
    fixed = [svox.Rays(*c) for c in cams]
    def step(i):
        c = cams[i % len(cams)]
        rays = svox.Rays(c[0].clone(), c[1].clone(), c[2].clone()) if fresh else fixed[i % len(cams)]
        out = r(p, rays)
        loss = ((out[:, :3] - target) ** 2).mean()
        if opt is not None:
            opt.zero_grad(set_to_none=True)
        else:
            p.grad = None
        loss.backward()
        if opt is not None:
            opt.step()
        if every_sync:
            loss.item()
    for i in range(8): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps): step(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

for opt_kind in ("none", "sgd", "adam"):
    for fresh in (False, True):
        print(f"optimizer {opt_kind:5} {'new tensors every step' if fresh else 'four fixed batches    '}: {run(opt_kind, fresh):.3f} ms/step"
              f"   (with loss.item() every step: {run(opt_kind, fresh, every_sync=True):.3f})", flush=True)
print("routes:", _C.LAST_ROUTE["forward"], "|", _C.LAST_ROUTE["backward"])
