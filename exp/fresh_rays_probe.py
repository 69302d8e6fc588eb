"""Per-step cost of batches that arrive as NEW tensors every step (what a training loop hands over), with and without the image recognition."""
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
o, d, v = [t.to(dev) for t in synth.pinhole_rays(size, size)]
perm = torch.randperm(o.shape[0], device=dev)
f2 = feats.clone().requires_grad_(True)
go = torch.ones((o.shape[0], 4), device=dev)

def loop(kind, fresh, reps=40):
    oo, dd, vv = (o, d, v) if kind == "image" else (o[perm].contiguous(), d[perm].contiguous(), v[perm].contiguous())
    same = svox.Rays(oo, dd, vv)
    def step():
        rays = svox.Rays(oo.clone(), dd.clone(), vv.clone()) if fresh else same
        out = r(f2, rays); out.backward(go); f2.grad = None
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

for det in (True, False):
    _C.DETECT_IMAGES = det
    for kind in ("image", "shuffled"):
        for fresh in (False, True):
            print(f"detect {det!s:5} {kind:8} {'fresh tensors' if fresh else 'same tensors ':13} {loop(kind, fresh):.3f} ms/step", flush=True)
