"""Auto (autograd) path on a sorted small batch: where does the host time go, and does anything wait for the GPU?"""
import os, sys, time, torch, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
from svox_t_amd import synth
import svox_t_amd.csrc as _C
dev = torch.device("cuda:0")
st = synth.shell_tree(8)
feats = synth.shell_features(st.n_features, 28).to(dev)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format="SH9", device=dev)
r = svox.VolumeRenderer(tree)
o, d, v = [t.to(dev) for t in synth.pinhole_rays(800, 800)]
idx = torch.randint(0, o.shape[0], (4096,), device=dev)
rays = svox.Rays(o[idx], d[idx], v[idx])
p = feats.clone().requires_grad_(True)
go = torch.ones((4096, 4), device=dev)
acc = {"fwd": 0.0, "bwd": 0.0, "n": 0}
f0, b0 = _C.volume_render, _C.volume_render_backward
def fw(*a, **k):
    t = time.perf_counter(); x = f0(*a, **k); acc["fwd"] += time.perf_counter() - t; return x
def bw(*a, **k):
    t = time.perf_counter(); x = b0(*a, **k); acc["bwd"] += time.perf_counter() - t; return x
_C.volume_render, _C.volume_render_backward = fw, bw
for smin in (16384, 512):
    _C.SORT_RAYS_MIN = smin
    def step():
        out = r(p, rays); out.backward(go); p.grad = None
    for _ in range(10): step()
    torch.cuda.synchronize(); acc.update(fwd=0.0, bwd=0.0)
    t0 = time.perf_counter()
    for _ in range(300): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"sort from {smin}: host {1e3*(t1-t0)/300:.3f} ms/step (inside volume_render {1e3*acc['fwd']/300:.3f}, inside volume_render_backward {1e3*acc['bwd']/300:.3f}), with drain {1e3*(t2-t0)/300:.3f}", flush=True)
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for _ in range(3): step()
    torch.cuda.set_sync_debug_mode("default")
    print("   synchronizing torch calls in 3 steps:", len(w), [str(x.message)[:80] for x in w[:3]], flush=True)
    print("   pool hints:", {k: (v[0], v[2], v[3], v[4]) for k, v in _C._POOL_HINT.items()}, flush=True)
