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
o, d, v = [t.to(dev) for t in synth.pinhole_rays(800, 800)]
p = feats.clone().requires_grad_(True)
g = torch.Generator(device=dev).manual_seed(1)
for Q in (1024, 4096, 16384, 65536, 262144):
    idx = torch.randint(0, o.shape[0], (Q,), device=dev, generator=g)
    rays = svox.Rays(o[idx], d[idx], v[idx]); go = torch.ones((Q, 4), device=dev)
    for smin, scratch in ((16384, True), (512, True), (512, False), (16384, False)):
        _C.SORT_RAYS_MIN, _C.GRAD_SCRATCH = smin, scratch
        for _ in range(5):
            out = r(p, rays); out.backward(go); p.grad = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(30):
            out = r(p, rays); out.backward(go); p.grad = None
        e1.record(); torch.cuda.synchronize()
        print(f"Q {Q:7d} sort from {smin:5d}, padded gradient scratch {scratch!s:5}: {e0.elapsed_time(e1) / 30:.3f} ms   {_C.LAST_ROUTE['backward'][:34]}", flush=True)
