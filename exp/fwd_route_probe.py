import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
from svox_t_amd import synth
import svox_t_amd.csrc as _C

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

dev = torch.device("cuda:0")
for depth, K, fmt, size in ((8, 28, "SH9", 800), (9, 32, "RGBA", 1024)):
    st = synth.shell_tree(depth)
    feats = synth.shell_features(st.n_features, K).to(dev)
    tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format=fmt, device=dev)
    r = svox.VolumeRenderer(tree)
    o, d, v = synth.pinhole_rays(size, size)
    rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
    with torch.no_grad():
        ms = timed(lambda: r(feats, rays))
    print(depth, "no_grad plain:", round(ms, 3), _C.LAST_ROUTE["forward"], _C._detect_image(rays), flush=True)
    with torch.no_grad():
        ms = timed(lambda: r(feats, rays, fast=True))
    print(depth, "no_grad fast:", round(ms, 3), _C.LAST_ROUTE["forward"], flush=True)
    f2 = feats.clone().requires_grad_(True)
    ms = timed(lambda: r(f2, rays))
    print(depth, "grad fwd only:", round(ms, 3), _C.LAST_ROUTE["forward"], flush=True)
    def step():
        out = r(f2, rays); out.backward(torch.ones_like(out)); f2.grad = None
    ms = timed(step)
    print(depth, "fwd+bwd:", round(ms, 3), _C.LAST_ROUTE["forward"], "|", _C.LAST_ROUTE["backward"], flush=True)
