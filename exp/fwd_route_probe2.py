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
depth, K, fmt, size = 9, 32, "RGBA", 1024
st = synth.shell_tree(depth)
feats = synth.shell_features(st.n_features, K).to(dev)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format=fmt, device=dev)
r = svox.VolumeRenderer(tree)
o, d, v = synth.pinhole_rays(size, size)
rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
with torch.no_grad():
    print("before:", round(timed(lambda: r(feats, rays)), 3), _C.LAST_ROUTE["forward"], flush=True)
    with tree.accumulate_weights() as acc:
        print("accum:", round(timed(lambda: r(feats, rays)), 3), _C.LAST_ROUTE["forward"], flush=True)
    print("after:", round(timed(lambda: r(feats, rays)), 3), _C.LAST_ROUTE["forward"], flush=True)
    print("after2:", round(timed(lambda: r(feats, rays)), 3), _C.LAST_ROUTE["forward"], flush=True)
    print("depth:", round(timed(lambda: r.render_depth(feats, rays)), 3), flush=True)
    print("after3:", round(timed(lambda: r(feats, rays)), 3), _C.LAST_ROUTE["forward"], flush=True)
