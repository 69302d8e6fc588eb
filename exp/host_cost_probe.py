"""Where the host time of the two plain calls goes (small batch: the GPU is never the limit), function by function."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
from svox_t_amd import synth
import svox_t_amd.csrc as _C
from svox_t_amd.renderer import _rays_spec_from_rays

dev = torch.device("cuda:0")
st = synth.shell_tree(8)
feats = synth.shell_features(st.n_features, 28).to(dev)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format="SH9", device=dev)
r = svox.VolumeRenderer(tree)
o, d, v = [t.to(dev) for t in synth.pinhole_rays(800, 800)]
idx = torch.randint(0, o.shape[0], (4096,), device=dev)
o, d, v = o[idx], d[idx], v[idx]
_C.SORT_RAYS_MIN = int(os.environ.get("PROBE_SORT_MIN", "16384"))
p = feats.clone().requires_grad_(True)
rays = svox.Rays(o, d, v)
go = torch.ones((o.shape[0], 4), device=dev)
def auto():
    out = r(p, rays); out.backward(go); p.grad = None
opt = r._get_options(False) if hasattr(r, "_get_options") else None
def direct():
    ts = tree._spec(p); rs = _rays_spec_from_rays(rays, None)
    out = _C.volume_render(ts, rs, opt)
    return _C.volume_render_backward(ts, rs, opt, go)
for fn in (auto, direct):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{fn.__name__}: host {1e3*(t1-t0)/300:.3f} ms/step, with drain {1e3*(t2-t0)/300:.3f}", flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): direct()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(35)
