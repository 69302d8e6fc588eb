"""Experiment: the tile shade gathers its feature rows from a copy padded to 128 bytes per row (one L2 line each instead of ~1.9)."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd import synth
dev = torch.device("cuda:0")
st = synth.shell_tree(8)
feats = synth.shell_features(st.n_features, 28).to(dev)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format="SH9", device=dev)
r = svox.VolumeRenderer(tree)
o, d, v = [t.to(dev) for t in synth.pinhole_rays(800, 800)]
rays = svox.Rays(o, d, v)
p = feats.clone().requires_grad_(True)
go = torch.ones((640000, 4), device=dev)
pad = torch.zeros((feats.shape[0], 32), device=dev)
pad[:, :28] = feats
def timed():
    for _ in range(30):
        out = r(p, rays, image_shape=(800, 800)); out.backward(go); p.grad = None
    ef0, ef1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tf = 0.0
    torch.cuda.synchronize()
    for _ in range(40):
        ef0.record(); out = r(p, rays, image_shape=(800, 800)); ef1.record()
        out.backward(go); p.grad = None
        torch.cuda.synchronize(); tf += ef0.elapsed_time(ef1)
    return tf / 40, out
if os.environ.get("PROBE_PAD") is not None:
    if os.environ["PROBE_PAD"] == "1":
        _C._lib.svoxt_exp_padded_rows.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _C._lib.svoxt_exp_padded_rows(pad.data_ptr(), 32)
    for _ in range(100):
        out = r(p, rays, image_shape=(800, 800)); out.backward(go); p.grad = None
    torch.cuda.synchronize()
    sys.exit(0)
base, out0 = timed()
has = hasattr(_C._lib, "svoxt_exp_padded_rows")
print("forward, rows as they are:", round(base, 4), "ms", flush=True)
if has:
    _C._lib.svoxt_exp_padded_rows.argtypes = [ctypes.c_void_p, ctypes.c_int]
    _C._lib.svoxt_exp_padded_rows(pad.data_ptr(), 32)
    t, out1 = timed()
    print("forward, rows padded to 128 bytes:", round(t, 4), "ms; same pixels:", torch.equal(out0, out1), flush=True)
    _C._lib.svoxt_exp_padded_rows(None, 0)
    print("forward, rows as they are again:", round(timed()[0], 4), "ms", flush=True)
