import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd.renderer import _rays_spec_from_rays
from tests.util import Case
from svox_t_amd import synth
gpu = torch.device("cuda:0")
case = Case(depth=8, K=28, data_format="SH9", width=800, height=800)
tree = case.tree(gpu)
r = svox.VolumeRenderer(tree)
rays = case.rays_gpu(gpu)
g = synth.grad_output(case.Q, 4).to(gpu)
def py_step():
    tree.features.grad = None
    r(tree.features, rays, image_shape=(800, 800)).backward(g)
def timed(fn, n=60):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
which = sys.argv[1]
if which != "none":
    spec = tree._spec(tree.features)
    ct = _C._pack_tree_accel(spec, which == "bricks")
print(which, "py_step", round(timed(py_step), 4), _C.LAST_ROUTE["forward"][:30], [(k[1], v[5]) for k, v in _C._ACCEL_CACHE.items()], flush=True)
print(which, "py_step again", round(timed(py_step), 4), flush=True)
