#!/usr/bin/env python3
"""Ad-hoc kernel timing on the headline workload (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd import synth
from svox_t_amd.renderer import _rays_spec_from_rays

dev = torch.device("cuda:0")
depth, K, fmt, W, H = 8, 28, "SH9", 800, 800
st = synth.shell_tree(depth)
feats = synth.shell_features(st.n_features, K)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format=fmt, device=dev)
r = svox.VolumeRenderer(tree)
o, d, v = synth.pinhole_rays(W, H)
rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
gout = synth.grad_output(W * H, 4).to(dev)

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

spec = tree._spec(tree.features); rs = _rays_spec_from_rays(rays)
opt = r._get_options()
optn = r._get_options(); optn.sigma_thresh = 3e38     # march everything, composite nothing
print("G env", os.environ.get("SVOXT_ACCEL_LOG2"))
print("fwd            %.3f ms" % timeit(lambda: _C.volume_render(spec, rs, opt)))
print("fwd no-payload %.3f ms" % timeit(lambda: _C.volume_render(spec, rs, optn)))
print("opacity        %.3f ms" % timeit(lambda: _C.opacity_render(spec, rs, opt)))
print("depth(full)    %.3f ms" % timeit(lambda: _C.render_depth(spec, rs, optn)))
print("depth          %.3f ms" % timeit(lambda: _C.render_depth(spec, rs, opt)))
print("count          %.3f ms" % timeit(lambda: _C.count_forward(spec, rs, opt)))
print("bwd            %.3f ms" % timeit(lambda: _C.volume_render_backward(spec, rs, opt, gout)))
print("opacity bwd    %.3f ms" % timeit(lambda: _C.opacity_render_backward(spec, rs, opt, gout[:, :1].contiguous())))
rsh = _rays_spec_from_rays(rays, (H, W))
print("fwd hint       %.3f ms" % timeit(lambda: _C.volume_render(spec, rsh, opt)))
print("bwd hint       %.3f ms" % timeit(lambda: _C.volume_render_backward(spec, rsh, opt, gout)))
if len(sys.argv) > 1 and sys.argv[1] == "perm":
    # same rays in random order: what does coherence buy?
    p = torch.randperm(W * H, device=dev)
    rp = svox.Rays(rays.origins[p].contiguous(), rays.dirs[p].contiguous(), rays.viewdirs[p].contiguous())
    rsp = _rays_spec_from_rays(rp)
    print("fwd permuted   %.3f ms" % timeit(lambda: _C.volume_render(spec, rsp, opt)))
    # sorted by step count would need the oracle; use 8x8 tiles instead
    idx = torch.arange(W * H, device=dev).view(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3).reshape(-1)
    rt = svox.Rays(rays.origins[idx].contiguous(), rays.dirs[idx].contiguous(), rays.viewdirs[idx].contiguous())
    rst = _rays_spec_from_rays(rt)
    print("fwd 8x8 tiles  %.3f ms" % timeit(lambda: _C.volume_render(spec, rst, opt)))
    print("bwd 8x8 tiles  %.3f ms" % timeit(lambda: _C.volume_render_backward(spec, rst, opt, gout[idx].contiguous())))
