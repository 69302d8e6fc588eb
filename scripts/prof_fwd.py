#!/usr/bin/env python3
"""GPU box: N forwards of the headline workload for a rocprofv3 kernel trace.
usage: prof_fwd.py [record=0|1] [n]   (SVOXT_FWD_SPLIT / SVOXT_FWD_LIST select the route)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd import synth
from svox_t_amd.renderer import _rays_spec_from_rays

record = len(sys.argv) > 1 and sys.argv[1] == "1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
which = sys.argv[3] if len(sys.argv) > 3 else "d8"
dev = torch.device("cuda:0")
depth, K, fmt, W, H = (8, 28, "SH9", 800, 800) if which == "d8" else (9, 32, "RGBA", 1024, 1024)
st = synth.shell_tree(depth)
feats = synth.shell_features(st.n_features, K)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format=fmt, device=dev)
r = svox.VolumeRenderer(tree)
o, d, v = synth.pinhole_rays(W, H)
rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
spec = tree._spec(tree.features)
rsh = _rays_spec_from_rays(rays, (H, W))
rsh.need_grad = False             # record=0: a forward nobody differentiates
opt = r._get_options()
for _ in range(n):
    _C.volume_render(spec, rsh, opt, record=record)
torch.cuda.synchronize()
print("done")
