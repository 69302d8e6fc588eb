#!/usr/bin/env python3
"""svoxt_ray_order on a shuffled 800x800 batch, 50 calls (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svox_t_amd as svox, svox_t_amd.csrc as _C
from svox_t_amd import synth
from svox_t_amd.renderer import _rays_spec_from_rays
dev = torch.device("cuda:0")
st = synth.shell_tree(8)
feats = synth.shell_features(st.n_features, 28)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format="SH9", device=dev)
r = svox.VolumeRenderer(tree)
o, d, v = synth.pinhole_rays(800, 800)
p = torch.randperm(o.shape[0])
rays = svox.Rays(o[p].contiguous().to(dev), d[p].contiguous().to(dev), v[p].contiguous().to(dev))
spec, rs, opt = tree._spec(tree.features), _rays_spec_from_rays(rays), r._get_options()
for _ in range(50):
    _C._ray_order32(spec, rs, opt)
torch.cuda.synchronize()
