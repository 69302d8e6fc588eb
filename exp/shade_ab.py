#!/usr/bin/env python3
"""The recording forward of the headline workload as TWO launches (march_rec_kernel, then shade_tile_kernel: FWD_OVERLAP
False) and as one (fwd_roles_kernel), timed with events: for A/B runs of shade variants (SVOXT_LIB=exp/libsvoxt_<name>.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svox_t_amd as svox, svox_t_amd.csrc as _C
from svox_t_amd import synth
from svox_t_amd.renderer import _rays_spec_from_rays
dev = torch.device("cuda:0")
st = synth.shell_tree(8)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, synth.shell_features(st.n_features, 28), data_format="SH9", device=dev)
o, d, v = synth.pinhole_rays(800, 800, c2w=synth.camera_pose(azimuth_deg=30.0))
rs = _rays_spec_from_rays(svox.Rays(o.to(dev), d.to(dev), v.to(dev)), (800, 800))
opt = svox.VolumeRenderer(tree)._get_options()
spec = tree._spec(tree.features)
for overlap in (False, True):
    _C.FWD_SPLIT, _C.FWD_OVERLAP = "1", overlap
    for _ in range(300):
        _C.volume_render(spec, rs, opt, record=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        _C.volume_render(spec, rs, opt, record=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"{os.path.basename(os.environ.get('SVOXT_LIB', 'in-tree'))}: recording forward, {'one launch (roles)' if overlap else 'two launches'}: "
          f"{e0.elapsed_time(e1) / 200:.4f} ms   [{_C.LAST_ROUTE['forward']}]", flush=True)
