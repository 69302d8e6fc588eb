#!/usr/bin/env python3
"""GPU box: what the sample lists of one forward occupy, dense against pooled (configs 3 and 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd import synth
from svox_t_amd.renderer import _rays_spec_from_rays
dev = torch.device("cuda:0")
for name, (depth, K, fmt, W, H) in {"config 3 (D=8 SH9 800x800)": (8, 28, "SH9", 800, 800),
                                     "config 4 (D=9 K=32 1024x1024)": (9, 32, "RGBA", 1024, 1024)}.items():
    st = synth.shell_tree(depth)
    tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, synth.shell_features(st.n_features, K), data_format=fmt, device=dev)
    r = svox.VolumeRenderer(tree)
    o, d, v = synth.pinhole_rays(W, H)
    rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
    spec, opt = tree._spec(tree.features), r._get_options()
    for pooled in (False, True):
        _C.LIST_POOL = pooled
        _C._POOL_HINT.clear()
        for it in range(4):                       # the pool size settles after the first forwards
            rs = _rays_spec_from_rays(rays, (H, W))
            out, lists = _C._volume_render(spec, rs, opt, True)
            torch.cuda.synchronize()
        n = (lists.aux[:, 0] & 0x7fffffff).long()
        over = int((lists.aux[:, 0] < 0).sum())
        used = int(((lists.pool_next.view(32, 16)[:, 0].long() + 1).sum())) if lists.pooled else None
        print(f"{name}: {'pooled' if pooled else 'dense '} cap {lists.S:3d}/ray  rec {lists.rec.numel() * 4 / 2**20:7.1f} MiB"
              f" (+ table {0 if lists.blocktab is None else lists.blocktab.numel() * 4 / 2**20:.1f} MiB)"
              f"  records {int(n.sum())} = {int(n.sum()) * 8 / 2**20:.1f} MiB, longest list {int(n.max())}, rays over the cap {over}"
              + (f", blocks used {used} of {lists.pool_blocks}" if used is not None else ""))
