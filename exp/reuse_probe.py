#!/usr/bin/env python3
"""How many gradient rows would leave the CU (one atomic row per (workgroup pass, feature row))
for different tile sizes / list windows of the merge kernel?  Uses the forward's sample lists."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd import synth
from svox_t_amd.renderer import _rays_spec_from_rays

dev = torch.device("cuda:0")
st = synth.shell_tree(8)
feats = synth.shell_features(st.n_features, 28).to(dev)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format="SH9", device=dev)
r = svox.VolumeRenderer(tree)
W = H = 800
o, d, v = synth.pinhole_rays(W, H)
rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
out, lists = _C.volume_render(tree._spec(feats), _rays_spec_from_rays(rays, (H, W)), r._get_options(), record=True)
S = lists.S
rec = lists.rec[:, :, 0].long()                       # [S, Q] feature rows
n = (lists.aux[:, 0].long() & 0x7fffffff)             # [Q]
k = torch.arange(S, device=dev)[:, None].expand(S, W * H)
valid = k < n[None, :]
q = torch.arange(W * H, device=dev)[None, :].expand(S, W * H)
rows = rec[valid]
kk = k[valid]
qq = q[valid]
py, px = qq // W, qq % W
M = int(feats.shape[0])
print("samples", rows.numel(), "mean per ray", rows.numel() / (W * H))


def count(tile, window):
    t = (py // tile) * ((W + tile - 1) // tile) + (px // tile)
    p = kk // window if window else torch.zeros_like(kk)
    key = (t * 128 + p) * M + rows
    return torch.unique(key).numel()


base = count(8, 16)
for tile in (8, 16, 32):
    for window in (4, 8, 16, 32, 0):
        c = count(tile, window)
        print(f"tile {tile:2d}x{tile:<2d} window {window if window else 'all':>3}: {c / 1e6:6.2f} M rows ({c / base:.2f} of the current 8x8 / 16)")
