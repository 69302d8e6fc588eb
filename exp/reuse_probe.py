#!/usr/bin/env python3
"""How many gradient rows would leave the CU -- one atomic row per (tile, list window, feature row)
-- for different list windows of a per-tile merge?  Reads the sample lists a recording forward
leaves (blocks of 8 positions x 64 rays, svoxt.h).   reuse_probe.py [d8_sh9_800 | d9_rgba32_1024]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd import synth
from svox_t_amd.renderer import _rays_spec_from_rays

which = sys.argv[1] if len(sys.argv) > 1 else "d8_sh9_800"
depth, K, fmt, W, H = {"d8_sh9_800": (8, 28, "SH9", 800, 800), "d9_rgba32_1024": (9, 32, "RGBA", 1024, 1024)}[which]
dev = torch.device("cuda:0")
st = synth.shell_tree(depth)
feats = synth.shell_features(st.n_features, K).to(dev)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format=fmt, device=dev)
r = svox.VolumeRenderer(tree)
o, d, v = synth.pinhole_rays(W, H)
rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
spec = _rays_spec_from_rays(rays, (H, W))
out, lists = _C.volume_render(tree._spec(feats), spec, r._get_options(), record=True)
torch.cuda.synchronize()
S, tiles = lists.S, lists.tiles
nb = S // 8
n = (lists.aux[:, 0].long() & 0x7fffffff)                      # [Q] records per ray (caller order)
if lists.pooled:
    tab = lists.blocktab.view(tiles, nb).long()                # block of (tile, k // 8), -1: none
else:
    tab = torch.arange(tiles * nb, device=dev).view(tiles, nb)
rec = lists.rec.view(lists.pool_blocks, 64, 8, 2)[:, :, :, 0].long()    # [block][lane][k % 8] -> feature row
# records per lane of a tile: the lists do not say which ray a lane is, but a lane's count is
# the number of its valid slots; take it from the table + a sentinel-free count: slots are filled
# in order, so count = what aux says for the ray the launch maps there (8x8 tiles of the image)
tx = W // 8
lane = torch.arange(64, device=dev)
t_id = torch.arange(tiles, device=dev)
py = (t_id // tx)[:, None] * 8 + (lane // 8)[None, :]
px = (t_id % tx)[:, None] * 8 + (lane % 8)[None, :]
cnt = n[(py * W + px).clamp(max=W * H - 1)]                    # [tiles, 64]
M = int(feats.shape[0])
total = int(cnt.sum())
print(which, "samples", total, "mean per ray", total / (W * H), "cap", S, "pooled", lists.pooled)
for window in (1, 2, 4, 8, 16, 0):                             # in blocks of 8 positions; 0 = the whole list
    uniq = 0
    for lo in range(0, tiles, 2048):                           # in slices of tiles: bounded memory
        hi = min(tiles, lo + 2048)
        b = tab[lo:hi]                                         # [t, nb]
        ok_b = b >= 0
        rows = rec[b.clamp(min=0)]                             # [t, nb, 64, 8]
        k = (torch.arange(nb, device=dev)[:, None, None] * 8 + torch.arange(8, device=dev)[None, None, :])
        valid = ok_b[:, :, None, None] & (k[None] < cnt[lo:hi][:, None, :, None])
        wid = (torch.arange(nb, device=dev) // window if window else torch.zeros(nb, dtype=torch.long, device=dev))
        key = ((torch.arange(hi - lo, device=dev)[:, None, None, None] * 64 + wid[None, :, None, None]) * M + rows)[valid]
        uniq += torch.unique(key).numel()
    print(f"window {window * 8 if window else 'all':>4} positions: {uniq / 1e6:7.2f} M (tile, window, row) triples = {total / uniq:5.2f} samples per row sent")
