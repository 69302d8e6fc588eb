import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd.renderer import _rays_spec_from_rays
from tests.util import Case
dev = torch.device("cuda:0")
c = Case(depth=9, K=32, data_format="RGBA", width=1024, height=1024)
tree = c.tree(dev); r = svox.VolumeRenderer(tree); rays = c.rays_gpu(dev)
spec = tree._spec(tree.features); opt = r._get_options()
rs = _rays_spec_from_rays(rays, (1024, 1024)); rs.need_grad = False
out, lists = _C.volume_render(spec, rs, opt, record=True)
n = (lists.aux[:, 0] & 0x7fffffff).long()          # records per ray (by ray index q)
H = W = 1024
img = n.view(H, W)
tiles = img.view(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3).reshape(-1, 64)
def stats(t, name):
    mx = t.max(dim=1).values; sm = t.sum(dim=1)
    win = ((mx + 15) // 16).sum().item()
    print(f"{name}: tiles {t.shape[0]}, non-empty {(mx > 0).sum().item()}, sum of windows {win}, records {sm.sum().item()}, slot use {sm.sum().item() / max(1, (((mx + 15) // 16) * 16 * 64).sum().item()):.3f}")
stats(tiles, "8x8 pixel tiles")
rs2 = _rays_spec_from_rays(rays, None)
perm = _C.ray_order(spec, rs2, opt)
stats(n[perm].view(-1, 64), "groups of 64 in svoxt_ray_order's order")
for th, tw in ((4, 16), (16, 4), (2, 32)):
    t2 = img.view(H // th, th, W // tw, tw).permute(0, 2, 1, 3).reshape(-1, 64)
    stats(t2, f"{th}x{tw} pixel tiles")
