"""Does the ORDER in which 8 x 8 pixel tiles are walked matter?  Same tiles, row-major vs Morton tile order (through rays.order)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd.renderer import _rays_spec_from_rays
from svox_t_amd import synth
from tests.util import Case
dev = torch.device("cuda:0")
def morton2(x, y, bits=8):
    r = torch.zeros_like(x)
    for b in range(bits):
        r |= ((x >> b) & 1) << (2 * b + 1)
        r |= ((y >> b) & 1) << (2 * b)
    return r
for kw, size in ((dict(depth=9, K=32, data_format="RGBA"), 1024),):
    c = Case(width=size, height=size, **kw)
    tree = c.tree(dev); r = svox.VolumeRenderer(tree); rays = c.rays_gpu(dev)
    f = tree.features.detach().clone().requires_grad_(True)
    spec = tree._spec(f); opt = r._get_options()
    g = synth.grad_output(c.Q, kw["K"] if kw["data_format"] == "RGBA" else 4).to(dev)
    T = size // 8
    ty, tx = torch.meshgrid(torch.arange(T), torch.arange(T), indexing="ij")
    wy, wx = torch.meshgrid(torch.arange(8), torch.arange(8), indexing="ij")
    def perm_for(tile_rank):          # tile_rank [T, T] -> position of tile in the walk
        order_tiles = torch.argsort(tile_rank.reshape(-1))
        tyy, txx = ty.reshape(-1)[order_tiles], tx.reshape(-1)[order_tiles]
        q = ((tyy[:, None] * 8 + wy.reshape(-1)[None, :]) * size + txx[:, None] * 8 + wx.reshape(-1)[None, :])
        return q.reshape(-1).to(torch.int32).to(dev)
    variants = {"row-major tiles (rays.order)": perm_for(ty * T + tx), "Morton tiles (rays.order)": perm_for(morton2(tx, ty)),
                "tiles in 8x8 super-tiles": perm_for(((ty // 8) * (T // 8 + 1) + tx // 8) * 64 + (ty % 8) * 8 + tx % 8),
                "super-tiles in Morton order": perm_for(morton2(tx // 8, ty // 8) * 64 + (ty % 8) * 8 + tx % 8),
                "super-tiles in 4x4 blocks": perm_for((((ty // 32) * (T // 32 + 1) + tx // 32) * 16 + ((ty // 8) % 4) * 4 + (tx // 8) % 4) * 64 + (ty % 8) * 8 + tx % 8),
                "super-tiles in 2x2 blocks": perm_for((((ty // 16) * (T // 16 + 1) + tx // 16) * 4 + ((ty // 8) % 2) * 2 + (tx // 8) % 2) * 64 + (ty % 8) * 8 + tx % 8)}
    def run(rs):
        def step():
            out, lists = _C.volume_render(spec, rs, opt, record=True)
            return _C.volume_render_backward(spec, rs, opt, g, lists=lists)
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / 10 * 1e3
    rs = _rays_spec_from_rays(rays, (size, size)); rs.need_grad = False
    print(kw, "image hint:", "%.3f ms" % run(rs), flush=True)
    for name, p in variants.items():
        rs2 = _rays_spec_from_rays(rays, None); rs2.need_grad = False; rs2.order = p; rs2.coherent = True
        print("   ", name, "%.3f ms" % run(rs2), _C.LAST_ROUTE["backward"][:30], flush=True)
