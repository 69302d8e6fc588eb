import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import svox_t_amd as svox, svox_t_amd.csrc as _C
from svox_t_amd.renderer import _rays_spec_from_rays
from tests.util import Case
from tests.test_gpu_roles_handoff import _canonical, _counters, _forward, CASES
gpu = torch.device("cuda:0")
for name in ("d5_rgba4_image", "cfg3"):
    kw, kind = CASES[name]
    c = Case(**kw); tree = c.tree(gpu); r = svox.VolumeRenderer(tree); opt = r._get_options(); spec = tree._spec(tree.features)
    shape = (kw["height"], kw["width"]) if kind == "image" else None
    rs = _rays_spec_from_rays(c.rays_gpu(gpu), shape); rs.need_grad = False
    _C.FWD_SPLIT = "1"; _C.FWD_OVERLAP = False
    out0, l0 = _forward(spec, rs, opt); want = _canonical(l0, shape, c.Q)
    _C.FWD_OVERLAP = True
    for flags in (0, 256, 512, 1024):
        _C.ROLES_FLAGS = flags
        out, l1 = _forward(spec, rs, opt)
        torch.cuda.synchronize()
        ctr = _counters(l1)
        W = kw["width"]
        diff = (out != out0).any(dim=1).nonzero().flatten()
        tiles = sorted(set((((diff // W) // 8) * (W // 8) + (diff % W) // 8).cpu().tolist()))
        ts = l1.tile_state[:l1.tiles].cpu().numpy()
        print(name, "flags", flags, "ctr", ctr, "busy", want[3], "diff pixels", diff.numel(), "tiles", tiles[:12], "tile%5 of diff", sorted(set(t % 5 for t in tiles)),
              "states of diff tiles", [int(ts[t]) for t in tiles[:6]], "n shaded-state", int((ts == 0x200).sum()), flush=True)
