#!/usr/bin/env python3
"""What the payloads WITHOUT a specialised kernel cost (they stay generic -- INTEGRATION.md says so):
forward and forward+backward at 800 x 800 on the depth-8 shell tree for an SH payload with two channels (K = 9: SH4 x 2),
an RGBA-style row of 6 floats (C = 5), and a component sub-range of SH9 (min_comp / max_comp: 1..4) -- next to SH9 itself.
The generic kernels keep their accumulators in global memory exactly as the reference does (rt_kernel.cu:300, 304) and
march in the backward (no sample lists).  No oracle here (parity: tests/test_gpu_render_parity.py d5_generic and friends)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd import synth

dev = torch.device("cuda:0")
depth, W, H = 8, 800, 800
st = synth.shell_tree(depth)
o, d, v = synth.pinhole_rays(W, H, c2w=synth.camera_pose(azimuth_deg=30.0))
rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
ROWS = [("SH9, 3 channels (specialised)", 28, "SH9", None, True)]
for pad in (True, False):      # (r05: one or two channels / other row widths are rendered as the next specialised payload: PAD_PAYLOADS)
    ROWS += [("SH4 x 2 channels" + ("" if pad else " [generic kernels]"), 9, "SH4", None, pad),
             ("SH9 x 1 channel" + ("" if pad else " [generic kernels]"), 10, "SH9", None, pad),
             ("RGBA-style row of 6 floats" + ("" if pad else " [generic kernels]"), 6, "RGBA", None, pad),
             ("RGBA-style row of 12 floats" + ("" if pad else " [generic kernels]"), 12, "RGBA", None, pad),
             ("SH9, components 1..4 only" + ("" if pad else " [generic kernels]"), 28, "SH9", (1, 4), pad)]
# (r05: more than three channels with a basis in groups of three: GROUP_PAYLOADS)
ROWS += [("SH9 x 4 channels (two groups of three)", 37, "SH9", None, True), ("SH9 x 4 channels [generic kernels]", 37, "SH9", None, False),
         ("SH4 x 6 channels (two groups of three)", 25, "SH4", None, True), ("SH4 x 6 channels [generic kernels]", 25, "SH4", None, False),
         ("RGBA-style row of 64 floats (31 + 31 + 1 channels)", 64, "RGBA", None, True), ("RGBA-style row of 64 floats [generic kernels]", 64, "RGBA", None, False)]
for label, K, fmt, comps, pad in ROWS:
    _C.PAD_PAYLOADS = _C.GROUP_PAYLOADS = pad
    feats = synth.shell_features(st.n_features, K)
    tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format=fmt, device=dev)
    r = svox.VolumeRenderer(tree) if comps is None else svox.VolumeRenderer(tree, min_comp=comps[0], max_comp=comps[1])
    f = tree.features
    cols = _C.get_out_data_dim(r._get_options(), K)
    gout = synth.grad_output(W * H, cols).to(dev)

    def fwd():
        with torch.no_grad():
            return r(f, rays, image_shape=(H, W))

    def both():
        f.grad = None
        r(f, rays, image_shape=(H, W)).backward(gout)

    res = []
    for fn in (fwd, both):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 10 * 1e3)
    print(f"{label:52s} K = {K:2d}: forward {res[0]:7.3f} ms ({W * H / res[0] / 1e3:7.1f} Mrays/s), forward+backward {res[1]:7.3f} ms "
          f"({W * H / res[1] / 1e3:7.1f} Mrays/s)   [{_C.LAST_ROUTE.get('forward')} | {_C.LAST_ROUTE.get('backward')}]", flush=True)
