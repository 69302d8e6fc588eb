import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd import synth
dev = torch.device("cuda:0")
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for depth, K, fmt, size in ((9, 28, "SH9", 800), (9, 28, "SH9", 1024), (9, 4, "RGBA", 1024), (8, 32, "RGBA", 800), (8, 28, "SH9", 1024), (8, 28, "SH9", 1600), (9, 32, "RGBA", 800)):
    st = synth.shell_tree(depth)
    feats = synth.shell_features(st.n_features, K).to(dev).requires_grad_(True)
    tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats.detach(), data_format=fmt, device=dev)
    r = svox.VolumeRenderer(tree)
    o, d, v = synth.pinhole_rays(size, size)
    rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
    C = K if fmt == "RGBA" and K > 4 else 4
    gout = torch.randn(size * size, C, device=dev)
    def step():
        feats.grad = None
        r(feats, rays, image_shape=(size, size)).backward(gout)
    res = []
    for thr in (1 << 62, 0):
        _C._lib.svoxt_set_super_tile_rows(thr)
        _C._POOL_HINT.clear()
        for _ in range(4):
            step(); torch.cuda.synchronize()
        res.append(timed(step))
    _C._lib.svoxt_set_super_tile_rows(-1)
    print(f"depth {depth} {fmt} K={K} {size}x{size} (features {st.n_features * K * 4 / 2**20:.0f} MiB): row-major {res[0]:.3f} ms, super-tiles {res[1]:.3f} ms", flush=True)
    del tree, feats, r
    torch.cuda.empty_cache()
