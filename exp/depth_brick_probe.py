import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
import svox_t_amd.csrc as _C
from svox_t_amd import synth
dev = torch.device("cuda:0")
def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for depth, K, fmt, size in ((8, 28, "SH9", 800), (9, 32, "RGBA", 1024)):
    st = synth.shell_tree(depth)
    feats = synth.shell_features(st.n_features, K).to(dev)
    tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format=fmt, device=dev)
    r = svox.VolumeRenderer(tree)
    o, d, v = synth.pinhole_rays(size, size)
    rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
    for bricks, g in ((None, None), (False, None), (True, None), (False, 8), (True, 8), (True, 7)):
        _C.ACCEL_BRICKS, _C.ACCEL_LOG2 = bricks, g
        with torch.no_grad():
            print(f"depth {depth}: bricks {bricks!s:5} g {g!s:4}: render_depth {timed(lambda: r.render_depth(feats, rays, image_shape=(size, size))):.4f} ms   "
                  f"opacity {timed(lambda: r.opacity_render(feats, rays) if hasattr(r, 'opacity_render') else None):.4f} ms", flush=True)
