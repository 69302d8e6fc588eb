#!/usr/bin/env python3
"""Can the march of one part of an image run beside the shade of another on this runtime?  (VERDICT r02
item 3; the r02 attempt inside the library -- 4 tile ranges, 5 cross-stream events -- doubled the forward.)
Measured from outside with what the operator layer offers: the recording forward (march_rec_kernel +
shade_tile_kernel) of the two halves of the headline image,
    serial      both halves on one stream                    (= the cost of cutting the image in two)
    two streams half 1 on the caller's stream, half 2 on a side stream, one fork and one join event
    staggered   as "two streams", but half 2 starts when half 1's march is done (an event recorded by a
                zero-size marker between ... not available from Python: approximated by launching half 2 first)
against the whole image in one call.  Prints ms per forward (mean of 50 after 10 warm-ups)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox                                      # noqa: E402
import svox_t_amd.csrc as _C                                   # noqa: E402
from svox_t_amd import synth                                   # noqa: E402
from svox_t_amd.renderer import _rays_spec_from_rays           # noqa: E402

dev = torch.device("cuda:0")
st = synth.shell_tree(8)
feats = synth.shell_features(st.n_features, 28)
tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format="SH9", device=dev)
r = svox.VolumeRenderer(tree)
W = H = 800
o, d, v = synth.pinhole_rays(W, H)
rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
opt = r._get_options()
spec = tree._spec(tree.features)
whole = _rays_spec_from_rays(rays, (H, W))
half = W * (H // 2)
halves = [_rays_spec_from_rays(svox.Rays(*(t[a:a + half].contiguous() for t in rays)), (H // 2, W)) for a in (0, half)]
for s in [whole] + halves:
    s.need_grad = False
side = torch.cuda.Stream()


def fwd(rs):
    return _C.volume_render(spec, rs, opt, record=True)


def one_call():
    fwd(whole)


def serial():
    fwd(halves[0]); fwd(halves[1])


def two_streams(first=0):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        fwd(halves[1 - first])
    fwd(halves[first])
    main.wait_stream(side)


def timeit(f, n=50, warm=10):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, f in (("whole image, one call", one_call), ("two halves, one stream", serial),
                ("two halves, two streams", two_streams), ("two halves, two streams (side stream first)", lambda: two_streams(1))):
    print(f"{name:48s} {timeit(f):.4f} ms", flush=True)
print("route:", _C.LAST_ROUTE["forward"])
