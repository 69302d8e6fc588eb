"""A 4096 x 4096 image (16.8 M rays: 26 x the headline batch) through the default route, against the oracle: index widths, pool sizes, grids."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox_t_amd as svox
import svox_t_amd.csrc as _C
from oracle import oracle as O
from svox_t_amd import synth
from tests.util import Case, assert_grads_close
gpu = torch.device("cuda:0")
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
c = Case(depth=8, K=28, data_format="SH9", width=W, height=H)
tree = c.tree(gpu)
r = svox.VolumeRenderer(tree)
rays = c.rays_gpu(gpu)
f = tree.features.detach().clone().requires_grad_(True)
g = synth.grad_output(c.Q, 4, seed=3)
gg = g.to(gpu)
for name, kw in (("declared image", dict(image_shape=(H, W))), ("undeclared", {})):
    f.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = r(f, rays, **kw); out.backward(gg)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    f.grad = None
    out = r(f, rays, **kw); out.backward(gg)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: first step {1e3*(t1-t0):.1f} ms, second {1e3*(t2-t1):.1f} ms ({c.Q/(t2-t1)/1e6:.0f} Mrays/s); {_C.LAST_ROUTE['forward'][:40]} | {_C.LAST_ROUTE['backward'][:40]}; "
          f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    got_out, got_grad = out.detach().cpu().numpy(), f.grad.cpu().numpy()
    if name == "declared image":
        t0 = time.perf_counter()
        want = O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts())
        gw, _, tight = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), g.numpy(), want_abs="both")
        print(f"oracle: {time.perf_counter()-t0:.1f} s", flush=True)
    np.testing.assert_array_equal(got_out, want)
    assert_grads_close(got_grad, gw, tight)
    print(f"{name}: forward bit-equal, gradient within 1e-5 of the tight scale", flush=True)
