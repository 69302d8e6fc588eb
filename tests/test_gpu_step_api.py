"""svoxt_step_plan / _forward / _backward (include/svoxt.h, ABI v21): a training step planned by the LIBRARY, for hosts
that do not want to re-implement the operator layer's routing (INTEGRATION.md route C).  Driven here through ctypes with
raw device pointers -- nothing of svox_t_amd/csrc/__init__.py's policy is involved: only its marshalling -- and held to
the oracle like every other route: forward bit for bit, gradient on the tight scale."""
import ctypes

import numpy as np
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from oracle import oracle as O
from svox_t_amd import synth
from svox_t_amd.csrc import _abi
from svox_t_amd.renderer import _rays_spec_from_rays
from tests.util import Case, assert_grads_close

pytestmark = pytest.mark.gpu

CASES = {
    "d6_sh9_image": (dict(depth=6, K=28, data_format="SH9", width=96, height=64), True, "grad_fused"),
    "d6_rgba32_image": (dict(depth=6, K=32, data_format="RGBA", width=96, height=96), True, "grad_wide"),
    "d5_rgba4_image": (dict(depth=5, K=4, data_format="RGBA", width=64, height=64), True, "grad_fused"),
    "d5_sh4_world_rays": (dict(depth=5, K=13, data_format="SH4", width=61, height=47, radius=[1.0, 1.2, 0.8], center=[0.1, -0.2, 0.3]), False, "render_bwd"),
    "d4_sh25_image": (dict(depth=4, K=76, data_format="SH25", width=48, height=48), True, "grad_fused"),
    "d5_rgba8_rays": (dict(depth=5, K=8, data_format="RGBA", width=64, height=64), False, "render_bwd"),
    "d5_sh4_two_channels": (dict(depth=5, K=9, data_format="SH4", width=48, height=48), True, "grad_fused"),    # rendered as three channels (pad_K = 13)
    "d5_rgba6_image": (dict(depth=5, K=6, data_format="RGBA", width=48, height=48), True, "grad_wide"),          # ... as a row of 8 floats
    "d5_rgba3_rays": (dict(depth=5, K=3, data_format="RGBA", width=47, height=31), False, "render_bwd"),         # ... of 4
    "d4_generic": (dict(depth=4, K=37, data_format="SH9", width=40, height=40), True, None),      # four channels: no lists, both calls march
}


def _run(case, gpu, image, thresholds=(0.0, 0.0), pool_blocks=0):
    tree = case.tree(gpu)
    r = svox.VolumeRenderer(tree)
    r.sigma_thresh, r.stop_thresh = thresholds
    opt = r._get_options()
    rays = case.rays_gpu(gpu)
    spec = tree._spec(tree.features)
    rs = _rays_spec_from_rays(rays, case.image if image else None)
    ct, cr, co = _C._pack_tree_accel(spec, True), _C._pack_rays(rs), _C._pack_opts(opt)
    step = _abi._CStep()
    _C._call("svoxt_step_plan", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co), pool_blocks, ctypes.byref(step))
    ws = torch.empty((step.workspace_bytes + 256,), dtype=torch.uint8, device=gpu)
    base = (ws.data_ptr() + 255) // 256 * 256
    cols = _C.get_out_data_dim(opt, case.K)
    out = torch.empty((case.Q, cols), dtype=torch.float32, device=gpu)
    grad = torch.full((tree.features.shape[0], case.K), float("nan"), dtype=torch.float32, device=gpu)
    g = synth.grad_output(case.Q, cols, seed=11).to(gpu)
    st = _C._stream(gpu)
    _C._call("svoxt_step_forward", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co), out.data_ptr(), ctypes.byref(step), base, st)
    _C._call("svoxt_step_backward", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co), g.data_ptr(), grad.data_ptr(),
             ctypes.byref(step), base, st)
    torch.cuda.synchronize()
    oo = O.make_options(format=case.format, basis_dim=case.basis_dim, sigma_thresh=thresholds[0], stop_thresh=thresholds[1])
    return out.cpu().numpy(), grad.cpu().numpy(), g.cpu().numpy(), oo, step


@pytest.mark.parametrize("name", list(CASES))
def test_step_api_matches_the_oracle(gpu, name):
    kw, image, _ = CASES[name]
    case = Case(**kw)
    case.image = (kw["height"], kw["width"])
    for th in ((0.0, 0.0), (1e-2, 1e-2)):
        out, grad, g, oo, step = _run(case, gpu, image, th)
        np.testing.assert_array_equal(out, O.volume_render(case.oracle_tree(), *case.rays_np(), oo))
        gw, _, tight = O.volume_render_backward(case.oracle_tree(), *case.rays_np(), oo, g, want_abs="both")
        assert_grads_close(grad, gw, tight)                 # (every element written: the NaN fill is gone)
        assert bool(step.records) == (name != "d4_generic")
        assert (step.pad_K > 0) == (name in ("d5_sh4_two_channels", "d5_rgba6_image", "d5_rgba3_rays")), step.pad_K


def test_step_api_with_a_pool_that_runs_dry(gpu):
    """Too few list blocks: rays that find none stop recording and march their remainder -- slower, never wrong."""
    kw, image, _ = CASES["d6_sh9_image"]
    case = Case(**kw)
    case.image = (kw["height"], kw["width"])
    out, grad, g, oo, step = _run(case, gpu, image, pool_blocks=32)
    assert step.lists.pool_blocks == 32
    np.testing.assert_array_equal(out, O.volume_render(case.oracle_tree(), *case.rays_np(), oo))
    gw, _, tight = O.volume_render_backward(case.oracle_tree(), *case.rays_np(), oo, g, want_abs="both")
    assert_grads_close(grad, gw, tight)


def test_step_api_headline_size_and_speed(gpu):
    """BASELINE configs[2] through the step API alone: same pixels as the oracle, gradient on the tight scale, and a
    step time within 15 % of the Python operator layer's on the same box (the policy is the same; what differs is the
    operator layer's kept gradient scratch and cached workspace)."""
    case = Case(depth=8, K=28, data_format="SH9", width=800, height=800)
    case.image = (800, 800)
    tree = case.tree(gpu)
    r = svox.VolumeRenderer(tree)
    opt = r._get_options()
    rays = case.rays_gpu(gpu)
    rs = _rays_spec_from_rays(rays, (800, 800))
    spec = tree._spec(tree.features)
    ct, cr, co = _C._pack_tree_accel(spec, True), _C._pack_rays(rs), _C._pack_opts(opt)
    step = _abi._CStep()
    _C._call("svoxt_step_plan", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co), 0, ctypes.byref(step))
    ws = torch.empty((step.workspace_bytes + 256,), dtype=torch.uint8, device=gpu)
    base = (ws.data_ptr() + 255) // 256 * 256
    out = torch.empty((case.Q, 4), dtype=torch.float32, device=gpu)
    grad = torch.empty((tree.features.shape[0], 28), dtype=torch.float32, device=gpu)
    g = synth.grad_output(case.Q, 4).to(gpu)
    st = _C._stream(gpu)

    def c_step():
        _C._call("svoxt_step_forward", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co), out.data_ptr(), ctypes.byref(step), base, st)
        _C._call("svoxt_step_backward", ctypes.byref(ct), ctypes.byref(cr), ctypes.byref(co), g.data_ptr(), grad.data_ptr(),
                 ctypes.byref(step), base, st)

    def py_step():
        tree.features.grad = None
        r(tree.features, rays, image_shape=(800, 800)).backward(g)

    def timed(fn, n=60):
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    t_c, t_py = timed(c_step), timed(py_step)
    t_c2 = timed(c_step)
    print(f"\n[step API, 800 x 800 / depth 8 / SH9] C step {t_c:.4f} / {t_c2:.4f} ms, Python operator layer {t_py:.4f} ms "
          f"({640 / t_c:.0f} vs {640 / t_py:.0f} Mrays/s); workspace {step.workspace_bytes / 2**20:.1f} MiB, pool {step.lists.pool_blocks} blocks")
    oo = case.oracle_opts()
    np.testing.assert_array_equal(out.cpu().numpy(), O.volume_render(case.oracle_tree(), *case.rays_np(), oo))
    gw, _, tight = O.volume_render_backward(case.oracle_tree(), *case.rays_np(), oo, g.cpu().numpy(), want_abs="both")
    assert_grads_close(grad.cpu().numpy(), gw, tight)
    assert min(t_c, t_c2) <= 1.15 * t_py
