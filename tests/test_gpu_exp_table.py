"""svoxt_tree.exp_table (ABI v17, r04): for RGBA-style rows of 8 / 16 / 32 floats the forward builds, in the pass that
builds the sigma bitmask, a table of the rows' exponentials exp(-x) -- the sigmoids of such a payload do not depend
on the view (rt_kernel.cu:304, 420, 476) -- and the shade kernel and both sweeps of the per-tile backward read it
instead of forming one exponential per sample and channel; their double-precision quotients take the compiler's
division sequence minus v_div_scale (div_unit_range: tests/test_gpu_div_exact.py).  A pure cache: the table holds
the oracle's expf bit for bit, the forward equals the oracle bit for bit with and without it, the backward holds the
tight scale, for ordinary and for extreme feature values (exponentials that overflow to +inf or underflow to 0)."""
import ctypes

import numpy as np
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from oracle import oracle as O
from svox_t_amd import synth
from tests.util import Case, assert_grads_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [8, 16, 32])
@pytest.mark.parametrize("extreme", [False, True])
def test_exp_table_changes_nothing(gpu, K, extreme, monkeypatch):
    c = Case(depth=5, K=K, data_format="RGBA", width=64, height=56)
    if extreme:
        # exp(-x) over the whole float range: +inf (x < -88.7), 0 (x > 87), denormal results, exact zeros, huge values
        f = c.features
        g = torch.Generator().manual_seed(3)
        sel = torch.rand(f[:, :-1].shape, generator=g)
        vals = torch.tensor([-200.0, -89.0, -88.72, -87.5, -1e-30, 0.0, 1e-30, 86.9, 87.1, 103.0, 1e30, -1e30])
        pick = vals[torch.randint(0, len(vals), f[:, :-1].shape, generator=g)]
        f[:, :-1] = torch.where(sel < 0.5, pick, f[:, :-1])
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    # ---- the table itself: the oracle's expf of the negated feature columns, sigma untouched; the mask as svoxt_sigma_mask_build's
    ct = _C._pack_tree(tree._spec(tree.features))
    mask = torch.zeros((_C._lib.svoxt_sigma_mask_bytes(ct.M) // 8,), dtype=torch.int64, device=gpu)
    mask2 = torch.zeros_like(mask)
    etab = torch.empty_like(tree.features.detach())
    _C._call("svoxt_exp_table_build", ctypes.byref(ct), ctypes.c_float(0.0), _C._ptr(mask), _C._ptr(etab), _C._stream(gpu))
    _C._call("svoxt_sigma_mask_build", ctypes.byref(ct), ctypes.c_float(0.0), _C._ptr(mask2), _C._stream(gpu))
    fe = c.features.numpy()
    want_tab = np.concatenate([O.expf(-fe[:, :-1]), fe[:, -1:]], axis=1)
    np.testing.assert_array_equal(etab.cpu().numpy().view(np.uint32), want_tab.astype(np.float32).view(np.uint32))
    M = fe.shape[0]
    bits = lambda m: np.unpackbits(m.cpu().numpy().view(np.uint8), bitorder="little")[:M]
    np.testing.assert_array_equal(bits(mask), bits(mask2))
    np.testing.assert_array_equal(bits(mask), (fe[:, -1] > 0).astype(np.uint8))
    # ---- forward + backward with and without it
    gout = synth.grad_output(c.Q, K)
    want = O.volume_render(c.oracle_tree(), *c.rays_np(), c.oracle_opts())
    gwant, _, tight = O.volume_render_backward(c.oracle_tree(), *c.rays_np(), c.oracle_opts(), gout.numpy(), want_abs="both")
    for table in (True, False):
        monkeypatch.setattr(_C, "EXP_TABLE", table)
        with torch.no_grad():
            inf = r(tree.features, c.rays_gpu(gpu)).cpu().numpy()           # scratch lists (inference)
        np.testing.assert_array_equal(inf, want)
        tree.features.grad = None
        out = r(tree.features, c.rays_gpu(gpu), image_shape=(56, 64))
        out.backward(gout.to(gpu))
        assert _C.LAST_ROUTE["backward"].startswith("grad_wide_kernel"), _C.LAST_ROUTE
        assert ("forward's table" in _C.LAST_ROUTE["backward"]) == table
        np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
        got = tree.features.grad.cpu().numpy()
        assert np.isfinite(got).all()
        assert_grads_close(got, gwant, tight, what=f"table={table}")
