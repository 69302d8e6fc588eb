"""BASELINE.json configs[4] as far as ONE GPU allows: the eight 1024 x 1024 cameras of
SURVEY.md 8(d) config 5 (azimuth 30 + 45 k degrees) on the depth-9 tree with rows of 32 floats, through
parallel.render_cameras with a world of one rank -- every camera this rank's -- gathered to [8, 1024, 1024, 32],
forward and backward, against the oracle.  On eight GPUs each rank renders one of these cameras with the same
kernels; what this run shows at size is one rank's share eight times over: lists, hand-over, pixels, the stacked
result, the gradient."""
import os

import numpy as np
import pytest
import torch

import svox_t_amd as svox
import svox_t_amd.csrc as _C
from oracle import oracle as O
from svox_t_amd import parallel, synth
from tests.util import Case, assert_grads_close

pytestmark = pytest.mark.gpu

W = H = 1024
FX = 1111.111 * W / 800.0
N_CAM = 8
MIB = float(1 << 20)


def test_config5_eight_cameras_depth9_features32_one_rank(gpu, capsys):
    import torch.distributed as dist
    c = Case(depth=9, K=32, data_format="RGBA", width=8, height=8)      # (the tree; its own 8 x 8 rays are not used)
    assert (c.st.n_internal, c.st.n_features) == (792753, 4738568)
    tree = c.tree(gpu)
    r = svox.VolumeRenderer(tree)
    ot, opt = c.oracle_tree(), c.oracle_opts()
    poses = [synth.camera_pose(azimuth_deg=30.0 + 45.0 * k) for k in range(N_CAM)]
    c2ws = torch.stack([torch.from_numpy(p.astype(np.float32)) for p in poses]).to(gpu)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29900 + os.getpid() % 90))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(gpu)
        base = torch.cuda.memory_allocated(gpu)
        feats = tree.features
        full = parallel.render_cameras(r, feats, c2ws, width=W, height=H, fx=FX)
        assert full.shape == (N_CAM, H, W, 32) and full.dtype == torch.float32
        torch.cuda.synchronize()
        peak_fwd = torch.cuda.max_memory_allocated(gpu) - base
        # ---- forward: every camera, a deterministic subsample of its pixels, bit for bit
        rng = np.random.default_rng(5)
        for k in range(N_CAM):
            o, d, v = O.camera_rays(poses[k], FX, FX, W, H)
            sel = rng.choice(W * H, size=16384, replace=False)
            want = O.volume_render(ot, o[sel], d[sel], v[sel], opt)
            got = full[k].detach().reshape(W * H, 32)[torch.from_numpy(sel).to(gpu)].cpu().numpy()
            np.testing.assert_array_equal(got, want, err_msg=f"camera {k}")
        # the cameras differ (a layout that repeated one camera would pass the subsample of that camera only)
        assert not torch.equal(full[0], full[1]) and not torch.equal(full[3], full[7])
        alpha = full.detach()[..., 31]
        assert float(alpha.min()) >= 0 and float(alpha.max()) <= 1 and float(alpha.mean()) > 0.05
        # ---- backward: the loss takes every camera; the oracle's gradient is the sum over the cameras
        gout = synth.grad_output(N_CAM * W * H, 32).reshape(N_CAM, H, W, 32)
        full.backward(gout.to(gpu))
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated(gpu) - base
        assert _C.LAST_ROUTE["backward"].startswith("grad_wide_kernel"), _C.LAST_ROUTE
        want = np.zeros((c.st.n_features, 32), dtype=np.float64)
        tight = np.zeros_like(want)
        for k in range(N_CAM):
            o, d, v = O.camera_rays(poses[k], FX, FX, W, H)
            wk, _, tk = O.volume_render_backward(ot, o, d, v, opt, gout[k].reshape(W * H, 32).numpy(), want_abs="both")
            want += wk
            tight += tk
        assert_grads_close(feats.grad.cpu().numpy(), want, tight, what="8 cameras, summed")
        # ---- memory: one rank's share per camera is lists 206 MiB + hand-over 206 MiB + pixels 128 MiB; here eight of
        # them are alive at once (the forwards of all cameras precede the first backward; each also holds its 578 MiB table of
        # exponentials, r04), plus the gathered
        # [8, 1024, 1024, 32] (1 GiB), the upstream gradient (1 GiB) and two gradient tables (578 MiB each)
        with capsys.disabled():
            print(f"\n[config 5 on one rank] peak device memory above the tree: forward {peak_fwd / MIB:.0f} MiB, "
                  f"forward+backward {peak / MIB:.0f} MiB; gathered result {full.numel() * 4 / MIB:.0f} MiB")
        assert peak < 16 * 1024 * MIB, peak
    finally:
        dist.destroy_process_group()
