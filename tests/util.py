"""Shared helpers for the parity tests: build a case once, run it through the
CPU oracle (numpy) and through the HIP path (torch tensors on the GPU)."""
from __future__ import annotations

import numpy as np

import svox_t_amd as svox
from oracle import oracle as O
from svox_t_amd import synth

FMT = {"RGBA": O.FORMAT_RGBA, "SH": O.FORMAT_SH, "SG": O.FORMAT_SG, "ASG": O.FORMAT_ASG}


class Case:
    """One workload: shell tree of depth D, feature width K, W x H pinhole rays."""

    def __init__(self, depth, K, data_format, width, height, radius=0.5, center=(0.5, 0.5, 0.5),
                 azimuth=30.0, seed=0):
        self.st = synth.shell_tree(depth)
        self.K = K
        self.data_format = data_format
        self.features = synth.shell_features(self.st.n_features, K, seed=seed)
        self.radius, self.center = radius, center
        # camera circles the *world* cube: centre and size follow radius/center
        r = np.atleast_1d(np.asarray(radius, dtype=np.float64)) * np.ones(3)
        c = np.asarray(center, dtype=np.float64)
        pose = synth.camera_pose(azimuth_deg=azimuth, radius=1.6 * 2 * float(r.max()), center=c)
        self.origins, self.dirs, self.vdirs = synth.pinhole_rays(width, height, c2w=pose)
        self.Q = width * height
        df = svox.DataFormat(data_format)
        self.format, self.basis_dim = df.format, df.basis_dim

    def tree(self, device="cpu"):
        return svox.N3Tree.from_arrays(self.st.child, self.st.data, self.st.parent_depth,
                                       self.features, data_format=self.data_format,
                                       radius=self.radius, center=self.center, device=device)

    def oracle_tree(self):
        t = self.tree()
        return O.Tree(self.features.numpy(), self.st.data, self.st.child,
                      offset=t.offset.numpy(), scaling=t.invradius.numpy())

    def oracle_opts(self, fast=False, **kw):
        th = 1e-2 if fast else 0.0
        return O.make_options(format=self.format, basis_dim=self.basis_dim,
                              sigma_thresh=th, stop_thresh=th, **kw)

    def rays_np(self):
        return self.origins.numpy(), self.dirs.numpy(), self.vdirs.numpy()

    def rays_gpu(self, device):
        return svox.Rays(self.origins.to(device), self.dirs.to(device), self.vdirs.to(device))


def assert_outputs_close(got, want, rtol=1e-5, atol=1e-6, what="output"):
    """|got - want| <= rtol * |want| + atol, elementwise (fp32 relative 1e-5 is
    the north-star tolerance; atol covers values that are exactly 0)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    err = np.abs(got - want)
    bound = rtol * np.abs(want) + atol
    bad = err > bound
    assert not bad.any(), (f"{what}: {bad.sum()} / {bad.size} elements off; max err {err.max():.3e}, "
                           f"worst ratio {(err / bound).max():.2f}")


def assert_grads_close(got, want, abs_sum, rtol=1e-5, what="grad"):
    """Float-atomic accumulation reorders sums, so the bound scales with the
    sum of |contributions| per entry (from the oracle): |got - want| <=
    rtol * sum|c| (+ a denormal-sized floor)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    bound = rtol * np.asarray(abs_sum, dtype=np.float64) + 1e-30
    bad = err > bound
    assert not bad.any(), (f"{what}: {bad.sum()} / {bad.size} entries off; "
                           f"worst ratio {(err / bound).max():.2f}, max err {err.max():.3e}")
    # entries the oracle never touches must be exactly zero
    untouched = np.asarray(abs_sum) == 0
    assert np.all(got[untouched] == 0), f"{what}: non-zero gradient where the oracle has no contribution"
