"""The C ABI used from C++ with no Python / torch in the process (examples/c_abi_demo.cpp):
compiled with hipcc against include/svoxt.h and libsvoxt_hip.so, fed a case through
files, checked against the CPU oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from svox_t_amd import synth
from tests.util import Case, assert_grads_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("fmt,K,side", [("SH9", 28, 64), ("RGBA", 32, 40)])
def test_c_abi_demo_matches_oracle(gpu, tmp_path, fmt, K, side):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = tmp_path / "c_abi_demo"
    libdir = os.path.join(ROOT, "svox_t_amd", "csrc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_abi_demo.cpp"), "-L", libdir, "-lsvoxt_hip",
                    f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
    c = Case(depth=5, K=K, data_format=fmt, width=side, height=side, radius=[1.0, 1.2, 0.8], center=[0.1, -0.2, 0.3])
    ot = c.oracle_tree()
    opt = c.oracle_opts()
    cols = O.out_data_dim(opt, K)
    g = synth.grad_output(c.Q, cols).numpy()
    n = c.st.child.shape[0]
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([n, c.st.n_features, K, c.Q, c.format, c.basis_dim, side, side], np.int64).tofile(f)
        for a, dt in ((c.st.child, np.int32), (c.st.data, np.int32), (ot.offset, np.float32), (ot.scaling, np.float32),
                      (c.features.numpy(), np.float32), (c.origins.numpy(), np.float32), (c.dirs.numpy(), np.float32),
                      (c.vdirs.numpy(), np.float32), (g, np.float32)):
            np.ascontiguousarray(a, dtype=dt).tofile(f)
    subprocess.run([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], check=True, timeout=120)
    raw = np.fromfile(tmp_path / "out.bin", dtype=np.uint8)
    assert int(np.frombuffer(raw[:8], np.int64)[0]) == cols
    body = np.frombuffer(raw[8:], np.float32)
    out = body[:c.Q * cols].reshape(c.Q, cols)
    depth = body[c.Q * cols:c.Q * cols + c.Q].reshape(c.Q, 1)
    grad = body[c.Q * cols + c.Q:].reshape(c.st.n_features, K)
    np.testing.assert_array_equal(out, O.volume_render(ot, *c.rays_np(), opt))
    np.testing.assert_array_equal(depth, O.render_depth(ot, *c.rays_np(), opt))
    want, ab = O.volume_render_backward(ot, *c.rays_np(), opt, g, want_abs=True)
    assert_grads_close(grad, want, ab)
