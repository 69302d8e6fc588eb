"""div_unit_range / rcp_unit_range (svoxt_device.h): the double-precision quotients of the shade and gradient kernels
for rows of 8 / 16 / 32 floats with the compiler's own rcp / fma sequence minus v_div_scale (an identity on their
operand range) -- against the `/` operator for EVERY float e >= 0 in d = 1 + e, on the GPU (exp/div_check.hip)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lean_double_division_equals_the_operator_for_every_denominator(gpu, tmp_path):
    exe = str(tmp_path / "div_check")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off",
                    "-I", os.path.join(ROOT, "svox_t_amd", "csrc"), "-o", exe, os.path.join(ROOT, "exp", "div_check.hip")],
                   check=True, capture_output=True, timeout=600)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and " 0 mismatches" in p.stdout, p.stdout + p.stderr
    print(p.stdout)
