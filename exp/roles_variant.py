#!/usr/bin/env python3
"""Builds exp/libsvoxt_roles8.so: the library with fwd_roles_kernel held to 64 registers
(__launch_bounds__(512, 8): four workgroups per CU instead of three, at the price of a few spilled
registers).  Run the benchmark with SVOXT_LIB=exp/libsvoxt_roles8.so to compare (NOTEBOOK.md step 33)."""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _flatten import CSRC, flat_source   # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = flat_source()
pat = r"__global__ void __launch_bounds__\(512\)\nfwd_roles_kernel"
assert len(re.findall(pat, src)) == 1
src = re.sub(pat, "__global__ void __launch_bounds__(512, 8)\nfwd_roles_kernel", src)
tmp = os.path.join(CSRC, "_roles8.hip")
open(tmp, "w").write(src)
out = os.path.join(ROOT, "exp", "libsvoxt_roles8.so")
try:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                    "-fno-fast-math", "-o", out, tmp] + [os.path.join(CSRC, f) for f in ("svoxt_build.hip", "svoxt_motion.hip", "svoxt_order.hip")],
                   check=True, cwd=CSRC)
finally:
    os.remove(tmp)
print(out)
