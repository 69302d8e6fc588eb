"""Build libsvoxt_hip.so (the HIP kernels + C ABI) in-tree with hipcc.

Cross-compiles for gfx950 without a GPU.  Usage: ``python svox_t_amd/build.py``
(run as a script: importing the package loads the library, which is what is
being built) or ``__graft_entry__.build()``.  The library lands next to its sources in
``svox_t_amd/csrc/`` (git-ignored, but shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_NAME = "libsvoxt_hip.so"
LIB_PATH = os.path.join(CSRC, LIB_NAME)
# what the compiler says each kernel needs (registers, scratch, LDS, occupancy), written by every build:
# tests/test_cabi_and_host.py holds it to "no kernel with a dynamic stack or more than a few spilled
# registers" (a private array that ends up in scratch memory is a round trip per access -- DESIGN.md
# step 13 -- and the one place an out-of-range private index could fault)
RESOURCES_PATH = os.path.join(CSRC, "libsvoxt_hip.resources.txt")
OBJ_DIR = os.path.join(CSRC, "build")          # object files (git-ignored)
SOURCES = ["svoxt_kernels.hip", "svoxt_bwd.hip", "svoxt_build.hip", "svoxt_motion.hip", "svoxt_order.hip", "svoxt_step.hip"]
HEADERS = ["svoxt_device.h", "svoxt_host.h", "svoxt_launch.h", "svoxt_lists.h", "svoxt_fwd_kernels.h", "svoxt_bwd_kernels.h",
           "svoxt_misc_kernels.h", "svoxt_tile_reduce.inc", os.path.join("..", "..", "include", "svoxt.h")]

# -ffp-contract=off is part of the numerical contract (svoxt_device.h): the
# stepping arithmetic must not be fused into FMAs.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; cannot build " + LIB_NAME)


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, verbose: bool):
    """One translation unit -> object file (device code for gfx950 embedded); returns (returncode, cmd, stderr)."""
    obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
    cmd = [_hipcc()] + [f for f in HIPCC_FLAGS if f != "-shared"] + \
        ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", obj, os.path.join(CSRC, src)]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
    return res.returncode, cmd, res.stderr, obj


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    # the translation units are independent (no relocatable device code): compiled side by side, then linked
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        results = list(pool.map(lambda s: _compile(s, verbose), SOURCES))
    remarks, other = [], []
    for _, _, err, _ in results:
        for line in err.splitlines():
            (remarks if "-Rpass-analysis=kernel-resource-usage" in line else other).append(line)
    # the remarks come with source excerpts ("  123 | code", "      | ^"): keep warnings and errors only
    noise = [ln for ln in other if not (ln.lstrip()[:1].isdigit() or ln.lstrip().startswith("|"))]
    failed = [r for r in results if r[0] != 0]
    if failed or any("warning:" in ln or "error:" in ln for ln in noise):
        print("\n".join(noise), file=sys.stderr)
    if failed:
        raise subprocess.CalledProcessError(failed[0][0], failed[0][1])
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH + ".tmp"] + [r[3] for r in results]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.run(link, cwd=CSRC, check=True)
    with open(RESOURCES_PATH + ".tmp", "w") as f:
        f.write("\n".join(ln.split("remark: ", 1)[-1].replace(" [-Rpass-analysis=kernel-resource-usage]", "")
                          for ln in remarks) + "\n")
    os.replace(RESOURCES_PATH + ".tmp", RESOURCES_PATH)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


def kernel_resources(path: str = RESOURCES_PATH) -> dict:
    """{mangled kernel name: {"VGPRs": n, "ScratchSize [bytes/lane]": n, "Dynamic Stack": "False", ...}}
    from the remarks of the last build."""
    out, cur = {}, None
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith("Function Name:"):
                cur = out.setdefault(line.split(":", 1)[1].strip(), {})
            elif cur is not None and ":" in line:
                k, v = line.rsplit(":", 1)
                cur[k.strip()] = v.strip()
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
