#!/usr/bin/env python3
"""Experiment: what the march kernel's sigma gather costs.  `build` writes three variants of the
library (text substitution in march_rec_kernel): A as is, B no sigma gather (every valid leaf is
recorded), C the gather reads a compact [M] column's access pattern (values are garbage: timing
only).  `run` times march_rec_kernel in each (HIP events around volume_render minus nothing --
use the rocprof line for kernel-only numbers) -- GPU box."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "svox_t_amd", "csrc")
V = {"A": None,
     "B": ("p_sigma = sig_col[(int64_t)s.idx * K];", "p_sigma = 1.f;"),
     "C": ("p_sigma = sig_col[(int64_t)s.idx * K];", "p_sigma = tr.features[s.idx];")}

def build():
    from _flatten import flat_source
    src0 = flat_source()
    for name, sub in V.items():
        src = src0
        if sub:
            assert sub[0] in src
            src = src.replace(sub[0], sub[1])
        tmp = os.path.join(CSRC, "_var_kernels.hip")
        open(tmp, "w").write(src)
        out = os.path.join(ROOT, "exp", f"libsvoxt_var{name}.so")
        try:
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                                   "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function", "-o", out, tmp,
                                   os.path.join(CSRC, "svoxt_build.hip"), os.path.join(CSRC, "svoxt_motion.hip"),
                                   os.path.join(CSRC, "svoxt_order.hip")])
        finally:
            os.remove(tmp)
        print(out)

def run():
    # one subprocess per variant / grid size (the library is loaded at import)
    for name in V:
        for g in ("7",) if os.environ.get("VAR_WORKLOAD") == "d9" else ("7", "6"):
            env = dict(os.environ, SVOXT_LIB=os.path.join(ROOT, "exp", f"libsvoxt_var{name}.so"), SVOXT_ACCEL_LOG2=g)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True, timeout=120)
            print(f"variant {name} grid 2^{g}: {r.stdout.strip()} {r.stderr.strip()[-200:] if r.returncode else ''}", flush=True)

def one():
    sys.path.insert(0, ROOT)
    import torch, ctypes
    import svox_t_amd as svox, svox_t_amd.csrc as _C
    from svox_t_amd import synth
    from svox_t_amd.renderer import _rays_spec_from_rays
    dev = torch.device("cuda:0")
    d9 = os.environ.get("VAR_WORKLOAD") == "d9"      # BASELINE configs[3]: the feature table leaves the Infinity Cache
    st = synth.shell_tree(9 if d9 else 8)
    feats = synth.shell_features(st.n_features, 32 if d9 else 28)
    tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format="RGBA" if d9 else "SH9", device=dev)
    r = svox.VolumeRenderer(tree)
    W = H = 1024 if d9 else 800
    o, d, v = synth.pinhole_rays(W, H)
    rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
    spec = tree._spec(tree.features); rsh = _rays_spec_from_rays(rays, (H, W)); opt = r._get_options()
    # the march kernel alone: call the C ABI pieces through a forward, time with events, and
    # subtract nothing -- instead run the forward with SVOXT_FWD_SPLIT=1 and report the whole
    # forward; the march share is what differs between the variants
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    os.environ.setdefault("SVOXT_FWD_SPLIT", "1")
    ms = timeit(lambda: _C.volume_render(spec, rsh, opt))
    print(f"split forward {ms:.4f} ms")

if __name__ == "__main__":
    {"build": build, "run": run, "one": one}[sys.argv[1]]()
