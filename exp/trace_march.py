#!/usr/bin/env python3
"""Experiment: timeline of march_rec_kernel's wavefronts.  `build` writes exp/libsvoxt_trace.so
(the library with march_rec_kernel stamping wall_clock64 -- 100 MHz -- at its start, at every
16th crossing of its longest lane and at its end); `run` (GPU box, SVOXT_LIB set by this
script) prints how long 16 crossings take as a function of when they happen."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "svox_t_amd", "csrc")
OUT = os.path.join(ROOT, "exp", "libsvoxt_trace.so")
NS = 16   # stamps per workgroup

def build():
    from _flatten import flat_source
    src = flat_source()
    pre = f'''
constexpr int kTraceN = 16384;
__device__ unsigned long long g_mtrace[kTraceN * {NS}];
'''
    marker = "// STOP: apply the early-termination rule"
    assert marker in src
    src = src.replace(marker, pre + marker, 1)
    a = src.index("march_rec_kernel(TreeDev tr, RaysDev rays, Opts opt, RecLists L, uint4* __restrict__ aux,")
    b = src.index("{\n", a) + 2
    src = src[:b] + f'''    unsigned long long* mt = g_mtrace + (size_t)(blockIdx.x < kTraceN ? blockIdx.x : 0) * {NS};
    if (threadIdx.x == 0) mt[0] = wall_clock64();
    int trace_it = 0;
''' + src[b:]
    # stamp inside the loop (first statement of the while body of this kernel)
    w = src.index("    while (t < r.tmax) {\n        Sample s;\n        march_step<N2, ACC>", b)
    w2 = src.index("{\n", w) + 2
    src = src[:w2] + f'''        ++trace_it;
        if ((trace_it & 15) == 0 && (trace_it >> 4) < {NS - 2}) atomicMax(mt + 1 + (trace_it >> 4), (unsigned long long)wall_clock64());
''' + src[w2:]
    e = src.index("    aux[q] = make_uint4((uint32_t)nrec | over, __float_as_uint(t_resume), __float_as_uint(1.f), 0u);\n}", w)
    src = src[:e] + f"    atomicMax(mt + 1, (unsigned long long)wall_clock64());\n    atomicMax(mt + {NS - 1}, (unsigned long long)trace_it);\n" + src[e:]
    src += f'''
extern "C" int svoxt_mtrace_read(void* host_out, int reset) {{
    hipDeviceSynchronize();
    const size_t n = sizeof(unsigned long long) * svoxt::kTraceN * {NS};
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(svoxt::g_mtrace), n, 0) != hipSuccess) return 1;
    if (reset) {{ static unsigned long long z[svoxt::kTraceN * {NS}]; if (hipMemcpyToSymbol(HIP_SYMBOL(svoxt::g_mtrace), z, n, 0) != hipSuccess) return 1; }}
    return 0;
}}
'''
    tmp = os.path.join(CSRC, "_trace_kernels.hip")
    open(tmp, "w").write(src)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-fno-fast-math", "-Wno-unused-function", "-Wno-unused-value", "-o", OUT, tmp,
           os.path.join(CSRC, "svoxt_build.hip"), os.path.join(CSRC, "svoxt_motion.hip"), os.path.join(CSRC, "svoxt_order.hip")]
    try:
        subprocess.check_call(cmd)
    finally:
        os.remove(tmp)
    print(OUT)

def run():
    os.environ["SVOXT_LIB"] = OUT
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import svox_t_amd as svox, svox_t_amd.csrc as _C
    from svox_t_amd import synth
    from svox_t_amd.renderer import _rays_spec_from_rays
    dev = torch.device("cuda:0")
    st = synth.shell_tree(8)
    feats = synth.shell_features(st.n_features, 28)
    tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format="SH9", device=dev)
    r = svox.VolumeRenderer(tree)
    W = H = 800
    o, d, v = synth.pinhole_rays(W, H)
    rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
    spec = tree._spec(tree.features); rsh = _rays_spec_from_rays(rays, (H, W)); opt = r._get_options()
    lib = _C._lib
    buf = np.zeros((16384, NS), dtype=np.uint64)
    for _ in range(3): _C.volume_render(spec, rsh, opt)
    torch.cuda.synchronize()
    lib.svoxt_mtrace_read(buf.ctypes.data_as(ctypes.c_void_p), 1)
    _C.volume_render(spec, rsh, opt)
    torch.cuda.synchronize()
    lib.svoxt_mtrace_read(buf.ctypes.data_as(ctypes.c_void_p), 1)
    t = buf[:10000].astype(np.int64)
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / 100.0; end = (t[:, 1] - t0) / 100.0; n = t[:, NS - 1]
    dur = end - start
    print(f"kernel span {end.max():.1f} us, last start {start.max():.1f} us, sum of durations {dur.sum()/1e3:.1f} ms, crossings sum {n.sum()} max {n.max()}")
    print("resident waves every 10 us:", " ".join(str(int(np.sum((start <= g) & (end > g)))) for g in np.arange(0, end.max(), 10.0)))
    # us per crossing, by when it happens: for every stamped 16-crossing interval
    marks = (t[:, 2:NS - 1] - t0) / 100.0        # stamp k (k>=1) = time after 16k crossings
    prev = np.concatenate([start[:, None], marks[:, :-1]], 1)
    ok = t[:, 2:NS - 1] > 0
    mid = 0.5 * (marks + prev)[ok]; per = ((marks - prev) / 16.0)[ok]
    for lo in range(0, 260, 20):
        sel = (mid >= lo) & (mid < lo + 20)
        if sel.any(): print(f"  t in [{lo:3d},{lo+20:3d}) us: {per[sel].mean():.3f} us per crossing over {int(sel.sum())} intervals (p10 {np.percentile(per[sel],10):.3f}, p90 {np.percentile(per[sel],90):.3f})")
    for i in np.argsort(-dur)[:5]:
        print(f"  wave {i}: start {start[i]:.1f} end {end[i]:.1f} crossings {n[i]} -> {dur[i]/max(n[i],1):.3f} us each; marks", " ".join(f"{x:.0f}" for x in marks[i][ok[i]]))

if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
