#!/usr/bin/env python3
"""Experiment: where grad_fused_kernel's time goes, phase by phase.  `build` writes
exp/libsvoxt_trace.so -- the library with thread 0 of every workgroup of grad_fused_kernel adding the
shader-clock cycles it spends in each phase (setup | sweep 1 | sweep 2: terms + advance | scan wait |
sort scatter | reduce | table clear) to a device array; `run` (GPU box) runs the headline backward and
prints each phase's share of a workgroup's life."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "svox_t_amd", "csrc")
OUT = os.path.join(ROOT, "exp", "libsvoxt_trace.so")
PHASES = ["setup", "sweep 1 (terms of 7 positions per round, wavefront 0 along the rays)",
          "sweep 2: terms + hash insert + advance", "sweep 2: wait for scan", "sweep 2: counting-sort scatter",
          "sweep 2: reduce (expand, stage, sum, atomics)", "sweep 2: table clear"]


def build():
    from _flatten import flat_source
    src = flat_source()
    a = src.index("grad_fused_kernel(TreeDev tr")
    b = src.index("// The backward of an image for RGBA-style rows of 8 / 16 / 32 floats", a)
    body = src[a:b]

    def sub(old, new, count=1):
        nonlocal body
        assert body.count(old) == count, (body.count(old), old)
        body = body.replace(old, new)

    ph = lambda i: f"    if (threadIdx.x == 0) {{ const unsigned long long now_ = clock64(); atomicAdd(&g_ph[{i}], now_ - tprev_); tprev_ = now_; }}\n"
    sub("    if (maxn == 0) return;                                   // the same in every wavefront of the workgroup\n",
        "    if (maxn == 0) return;\n    unsigned long long tprev_ = clock64();\n    if (threadIdx.x == 0) atomicAdd(&g_ph[7], 1ull);\n")
    sub("    if constexpr (EXACT) {\n        // ---- sweep 1", ph(0) + "    if constexpr (EXACT) {\n        // ---- sweep 1")
    sub("    int kb = 0;\n    while (kb < maxn) {\n", ph(1) + "    int kb = 0;\n    while (kb < maxn) {\n")
    sub("    if (wave == 1) {\n        constexpr int PER = T / 64;", ph(2) + "    if (wave == 1) {\n        constexpr int PER = T / 64;")
    sub("    const int nb = __builtin_amdgcn_readfirstlane(s_nb);", ph(3) + "    const int nb = __builtin_amdgcn_readfirstlane(s_nb);")
    sub("    lds_barrier();\n    // ---- reduce: 64 sorted records at a time per wavefront",
        "    lds_barrier();\n" + ph(4) + "    // ---- reduce: 64 sorted records at a time per wavefront")
    sub("        if (kb >= maxn) break;                                   // last pass (scalar condition)\n",
        ph(5) + "        if (kb >= maxn) break;\n")
    sub("        for (int i = threadIdx.x; i < T; i += NT) { keys[i] = -1; cnt[i] = 0; }\n        lds_barrier();\n    }\n}",
        "        for (int i = threadIdx.x; i < T; i += NT) { keys[i] = -1; cnt[i] = 0; }\n        lds_barrier();\n" + ph(6) + "    }\n}")
    # the device array goes in front of the kernel's template line
    t = src.rindex("template <int FMT, int BD, bool EXACT, bool COUNT = false, int TERMS = 0>", 0, a)
    src = src[:t] + "__device__ unsigned long long g_ph[8];\n" + src[t:a] + body + src[b:]
    src += '''
extern "C" int svoxt_phase_read(void* host_out, int reset) {
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(svoxt::g_ph), 64, 0) != hipSuccess) return 1;
    if (reset) { static unsigned long long z[8]; if (hipMemcpyToSymbol(HIP_SYMBOL(svoxt::g_ph), z, 64, 0) != hipSuccess) return 1; }
    return 0;
}
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
    dev = torch.device("cuda:0")
    st = synth.shell_tree(8)
    feats = synth.shell_features(st.n_features, 28)
    tree = svox.N3Tree.from_arrays(st.child, st.data, st.parent_depth, feats, data_format="SH9", device=dev)
    r = svox.VolumeRenderer(tree)
    W = H = 800
    o, d, v = synth.pinhole_rays(W, H)
    rays = svox.Rays(o.to(dev), d.to(dev), v.to(dev))
    g = synth.grad_output(W * H, 4).to(dev)
    lib = ctypes.CDLL(OUT)
    buf = (ctypes.c_ulonglong * 8)()
    for it in range(6):
        tree.features.grad = None
        out = r(tree.features, rays, image_shape=(H, W))
        out.backward(g)
        torch.cuda.synchronize()
        if it == 2:
            lib.svoxt_phase_read(buf, 1)          # warm-up done: reset
    lib.svoxt_phase_read(buf, 0)
    ph = np.array(list(buf), dtype=np.float64)
    wgs, cyc = ph[7], ph[:7]
    print(_C.LAST_ROUTE["backward"])
    print(f"workgroups with samples: {wgs / 3:.0f} per launch; cycles per workgroup {cyc.sum() / wgs:.0f} "
          f"(shader clock; ~{cyc.sum() / wgs / 2.1e3:.2f} us at 2.1 GHz)")
    for name, c in zip(PHASES, cyc):
        print(f"  {c / cyc.sum():6.1%}  {c / wgs:8.0f} cycles  {name}")


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
