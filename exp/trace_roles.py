#!/usr/bin/env python3
"""Where fwd_roles_kernel's time goes (NOTEBOOK.md step 33): `build` writes exp/libsvoxt_rtrace.so, the library
with three time stamps added to the kernel (wall_clock64, 100 MHz): the first workgroup's start, the end of the
LAST march, the end of the last shade -- plus the end of the last shade among the tiles whose march ended in the
first half of the marching phase; `run` (GPU box) runs the headline's recording forward and prints them."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "svox_t_amd", "csrc")
OUT = os.path.join(ROOT, "exp", "libsvoxt_rtrace_noshade.so" if os.environ.get("RT_NOSHADE") else "libsvoxt_rtrace.so")


def build():
    sys.path.insert(0, os.path.join(ROOT, "exp"))
    from _flatten import flat_source
    src = flat_source()

    def sub(old, new):
        nonlocal src
        assert src.count(old) == 1, (src.count(old), old)
        src = src.replace(old, new)

    # per-tile stamps, no shared counters (20 000 atomics on one address cost more than the kernel)
    sub("constexpr int kRolePolls = 20000;\n", "constexpr int kRolePolls = 20000;\n__device__ unsigned long long g_rt[3 * 16384];\n")
    sub("        const uint32_t ax = march_rec_tile<true, false, ACC, true>(tr, rays, opt, L, aux, sigma_mask, tile, rstage, ltab);\n",
        "        if ((threadIdx.x & 63) == 0 && tile < 16384) g_rt[tile] = wall_clock64();\n"
        "        const uint32_t ax = march_rec_tile<true, false, ACC, true>(tr, rays, opt, L, aux, sigma_mask, tile, rstage, ltab);\n")
    sub("        __hip_atomic_store(queue + pos, (int32_t)tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n",
        "        __hip_atomic_store(queue + pos, (int32_t)tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
        "            if ((tile & 0xffff) < 16384) g_rt[16384 + (tile & 0xffff)] = wall_clock64();\n")
    sub("    if (threadIdx.x == 0) tile_state[tile] = kTileShaded;        // (read by the fallback launch: after this kernel)\n",
        "    if (threadIdx.x == 0) { tile_state[tile] = kTileShaded; if (tile < 16384) g_rt[2 * 16384 + tile] = wall_clock64(); }\n")
    if os.environ.get("RT_NOSHADE"):
        # the shading workgroups leave at once (the fallback launch shades every tile): how long do the marches take
        # in this grid when nothing runs beside them?
        sub("    if (b >= mine) return;\n", "    if (b >= mine) return;\n    if (ntiles > 0) return;\n")
    src += '''
extern "C" int svoxt_rt_read(void* host_out, int reset) {
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(svoxt::g_rt), 3 * 16384 * 8, 0) != hipSuccess) return 1;
    if (reset) { static unsigned long long z[3 * 16384]; if (hipMemcpyToSymbol(HIP_SYMBOL(svoxt::g_rt), z, sizeof(z), 0) != hipSuccess) return 1; }
    return 0;
}
'''
    tmp = os.path.join(CSRC, "_rtrace_kernels.hip")
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
    import torch
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
    lib = ctypes.CDLL(OUT)
    import numpy as np
    buf = np.zeros(3 * 16384, dtype=np.uint64)
    ptr = buf.ctypes.data_as(ctypes.c_void_p)
    for it in range(8):
        out = r(tree.features, rays, image_shape=(H, W))
        torch.cuda.synchronize()
        if it < 7:
            lib.svoxt_rt_read(ptr, 1)      # reset before every forward; the last one is read
        del out
    lib.svoxt_rt_read(ptr, 0)
    T = 10000
    ms, me, se = (buf[i * 16384:i * 16384 + T].astype(np.float64) / 100.0 for i in range(3))    # us (100 MHz)
    t0 = ms.min()
    ms, me = ms - t0, me - t0
    shaded = se > 0                        # (r03: a tile without samples is finished by its march and never shaded)
    print(f"tiles shaded by a shading workgroup: {int(shaded.sum())} of {T}")
    print(_C.LAST_ROUTE["forward"])
    print(f"march starts: 50 % by {np.percentile(ms, 50):6.1f} us, 90 % by {np.percentile(ms, 90):6.1f}, last {ms.max():6.1f}")
    print(f"march ends  : 50 % by {np.percentile(me, 50):6.1f} us, 90 % by {np.percentile(me, 90):6.1f}, 99 % by {np.percentile(me, 99):6.1f}, last {me.max():6.1f}")
    dur = me - ms
    print(f"march duration: median {np.median(dur):6.1f} us, 99 % {np.percentile(dur, 99):6.1f}, longest {dur.max():6.1f} (started at {ms[dur.argmax()]:.1f})")
    if se.max() > 0:
        se, me = se[shaded] - t0, me[shaded]
        lag = se - me
        print(f"marches of the shaded tiles end: 50 % by {np.percentile(me, 50):6.1f} us, 90 % by {np.percentile(me, 90):6.1f}, last {me.max():6.1f}")
        print(f"shade ends  : 50 % by {np.percentile(se, 50):6.1f} us, 90 % by {np.percentile(se, 90):6.1f}, last {se.max():6.1f}")
        print(f"march end -> shade end of the same tile: median {np.median(lag):6.1f} us, 90 % {np.percentile(lag, 90):6.1f}, max {lag.max():6.1f}; "
              f"the last-shaded tile's march ended at {me[se.argmax()]:.1f}")


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
