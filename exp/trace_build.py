#!/usr/bin/env python3
"""Experiment: a build of the library whose render_fwd_kernel / render_bwd_kernel /
grad_merge_kernel stamp every workgroup's start / end time (wall_clock64, 100 MHz) and
hardware id into a device array; read with svoxt_trace_read(slot).  Writes
exp/libsvoxt_trace.so; use with SVOXT_LIB=exp/libsvoxt_trace.so (exp/trace_run.py)."""
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "svox_t_amd", "csrc")
from _flatten import flat_source
src = flat_source()
pre = '''
constexpr int kTraceN = 65536;
__device__ unsigned long long g_trace[3][3 * kTraceN];
struct TraceScope {
    int slot;
    __device__ __forceinline__ TraceScope(int s) : slot(s) {
        const unsigned b = blockIdx.y * gridDim.x + blockIdx.x;
        if (threadIdx.x == 0 && b < kTraceN) {
            g_trace[slot][3 * b] = wall_clock64();
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_trace[slot][3 * b + 2] = ((unsigned long long)xcc << 32) | hw;
        }
    }
    __device__ __forceinline__ ~TraceScope() {
        const unsigned b = blockIdx.y * gridDim.x + blockIdx.x;
        if (b < kTraceN) atomicMax(&g_trace[slot][3 * b + 1], (unsigned long long)wall_clock64());
    }
};
'''
marker = "// XF (SH only): per-leaf view rotations (tree.xform): the basis is re-evaluated"
assert marker in src
src = src.replace(marker, pre + marker, 1)
for slot, head in enumerate(("render_fwd_kernel(TreeDev tr, RaysDev rays, Opts opt, float* __restrict__ out,",
                             "render_bwd_kernel(TreeDev tr, RaysDev rays, Opts opt, const float* __restrict__ grad_out,",
                             "grad_fused_kernel(TreeDev tr, RaysDev rays, Opts opt, const float* __restrict__ grad_out,")):
    a = src.index(head)
    b = src.index(") {\n", a) + 4
    src = src[:b] + f"    TraceScope trace_scope({slot});\n" + src[b:]
src += '''
extern "C" int svoxt_trace_read(int slot, void* host_out, int reset) {
    hipDeviceSynchronize();
    const size_t n = sizeof(unsigned long long) * 3 * svoxt::kTraceN;
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(svoxt::g_trace), n, n * slot) != hipSuccess) return 1;
    if (reset) { static unsigned long long z[3 * svoxt::kTraceN]; if (hipMemcpyToSymbol(HIP_SYMBOL(svoxt::g_trace), z, n, n * slot) != hipSuccess) return 1; }
    return 0;
}
'''
tmp = os.path.join(CSRC, "_trace_kernels.hip")
open(tmp, "w").write(src)
out = os.path.join(ROOT, "exp", "libsvoxt_trace.so")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
       "-fno-fast-math", "-Wno-unused-function", "-Wno-unused-value", "-o", out, tmp,
       os.path.join(CSRC, "svoxt_build.hip"), os.path.join(CSRC, "svoxt_motion.hip"),
       os.path.join(CSRC, "svoxt_order.hip")]
try:
    subprocess.check_call(cmd)
finally:
    os.remove(tmp)
print(out)
