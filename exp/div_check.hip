// exp/div_check.hip -- div_unit_range / rcp_unit_range (svoxt_device.h) against the `/` operator, exhaustively in the
// denominator: every float e >= 0 (all 2^31 - 2^23 + 1 bit patterns up to +inf) as d = 1.0 + double(e), against a set of
// numerators that covers the operand range of the render path -- 0, the smallest denormal float, the smallest normal,
// 1 - 2^-24, 1, and pseudo-random floats in (0, 1] -- bit for bit.  Prints the number of mismatches (0) and exits
// non-zero otherwise.  Build and run on the GPU box:
//     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I svox_t_amd/csrc -o /tmp/div_check exp/div_check.hip && /tmp/div_check
// (tests/test_gpu_div_exact.py does exactly that.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "svoxt_device.h"

constexpr int kW = 24;

__global__ void check(const float* __restrict__ w, unsigned long long* __restrict__ bad, uint32_t e_lo, uint32_t e_n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e_n) return;
    const float e = __uint_as_float(e_lo + i);
    const double d = 1.0 + (double)e;
    unsigned long long nbad = 0;
#pragma unroll 1
    for (int j = 0; j < kW; ++j) {
        const double n = (double)w[j];
        const double a = svoxt::div_unit_range(n, d), b = n / d;
        nbad += __double_as_longlong(a) != __double_as_longlong(b) ? 1ull : 0ull;
        // ... and through the expression the kernels evaluate: acc = float(double(acc) + q)
        const float fa = (float)((double)w[(j + 7) % kW] + a), fb = (float)((double)w[(j + 7) % kW] + b);
        nbad += __float_as_uint(fa) != __float_as_uint(fb) ? 1ull : 0ull;
    }
    const double a = svoxt::rcp_unit_range(d), b = 1.0 / d;
    nbad += __double_as_longlong(a) != __double_as_longlong(b) ? 1ull : 0ull;
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    float w[kW] = {0.f, 1.401298464e-45f, 1.17549435e-38f, 0.99999994f, 1.f, 0.5f, 0.25f, 1e-3f, 1e-7f, 1e-20f, 1e-30f, 3e-39f};
    uint32_t s = 12345u;
    for (int j = 12; j < kW; ++j) {             // pseudo-random floats in (0, 1]
        s = s * 1664525u + 1013904223u;
        w[j] = (float)((s >> 8) + 1u) / 16777216.f;
    }
    float* dw; unsigned long long* dbad;
    if (hipMalloc(&dw, sizeof(w)) != hipSuccess || hipMalloc(&dbad, 8) != hipSuccess) { fprintf(stderr, "no GPU\n"); return 2; }
    hipMemcpy(dw, w, sizeof(w), hipMemcpyHostToDevice);
    hipMemset(dbad, 0, 8);
    const uint64_t total = 0x7f800000ull + 1ull;          // +0 ... +inf
    const uint32_t chunk = 1u << 28;
    for (uint64_t lo = 0; lo < total; lo += chunk) {
        const uint32_t n = (uint32_t)((total - lo) < chunk ? (total - lo) : chunk);
        hipLaunchKernelGGL(check, dim3((n + 255) / 256), dim3(256), 0, 0, dw, dbad, (uint32_t)lo, n);
    }
    unsigned long long bad = ~0ull;
    if (hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 2; }
    printf("div_check: %llu denominators x %d numerators (+ the reciprocal): %llu mismatches\n", (unsigned long long)total, kW, bad);
    return bad == 0 ? 0 : 1;
}
