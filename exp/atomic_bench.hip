// Microbenchmark: rate of float atomic adds shaped like the backward's flush
// (two 112-byte row segments per wave-instruction, random rows of a 75 MB table),
// for different memory scopes / row strides.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
template <int SCOPE, int STRIDE>
__global__ void k(float* g, const int* rows, int nrounds, int M) {
    const int lane = threadIdx.x & 63, half = lane >> 5, j = lane & 31;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int r = 0; r < nrounds; ++r) {
        const int row = rows[(wave * nrounds + r) * 2 + half];
        if (j < 28) __hip_atomic_fetch_add(g + (size_t)row * STRIDE + j, 1.0f, __ATOMIC_RELAXED, SCOPE);
    }
}
int main() {
    const int M = 668912, nw = 10000, nr = 300;
    float* g; int* rows;
    hipMalloc(&g, (size_t)M * 32 * 4); hipMemset(g, 0, (size_t)M * 32 * 4);
    std::vector<int> h((size_t)nw * nr * 2);
    srand(1); for (auto& x : h) x = rand() % M;
    hipMalloc(&rows, h.size() * 4); hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern) {
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(nw / 4), dim3(256), 0, 0, g, rows, nr, M);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it == 2) printf("%-28s %.3f ms  %.2f M wave-atomics, %.1f ns/instr/CU, %.2f TB/s useful\n", name, ms,
                                nw * (double)nr / 1e6, ms * 1e6 / (nw * (double)nr / 256), nw * (double)nr * 224 / ms / 1e9);
        }
    };
    run("agent scope, stride 28", k<__HIP_MEMORY_SCOPE_AGENT, 28>);
    run("agent scope, stride 32", k<__HIP_MEMORY_SCOPE_AGENT, 32>);
    run("workgroup scope, stride 28", k<__HIP_MEMORY_SCOPE_WORKGROUP, 28>);
    run("workgroup scope, stride 32", k<__HIP_MEMORY_SCOPE_WORKGROUP, 32>);
    run("wavefront scope, stride 32", k<__HIP_MEMORY_SCOPE_WAVEFRONT, 32>);
    run("system scope, stride 28", k<__HIP_MEMORY_SCOPE_SYSTEM, 28>);
    return 0;
}
