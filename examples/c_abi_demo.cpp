// c_abi_demo.cpp -- libsvoxt_hip.so driven through its C ABI alone (include/svoxt.h):
// no Python, no torch; device memory from hipMalloc, one stream.
//
//   hipcc --offload-arch=gfx950 -O2 -I include examples/c_abi_demo.cpp \
//         -L svox_t_amd/csrc -lsvoxt_hip -Wl,-rpath,$PWD/svox_t_amd/csrc -o c_abi_demo
//   ./c_abi_demo in.bin out.bin
//
// in.bin  (written by tests/test_gpu_c_abi_demo.py): int64 header {n_internal, M, K, Q,
//         format, basis_dim, width, height}, then child int32[n*8], data int32[n*8],
//         offset f32[3], scaling f32[3], features f32[M*K], origins / dirs / vdirs
//         f32[Q*3] each, grad_out f32[Q*cols]
// out.bin: int64 cols, then out f32[Q*cols], depth f32[Q], grad f32[M*K]
//
// What it shows: the call sequence a C or C++ host needs -- svoxt_accel_build (optional),
// svoxt_step_plan + svoxt_step_forward + svoxt_step_backward (the fast training pair, routed by
// the library: ABI v21), svoxt_render_depth -- with every buffer owned by the caller.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "svoxt.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define SVOXT_OK_(x) do { int rc_ = (x); if (rc_ != SVOXT_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, svoxt_last_error()); return 3; } } while (0)

template <typename T>
static bool read_vec(std::FILE* f, std::vector<T>& v, size_t n) {
    v.resize(n);
    return n == 0 || std::fread(v.data(), sizeof(T), n, f) == n;
}

template <typename T>
static T* to_device(const std::vector<T>& v) {
    T* p = nullptr;
    if (hipMalloc(&p, v.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
    if (!v.empty() && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return p;
}

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 1; }
    if (svoxt_abi_version() != SVOXT_ABI_VERSION) { std::fprintf(stderr, "header / library ABI mismatch\n"); return 1; }
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 1; }
    int64_t h[8];
    if (std::fread(h, sizeof(int64_t), 8, f) != 8) return 1;
    const int64_t n = h[0], M = h[1], K = h[2], Q = h[3];
    std::vector<int32_t> child, data;
    std::vector<float> offset, scaling, features, origins, dirs, vdirs, grad_out;
    svoxt_options opt = {};
    opt.step_size = 1e-3f; opt.background_brightness = 1.0f;
    opt.format = (int32_t)h[4]; opt.basis_dim = (int32_t)h[5];
    opt.ndc_width = -1; opt.min_comp = 0; opt.max_comp = opt.basis_dim - 1;
    const int cols = svoxt_out_data_dim(&opt, (int32_t)K);
    if (!read_vec(f, child, n * 8) || !read_vec(f, data, n * 8) || !read_vec(f, offset, 3) || !read_vec(f, scaling, 3) ||
        !read_vec(f, features, M * K) || !read_vec(f, origins, Q * 3) || !read_vec(f, dirs, Q * 3) ||
        !read_vec(f, vdirs, Q * 3) || !read_vec(f, grad_out, Q * cols)) { std::fprintf(stderr, "short input\n"); return 1; }
    std::fclose(f);

    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    svoxt_tree tree = {};
    tree.features = to_device(features); tree.M = M; tree.K = (int32_t)K; tree.N = 2;
    tree.data = to_device(data); tree.child = to_device(child); tree.n_internal = n;
    tree.offset = to_device(offset); tree.scaling = to_device(scaling);
    svoxt_rays rays = {};
    rays.origins = to_device(origins); rays.dirs = to_device(dirs); rays.vdirs = to_device(vdirs); rays.Q = Q;
    rays.image_width = (int32_t)h[6]; rays.image_height = (int32_t)h[7];
    float* d_gout = to_device(grad_out);

    // optional: the acceleration grid (a cache of the descent; results do not depend on it) -- in 4 x 4 x 4 bricks, the
    // layout for a grid that training steps render through (SVOXT_ACCEL_BRICKS)
    const int g = 5 | SVOXT_ACCEL_BRICKS;
    void* cells = nullptr;
    HIP_OK(hipMalloc(&cells, svoxt_accel_bytes(g, tree.n_internal)));
    SVOXT_OK_(svoxt_accel_build(&tree, g, cells, st));
    tree.accel = cells; tree.accel_log2 = g;

    float *d_out = nullptr, *d_depth = nullptr, *d_grad = nullptr;
    HIP_OK(hipMalloc(&d_out, sizeof(float) * Q * cols));
    HIP_OK(hipMalloc(&d_depth, sizeof(float) * Q));
    HIP_OK(hipMalloc(&d_grad, sizeof(float) * M * K));

    // One training step: the library plans the route for this payload (sample lists from a pool, the sigma bitmask,
    // march and shade in one launch, the forward's hand-over, the per-tile exact backward, padded gradient rows --
    // whatever applies) and lays out ONE workspace; the host allocates it and makes two calls.  (The mechanisms
    // underneath -- svoxt_volume_render_fwd_record / _bwd_replay and friends -- remain callable one by one.)
    svoxt_step step;
    SVOXT_OK_(svoxt_step_plan(&tree, &rays, &opt, /*pool_blocks: first guess*/ 0, &step));
    void* workspace = nullptr;
    HIP_OK(hipMalloc(&workspace, (size_t)step.workspace_bytes));          // (hipMalloc: 256-byte aligned and more)
    SVOXT_OK_(svoxt_step_forward(&tree, &rays, &opt, d_out, &step, workspace, st));
    SVOXT_OK_(svoxt_step_backward(&tree, &rays, &opt, d_gout, d_grad, &step, workspace, st));
    if (step.records) {
        // how much of the list pool the batch took: the next plan can ask for that much (+ a margin) instead of the guess
        std::vector<int32_t> ctr(32 * 16);
        HIP_OK(hipMemcpyAsync(ctr.data(), step.lists.pool_next, ctr.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        int32_t mx = -1;
        for (int i = 0; i < 32; ++i) mx = ctr[16 * i] > mx ? ctr[16 * i] : mx;
        std::printf("c_abi_demo: list pool %lld blocks, used %lld; workspace %.1f MiB\n", (long long)step.lists.pool_blocks,
                    (long long)(mx + 1) * 32, step.workspace_bytes / 1048576.0);
    }
    SVOXT_OK_(svoxt_render_depth(&tree, &rays, &opt, d_depth, st));
    HIP_OK(hipStreamSynchronize(st));

    std::vector<float> out(Q * cols), depth(Q), grad(M * K);
    HIP_OK(hipMemcpy(out.data(), d_out, out.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(depth.data(), d_depth, depth.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(grad.data(), d_grad, grad.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::FILE* o = std::fopen(argv[2], "wb");
    if (!o) { std::perror(argv[2]); return 1; }
    const int64_t c64 = cols;
    std::fwrite(&c64, sizeof(c64), 1, o);
    std::fwrite(out.data(), sizeof(float), out.size(), o);
    std::fwrite(depth.data(), sizeof(float), depth.size(), o);
    std::fwrite(grad.data(), sizeof(float), grad.size(), o);
    std::fclose(o);
    std::printf("c_abi_demo: %lld rays, %d columns, %lld x %lld gradient\n", (long long)Q, cols, (long long)M, (long long)K);
    return 0;
}
