// svoxt_order.hip -- ray ordering for batches that are not images.
//
// The marching kernels put 64 consecutive rays of a batch on one wavefront.  For an image
// walked in 8x8 tiles those rays cross the same leaves; for a batch in arbitrary order
// (training on rays drawn at random from many cameras) every lane goes its own way: each
// step's loads hit 64 different cache lines and the wavefront lasts as long as its longest
// ray.  svoxt_ray_order gives the permutation that sorts a batch by the Morton code of the
// point where each ray enters the tree's cube (10 bits per axis; rays that miss the cube
// last).  The caller gathers origins / dirs / viewdirs with it, renders the sorted batch
// (any entry point of include/svoxt.h) and scatters the output rows back -- results are per
// ray and do not depend on the order; 800x800 shuffled rays of one camera, depth-8 SH9 tree:
// forward 0.82 -> 0.31 ms, forward+backward 1.65 -> 1.2 ms (0.85 with the two-kernel
// backward); rays drawn from 8 cameras: 1.70 -> 1.35 ms.
//
// No counterpart in the reference (which marches rays in the order given).
// The sort itself is rocPRIM's radix sort (the vendor primitive; keys + permutation).

#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "svoxt_host.h"

namespace svoxt {

constexpr int kOrderBlock = 256;

__device__ __forceinline__ uint32_t spread10(uint32_t x) {      // 10 bits -> every third bit
    x &= 0x3ffu;
    x = (x | (x << 16)) & 0x30000ffu;
    x = (x | (x << 8)) & 0x300f00fu;
    x = (x | (x << 4)) & 0x30c30c3u;
    x = (x | (x << 2)) & 0x9249249u;
    return x;
}

__global__ void __launch_bounds__(kOrderBlock)
ray_key_kernel(TreeDev tr, RaysDev rays, Opts opt, uint32_t* __restrict__ keys, int32_t* __restrict__ ids) {
    const int64_t q = (int64_t)blockIdx.x * kOrderBlock + threadIdx.x;
    if (q >= rays.Q) return;
    Ray r;
    uint32_t key = 0xffffffffu;                      // misses go last (their wavefronts end at once)
    if (setup_ray(tr, rays, opt, q, r)) {
        const float t = r.tmin;                      // >= 0: the origin itself when it lies inside
        const float px = fminf(fmaxf(r.ox + t * r.dx, 0.f), kClampHi);
        const float py = fminf(fmaxf(r.oy + t * r.dy, 0.f), kClampHi);
        const float pz = fminf(fmaxf(r.oz + t * r.dz, 0.f), kClampHi);
        key = (spread10((uint32_t)(px * 1024.f)) << 2) | (spread10((uint32_t)(py * 1024.f)) << 1) |
              spread10((uint32_t)(pz * 1024.f));
    }
    keys[q] = key;
    ids[q] = (int32_t)q;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace svoxt

using namespace svoxt;

extern "C" {

int64_t svoxt_ray_order_workspace_bytes(int64_t Q) {
    if (Q < 0 || Q > 0x7fffffff) return -1;
    size_t temp = 0;
    uint32_t* k = nullptr;
    int32_t* v = nullptr;
    if (Q > 0 && rocprim::radix_sort_pairs(nullptr, temp, k, k, v, v, (size_t)Q, 0, 32, (hipStream_t)0) != hipSuccess)
        return -1;
    return (int64_t)(3 * align256(sizeof(uint32_t) * (size_t)Q) + align256(temp));
}

int svoxt_ray_order(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt,
                    int32_t* perm, void* workspace, int64_t workspace_bytes, void* stream) {
    const char* fn = "svoxt_ray_order";
    int rc;
    if ((rc = check_tree(tree, fn)) || (rc = check_rays(rays, fn)) || (rc = check_opts(opt, tree, fn, false))) return rc;
    if (rays->Q == 0) return SVOXT_OK;
    if (perm == nullptr || workspace == nullptr) return set_error(SVOXT_ERR_INVALID, "%s: perm or workspace is NULL", fn);
    if (rays->image_width > 0) return set_error(SVOXT_ERR_INVALID, "%s: the batch is declared an image (walked in tiles already)", fn);
    const int64_t need = svoxt_ray_order_workspace_bytes(rays->Q);
    if (need < 0 || workspace_bytes < need) return set_error(SVOXT_ERR_INVALID, "%s: workspace too small", fn);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)rays->Q, plane = align256(sizeof(uint32_t) * n);
    char* w = static_cast<char*>(workspace);
    uint32_t* keys_in = reinterpret_cast<uint32_t*>(w);
    uint32_t* keys_out = reinterpret_cast<uint32_t*>(w + plane);
    int32_t* ids = reinterpret_cast<int32_t*>(w + 2 * plane);
    void* temp = w + 3 * plane;
    size_t temp_bytes = (size_t)workspace_bytes - 3 * plane;
    TreeDev tr = to_dev(tree);
    hipLaunchKernelGGL(ray_key_kernel, dim3((unsigned)((n + kOrderBlock - 1) / kOrderBlock)), dim3(kOrderBlock), 0, st,
                       tr, to_dev(rays), to_dev(opt), keys_in, ids);
    if ((rc = check_launch(fn))) return rc;
    const hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, ids, perm, n, 0, 32, st);
    if (e != hipSuccess) return set_error(SVOXT_ERR_HIP, "%s: radix sort: %s", fn, hipGetErrorString(e));
    return SVOXT_OK;
}

}  // extern "C"
