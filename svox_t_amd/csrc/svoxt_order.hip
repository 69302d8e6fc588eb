// svoxt_order.hip -- ray ordering for batches that are not images.
//
// The marching kernels put 64 consecutive rays of a batch on one wavefront.  For an image
// walked in 8x8 tiles those rays cross the same leaves; for a batch in arbitrary order
// (training on rays drawn at random from many cameras) every lane goes its own way: each
// step's loads hit 64 different cache lines and the wavefront lasts as long as its longest
// ray.  svoxt_ray_order gives the permutation that sorts a batch by the Morton code of the
// point where each ray enters the tree's cube (the key: see SVOXT_ORDER_FACE_BITS below; rays that miss the cube
// last).  The caller gathers origins / dirs / viewdirs with it, renders the sorted batch
// (any entry point of include/svoxt.h) and scatters the output rows back -- results are per
// ray and do not depend on the order; 800x800 shuffled rays of one camera, depth-8 SH9 tree:
// forward 0.82 -> 0.31 ms, forward+backward 1.65 -> 1.2 ms (0.85 with the two-kernel
// backward); rays drawn from 8 cameras: 1.70 -> 1.35 ms.
//
// No counterpart in the reference (which marches rays in the order given).
// The sort is a counting sort on the cell key: count (a returning atomic per ray gives its rank in
// the cell), exclusive scan of the 655 361 counters (two small kernels of our own, below: r05 -- rocPRIM's device scan
// was the vendor primitive until then and brought 121 kernel instances into the library for this one call), scatter --
// 5 launches and ~0.04 ms for 640 000 rays where a radix sort of (key, id) pairs takes 21 launches and 0.14 ms.

#include <cstring>
#include <hip/hip_runtime.h>

#include "svoxt_host.h"

namespace svoxt {

constexpr int kOrderBlock = 256;
// (r04) Keys: a ray that starts outside the cube enters through a face -- (face, u, v) in cells of 1/256 (2-D Morton
// within the face: 393 216 counters) --, one that starts inside is keyed by its origin's 3-D cell of 1/64 (262 144 more).
// Until r04 every ray had a 3-D cell of 1/128 (2 M counters, most of them never touched: entry points lie on three faces):
// the reference-API route of the headline 1 043-1 052 -> 1 080-1 093 Mrays/s (finer groups of 64 rays AND an eighth of
// the counters to clear and scan); 3-D cells of 1/64 or 1/32 alone: 1 003 / 944; faces of 1/512 or 1/1024: 1 055-1 080 /
// 1 009-1 020 (the scan again); faces of 1/128: 1 031-1 061.
constexpr int kAxisBits = 6;     // cells of 2^-kAxisBits of the cube per axis
constexpr int kFaceBits = 8;     // rays that enter through a face are keyed by (face, u, v), 2^-kFaceBits cells
constexpr int kOrderBits = 3 * kAxisBits;   // Morton bits of the 3-D key
constexpr uint32_t kFaceCells = kFaceBits > 0 ? 6u << (2 * kFaceBits) : 0u;      // the face keys come first
constexpr uint32_t kMissKey = kFaceCells + (1u << kOrderBits);

__device__ __forceinline__ uint32_t spread2d(uint32_t x) {      // 16 bits -> every second bit
    x &= 0xffffu;
    x = (x | (x << 8)) & 0x00ff00ffu;
    x = (x | (x << 4)) & 0x0f0f0f0fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
__device__ __forceinline__ uint32_t spread10(uint32_t x) {      // 10 bits -> every third bit
    x &= 0x3ffu;
    x = (x | (x << 16)) & 0x30000ffu;
    x = (x | (x << 8)) & 0x300f00fu;
    x = (x | (x << 4)) & 0x30c30c3u;
    x = (x | (x << 2)) & 0x9249249u;
    return x;
}

// key of ray q and its rank among the rays of the same cell.  Neighbouring rays share cells (and all
// rays that miss the cube share the last one), so a wavefront first groups its lanes by key and sends ONE
// returning atomic per distinct key (64 colliding atomics per wavefront: 0.35 ms for 640 000 rays).  Which wavefront reaches a cell first is not fixed from run to run -- results are per ray and do
// not depend on the order within a cell.
__global__ void __launch_bounds__(kOrderBlock)
ray_key_kernel(TreeDev tr, RaysDev rays, Opts opt, uint32_t* __restrict__ keys, uint32_t* __restrict__ ranks,
               uint32_t* __restrict__ counts) {
    const int64_t q = (int64_t)blockIdx.x * kOrderBlock + threadIdx.x;
    const bool in = q < rays.Q;
    Ray r;
    uint32_t key = kMissKey;                         // misses go last (their wavefronts end at once)
    if (in && setup_ray(tr, rays, opt, q, r)) {
        const float t = r.tmin;                      // >= 0: the origin itself when it lies inside
        const float px = fminf(fmaxf(r.ox + t * r.dx, 0.f), kClampHi);
        const float py = fminf(fmaxf(r.oy + t * r.dy, 0.f), kClampHi);
        const float pz = fminf(fmaxf(r.oz + t * r.dz, 0.f), kClampHi);
        // 7 bits per axis: cells of 1/128 of the cube -- a 64-ray group of the sorted batch spans a handful of
        // neighbouring cells; finer keys would order rays WITHIN what a wavefront holds anyway
        constexpr float kCells = (float)(1 << kAxisBits);
        key = kFaceCells + ((spread10((uint32_t)(px * kCells)) << 2) | (spread10((uint32_t)(py * kCells)) << 1) |
                            spread10((uint32_t)(pz * kCells)));
        if constexpr (kFaceBits > 0) {
            // a ray that starts outside enters through a face: the entry point has two free coordinates, so 2^-kFaceBits
            // cells of a face cost 6 x 4^kFaceBits counters where cells as fine in three dimensions cost 8^kFaceBits
            if (t > 0.f) {
                const float ex = fminf(px, 1.f - px), ey = fminf(py, 1.f - py), ez = fminf(pz, 1.f - pz);
                const int a = (ex <= ey && ex <= ez) ? 0 : (ey <= ez ? 1 : 2);          // the axis the entry face is normal to
                const float pa = a == 0 ? px : a == 1 ? py : pz;
                const float u = a == 0 ? py : px, v = a == 2 ? py : pz;
                constexpr float kF = (float)(1 << kFaceBits);
                const uint32_t face = (uint32_t)(2 * a + (pa > 0.5f ? 1 : 0));
                key = (face << (2 * kFaceBits)) | (spread2d((uint32_t)(u * kF)) << 1) | spread2d((uint32_t)(v * kF));
            }
        }
    }
    const int lane = threadIdx.x & 63;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    // group the lanes by key first (no memory operation in the loop), then all group leaders send their
    // atomics in ONE instruction -- one round trip per wavefront instead of one per distinct key
    int my_leader = lane;
    uint32_t my_rank = 0, my_count = 0;
    unsigned long long todo = __ballot(in);
    while (todo != 0ull) {                           // (wavefront-uniform: one turn per distinct key)
        const int leader = (int)__builtin_ctzll(todo);
        const uint32_t k = (uint32_t)__shfl((int)key, leader, 64);
        const bool mine = in && key == k;
        const unsigned long long m = __ballot(mine);
        if (mine) {
            my_leader = leader;
            my_rank = (uint32_t)__popcll(m & lane_lt);
            my_count = (uint32_t)__popcll(m);
        }
        todo &= ~m;
    }
    uint32_t base = 0;
    if (in && lane == my_leader) base = atomicAdd(counts + key, my_count);
    base = (uint32_t)__shfl((int)base, my_leader, 64);
    const uint32_t rank = base + my_rank;
    if (!in) return;
    keys[q] = key;
    ranks[q] = rank;
}

__global__ void __launch_bounds__(kOrderBlock)
ray_place_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ ranks,
                 const uint32_t* __restrict__ starts, int64_t Q, int32_t* __restrict__ perm) {
    const int64_t q = (int64_t)blockIdx.x * kOrderBlock + threadIdx.x;
    if (q < Q) perm[starts[keys[q]] + ranks[q]] = (int32_t)q;
}

// dst[i, :] = src[perm[i], :] (GATHER) or dst[perm[i], :] = src[i, :]: rows of `cols` floats, one thread per float
__global__ void __launch_bounds__(kOrderBlock)
permute_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ perm, float* __restrict__ dst,
                    int64_t n, int cols, bool gather) {
    const int64_t i = (int64_t)blockIdx.x * kOrderBlock + threadIdx.x;
    if (i >= n * cols) return;
    const int64_t row = i / cols;
    const int c = (int)(i - row * cols);
    const int64_t other = perm[row];
    if (gather) dst[i] = src[other * cols + c];          // (uniform)
    else dst[other * cols + c] = src[i];
}

// the three ray arrays in one launch: thread i < 3n takes float i of the sorted origins, dirs and vdirs
__global__ void __launch_bounds__(kOrderBlock)
gather_rays_kernel(RaysDev rays, const int32_t* __restrict__ perm, float* __restrict__ o, float* __restrict__ d,
                   float* __restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * kOrderBlock + threadIdx.x;
    if (i >= rays.Q * 3) return;
    const int64_t row = i / 3;
    const int64_t from = (int64_t)perm[row] * 3 + (i - row * 3);
    o[i] = rays.origins[from];
    d[i] = rays.dirs[from];
    v[i] = rays.vdirs[from];
}

constexpr size_t kOrderCells = (size_t)kMissKey + 1;            // one counter per cell, the last one for rays that miss the cube

// Exclusive scan of the cell counters, two launches, no cross-workgroup synchronisation: a workgroup owns kScanChunk
// consecutive counters (256 threads x 16); the first kernel leaves each chunk's total, the second adds up the totals of
// the chunks before its own (at most 161 of them for kOrderCells: every thread a few, one LDS reduction) and scans its
// chunk from there -- thread t its 16 counters in registers, the threads' sums by a wavefront scan and one LDS step.
constexpr int kScanThreads = 256, kScanItems = 16, kScanChunk = kScanThreads * kScanItems;
constexpr size_t kScanChunks = (kOrderCells + kScanChunk - 1) / kScanChunk;

__device__ __forceinline__ uint32_t block_sum_256(uint32_t v, uint32_t* lds /* [4] */) {
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    const uint32_t s = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return s;
}

__global__ void __launch_bounds__(kScanThreads)
scan_chunk_sums_kernel(const uint32_t* __restrict__ counts, size_t n, uint32_t* __restrict__ chunk_sums) {
    __shared__ uint32_t lds[4];
    const size_t base = (size_t)blockIdx.x * kScanChunk + (size_t)threadIdx.x * kScanItems;
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) if (base + j < n) v += counts[base + j];
    const uint32_t s = block_sum_256(v, lds);
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = s;
}

__global__ void __launch_bounds__(kScanThreads)
scan_chunks_kernel(const uint32_t* __restrict__ counts, size_t n, const uint32_t* __restrict__ chunk_sums,
                   uint32_t* __restrict__ starts) {
    __shared__ uint32_t lds[4];
    uint32_t before = 0;
    for (unsigned c = threadIdx.x; c < blockIdx.x; c += kScanThreads) before += chunk_sums[c];
    const uint32_t offset = block_sum_256(before, lds);
    const size_t base = (size_t)blockIdx.x * kScanChunk + (size_t)threadIdx.x * kScanItems;
    uint32_t v[kScanItems], mine = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) { v[j] = base + j < n ? counts[base + j] : 0u; mine += v[j]; }
    // exclusive scan of the threads' sums: within the wavefront, then over the four wavefronts
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += u;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    uint32_t run = offset + incl - mine;
    for (int w = 0; w < wave; ++w) run += lds[w];
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        if (base + j < n) starts[base + j] = run;
        run += v[j];
    }
}

// Is this batch a row-major pinhole image nobody declared?  The first 65 536 rays are looked at: the L1 distance between
// consecutive directions is small along an image row and jumps where the next row starts.  res[0]: every origin equals
// the first; res[1]: jumps found (at most 2); res[2], res[3]: their positions; res[4]: the caller's ticket.  The
// threshold is four times the (lower) median of the first 1 024 distances, which every workgroup works out for itself.
// Two launches: 64 workgroups, a ray per thread, leave their two first jumps in res[8 + 3 b ..]; one wavefront merges.
constexpr int kProbeThreads = 1024, kProbeBlocks = 64, kProbeNone = 0x7fffffff;
__device__ __forceinline__ float probe_step(const float* __restrict__ d, int i) {
    return (fabsf(d[3 * (i + 1)] - d[3 * i]) + fabsf(d[3 * (i + 1) + 1] - d[3 * i + 1])) + fabsf(d[3 * (i + 1) + 2] - d[3 * i + 2]);
}
__global__ void __launch_bounds__(kProbeThreads)
image_probe_kernel(const float* __restrict__ o, const float* __restrict__ d, int64_t Q, int32_t* __restrict__ res) {
    __shared__ float s[kProbeThreads];
    __shared__ float s_med;
    __shared__ int s_same, s_first, s_second;
    const int t = (int)threadIdx.x;
    const int n = (int)(Q < (int64_t)kProbeThreads * kProbeBlocks ? Q : (int64_t)kProbeThreads * kProbeBlocks);
    if (t == 0) { s_med = __builtin_inff(); s_same = 1; s_first = kProbeNone; s_second = kProbeNone; }
    s[t] = probe_step(d, t);                                 // (the host refuses Q < 4 096)
    __syncthreads();
    const float mine = s[t];
    int lt = 0, le = 0;
    for (int j = 0; j < kProbeThreads; ++j) { const float x = s[j]; lt += x < mine ? 1 : 0; le += x <= mine ? 1 : 0; }
    if (lt <= kProbeThreads / 2 - 1 && kProbeThreads / 2 - 1 < le) s_med = mine;       // (every such thread holds the same value)
    __syncthreads();
    const float thr = s_med * 4.f;
    const int i = (int)blockIdx.x * kProbeThreads + t;
    bool jump = false;
    if (i < n) {
        if (!(o[3 * i] == o[0] && o[3 * i + 1] == o[1] && o[3 * i + 2] == o[2])) s_same = 0;
        jump = i < n - 1 && probe_step(d, i) > thr;
        if (jump) atomicMin(&s_first, i);
    }
    __syncthreads();
    if (jump && i > s_first) atomicMin(&s_second, i);
    __syncthreads();
    if (t == 0) {
        res[8 + 3 * blockIdx.x] = s_same;
        res[9 + 3 * blockIdx.x] = s_first;
        res[10 + 3 * blockIdx.x] = s_second;
    }
}
__global__ void __launch_bounds__(64)
image_probe_merge_kernel(int32_t* __restrict__ res, int32_t ticket) {
    const int b = (int)threadIdx.x;                          // kProbeBlocks == 64: a workgroup's partial per lane
    const int same = res[8 + 3 * b], f = res[9 + 3 * b], s2 = res[10 + 3 * b];
    int first = f;
    for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, 64));
    int second = f > first ? f : s2;                         // this workgroup's smallest jump past the batch's first
    for (int off = 32; off > 0; off >>= 1) second = min(second, __shfl_xor(second, off, 64));
    const unsigned long long all_same = __ballot(same != 0);
    if (b == 0) {
        res[0] = all_same == ~0ull ? 1 : 0;
        res[1] = (first != kProbeNone ? 1 : 0) + (second != kProbeNone ? 1 : 0);
        res[2] = first != kProbeNone ? first : 0;
        res[3] = second != kProbeNone ? second : 0;
        res[4] = ticket;
    }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace svoxt

using namespace svoxt;

extern "C" {

int64_t svoxt_ray_order_workspace_bytes(int64_t Q) {
    if (Q < 0 || Q > 0x7fffffff) return -1;
    // keys, ranks (per ray); counts, starts (per cell); the scan's chunk totals
    return (int64_t)(2 * align256(sizeof(uint32_t) * (size_t)Q) + 2 * align256(sizeof(uint32_t) * kOrderCells) +
                     align256(sizeof(uint32_t) * kScanChunks));
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
    const size_t n = (size_t)rays->Q, plane = align256(sizeof(uint32_t) * n), cells = align256(sizeof(uint32_t) * kOrderCells);
    char* w = static_cast<char*>(workspace);
    uint32_t* keys = reinterpret_cast<uint32_t*>(w);
    uint32_t* ranks = reinterpret_cast<uint32_t*>(w + plane);
    uint32_t* counts = reinterpret_cast<uint32_t*>(w + 2 * plane);
    uint32_t* starts = reinterpret_cast<uint32_t*>(w + 2 * plane + cells);
    uint32_t* chunk_sums = reinterpret_cast<uint32_t*>(w + 2 * plane + 2 * cells);
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(uint32_t) * kOrderCells, st);
    if (e != hipSuccess) return set_error(SVOXT_ERR_HIP, "%s: hipMemsetAsync: %s", fn, hipGetErrorString(e));
    TreeDev tr = to_dev(tree);
    const unsigned nb = (unsigned)((n + kOrderBlock - 1) / kOrderBlock);
    hipLaunchKernelGGL(ray_key_kernel, dim3(nb), dim3(kOrderBlock), 0, st, tr, to_dev(rays, tree), to_dev(opt), keys, ranks, counts);
    if ((rc = check_launch(fn))) return rc;
    hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3((unsigned)kScanChunks), dim3(kScanThreads), 0, st, counts, kOrderCells, chunk_sums);
    hipLaunchKernelGGL(scan_chunks_kernel, dim3((unsigned)kScanChunks), dim3(kScanThreads), 0, st, counts, kOrderCells, chunk_sums, starts);
    if ((rc = check_launch(fn))) return rc;
    hipLaunchKernelGGL(ray_place_kernel, dim3(nb), dim3(kOrderBlock), 0, st, keys, ranks, starts, (int64_t)n, perm);
    return check_launch(fn);
}

int svoxt_image_probe(const svoxt_rays* rays, int32_t* result, int32_t ticket, void* stream) {
    const char* fn = "svoxt_image_probe";
    static_assert(8 + 3 * kProbeBlocks <= SVOXT_IMAGE_PROBE_WORDS, "the partials fit the result buffer");
    int rc;
    if ((rc = check_rays(rays, fn))) return rc;
    if (result == nullptr) return set_error(SVOXT_ERR_INVALID, "%s: result is NULL", fn);
    if (rays->c2w != nullptr || rays->origins == nullptr || rays->dirs == nullptr)
        return set_error(SVOXT_ERR_INVALID, "%s: a batch of ray arrays is needed", fn);
    if (rays->Q < 4096) return set_error(SVOXT_ERR_INVALID, "%s: fewer than 4096 rays", fn);
    hipLaunchKernelGGL(image_probe_kernel, dim3(kProbeBlocks), dim3(kProbeThreads), 0, (hipStream_t)stream, rays->origins, rays->dirs,
                       rays->Q, result);
    hipLaunchKernelGGL(image_probe_merge_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, result, ticket);
    return check_launch(fn);
}

int svoxt_gather_rays(const svoxt_rays* rays, const int32_t* perm, float* origins, float* dirs, float* vdirs, void* stream) {
    const char* fn = "svoxt_gather_rays";
    int rc;
    if ((rc = check_rays(rays, fn))) return rc;
    if (rays->Q == 0) return SVOXT_OK;
    if (rays->c2w != nullptr) return set_error(SVOXT_ERR_INVALID, "%s: camera mode has no ray arrays", fn);
    if (perm == nullptr || origins == nullptr || dirs == nullptr || vdirs == nullptr)
        return set_error(SVOXT_ERR_INVALID, "%s: NULL argument", fn);
    const int64_t n = rays->Q * 3;
    hipLaunchKernelGGL(gather_rays_kernel, dim3((unsigned)((n + kOrderBlock - 1) / kOrderBlock)), dim3(kOrderBlock), 0,
                       (hipStream_t)stream, to_dev(rays, nullptr), perm, origins, dirs, vdirs);
    return check_launch(fn);
}

int svoxt_permute_rows(const float* src, const int32_t* perm, float* dst, int64_t n, int32_t cols, int32_t scatter, void* stream) {
    const char* fn = "svoxt_permute_rows";
    if (n < 0 || cols < 1) return set_error(SVOXT_ERR_INVALID, "%s: bad extents", fn);
    if (n == 0) return SVOXT_OK;
    if (src == nullptr || perm == nullptr || dst == nullptr) return set_error(SVOXT_ERR_INVALID, "%s: NULL argument", fn);
    const int64_t tot = n * cols;
    const unsigned nb = (unsigned)((tot + kOrderBlock - 1) / kOrderBlock);
    hipLaunchKernelGGL(permute_rows_kernel, dim3(nb), dim3(kOrderBlock), 0, (hipStream_t)stream, src, perm, dst, n, (int)cols, scatter == 0);
    return check_launch(fn);
}

}  // extern "C"
