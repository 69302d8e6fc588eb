// svoxt_misc_kernels.h -- the kernels around the render path: opacity forward / backward from lists,
// depth, the roofline counters, point query and its unique-leaf compaction, row compaction, and the
// acceleration-grid build.  See the file header of svoxt_kernels.hip.
#pragma once
#include <type_traits>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "svoxt_device.h"
#include "svoxt_lists.h"

#pragma clang fp contract(off)

namespace svoxt {

// ---------------------------------------------------------------------------
// Opacity (rt_kernel.cu:500-560, :1110-1126) and depth (:782-834, :866-882)
// ---------------------------------------------------------------------------

// REC (thresholds 0): also records each ray's samples with sigma > 0 as (feature row,
// delta_t) in rec[k][q] and aux[q] = {count | overflow << 31, t of the first unrecorded
// sample, final transmittance, -} for svoxt_opacity_render_bwd_replay.
template <bool N2, bool REC = false>
__global__ void __launch_bounds__(kBlock)
opacity_fwd_kernel(TreeDev tr, RaysDev rays, Opts opt, float* __restrict__ out,
                   RecLists L = RecLists{}, uint4* __restrict__ aux = nullptr) {
    __shared__ uint2 rstage[REC ? kRecBlock * kBlock : 1];
    __shared__ int32_t ltab[REC ? kMaxRecBlocks : 1];
    if constexpr (REC) rec_tab_init(ltab);
    const int S = L.S;
    int64_t cur_block = 0;
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t q = ray_of_thread(rays, tid);
    if (q >= rays.Q) return;
    Ray r;
    if (!setup_ray(tr, rays, opt, q, r)) {
        out[q] = 0.f;
        if constexpr (REC) aux[q] = make_uint4(0u, 0u, __float_as_uint(1.f), 0u);
        return;
    }
    const int K = tr.K;
    float light = 1.f, t = r.tmin;
    int nrec = 0;
    bool over = false, stopped = false;
    float t_resume = 0.f;
    while (t < r.tmax) {
        Sample s;
        march_step<N2>(tr, r, opt.step_size, t, s);
        if (s.valid) {
            const float sigma = tr.features[(int64_t)s.idx * K + (K - 1)];
            // (REC, r05: every sigma > 0 is recorded -- the backward's set, rt_kernel.cu:382 -- and composited only by the
            // forward's own rules: sigma > sigma_thresh, not after T <= stop_thresh)
            if (sigma > (REC ? 0.f : opt.sigma_thresh)) {
                if constexpr (REC) {
                    bool room = nrec < S;
                    if (room && (nrec & 7) == 0) {
                        cur_block = rec_block_begin(L, ltab, tid >> 6, nrec >> 3);
                        room = cur_block >= 0;
                    }
                    if (room) {
                        rec_stage_put(rstage, (int)threadIdx.x, L.rec, cur_block, nrec, (uint32_t)s.idx, s.delta_t);
                        ++nrec;
                    } else if (!over) {
                        over = true;
                        t_resume = t;
                    }
                }
                if (!REC || (sigma > opt.sigma_thresh && !stopped)) {
                    light *= pexpf(-s.delta_t * r.delta_scale * sigma);
                    if (light <= opt.stop_thresh) {
                        if constexpr (!REC) break;
                        stopped = true;
                    }
                }
            }
        }
        t = march_advance(t, s.delta_t);
    }
    out[q] = 1.f - light;
    if constexpr (REC) {
        rec_stage_finish(rstage, (int)threadIdx.x, L.rec, cur_block, nrec);
        aux[q] = make_uint4((uint32_t)nrec | (over ? kRecOverflow : 0u), __float_as_uint(t_resume),
                            __float_as_uint(light), 0u);
        if (over) note_overflow(L);
    }
}

// opacity_render_backward from recorded lists (C = 0 of trace_ray_backward,
// rt_kernel.cu:331-496, 1593-1616): the only gradient is the sigma entry
//     delta_t * delta_scale * grad_output * T_ray            (:486-490 with no colour terms)
// with T_ray the final transmittance as the reference's backward computes it
// (exponent associated as in :397).  One walk: rec[k][q] <- (row, (delta_t * delta_scale) *
// grad_output) and aux[q].w <- T_ray; opacity_merge_kernel multiplies by T_ray and adds
// up per tile.  Rays whose list overflowed march their tail here (twice: T_ray must be
// complete before their tail samples can be sent) with per-lane atomics.
template <bool N2>
__global__ void __launch_bounds__(kBlock)
opacity_walk_kernel(TreeDev tr, RaysDev rays, Opts opt, const float* __restrict__ grad_out,
                    float* __restrict__ grad, int gstride, RecLists L, uint4* __restrict__ aux) {
    uint2* __restrict__ rec = L.rec;
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t q = ray_of_thread(rays, tid);
    if (q >= rays.Q) return;
    const uint4 a = aux[q];
    const int nrec = (int)(a.x & ~kRecOverflow);
    const bool over = (a.x & kRecOverflow) != 0u;
    if (nrec == 0 && !over) return;
    Ray r;
    if (!setup_ray(tr, rays, opt, q, r)) return;
    const int K = tr.K;
    const float g = grad_out[q];
    float light = 1.f;
    // (r04) A block of 8 records is the ray's own 64-byte line: fetched whole, its 8 sigma gathers requested back to
    // back, the 8 exponentials formed -- all independent -- and only the products taken in list order; one record,
    // one gather, one exponential at a time the walk was a chain of two dependent loads per sample (0.27 ms at
    // 800 x 800 / depth 8: more than the recording forward and the merge together).
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    for (int kb = 0; kb < nrec; kb += kRecBlock) {
        v4u* line = reinterpret_cast<v4u*>(rec + rec_index(L, tid, kb));
        const int n_here = nrec - kb;
        v4u w[kRecBlock / 2];
#pragma unroll
        for (int j = 0; j < kRecBlock / 2; ++j) w[j] = line[j];
        float sig[kRecBlock];
#pragma unroll
        for (int j = 0; j < kRecBlock; ++j) {
            const uint32_t row = j < n_here ? ((j & 1) ? w[j >> 1].z : w[j >> 1].x) : 0u;      // (slots past the count hold stale bits)
            sig[j] = tr.features[(int64_t)(int32_t)row * K + (K - 1)];
        }
        float att[kRecBlock];
#pragma unroll
        for (int j = 0; j < kRecBlock; ++j) {
            const float delta_t = __uint_as_float((j & 1) ? w[j >> 1].w : w[j >> 1].y);
            att[j] = pexpf(-delta_t * sig[j] * r.delta_scale);
            const float v = delta_t * r.delta_scale * g;
            if (j < n_here) { if (j & 1) w[j >> 1].w = __float_as_uint(v); else w[j >> 1].y = __float_as_uint(v); }
        }
#pragma unroll
        for (int j = 0; j < kRecBlock; ++j)
            if (j < n_here) light *= att[j];
#pragma unroll
        for (int j = 0; j < kRecBlock / 2; ++j) line[j] = w[j];
    }
    if (over) {
        const float t0 = __uint_as_float(a.y);
        for (int pass = 0; pass < 2; ++pass) {
            const float light_ray = light;                  // complete only in the second pass
            float t = t0;
            while (t < r.tmax) {
                Sample s;
                march_step<N2>(tr, r, opt.step_size, t, s);
                if (s.valid) {
                    const float sigma = tr.features[(int64_t)s.idx * K + (K - 1)];
                    if (sigma > 0.f) {
                        if (pass == 0) light *= pexpf(-s.delta_t * sigma * r.delta_scale);
                        else atomicAdd(grad + (int64_t)s.idx * gstride + (K - 1), s.delta_t * r.delta_scale * g * light_ray);
                    }
                }
                t = march_advance(t, s.delta_t);
            }
        }
    }
    aux[q].w = __float_as_uint(light);
}

// Per-tile sum of the walk's records: W wavefronts share a hash table of T feature rows
// (atomicCAS on the key, ds_add_f32 on the one value per row: 1 LDS float atomic per
// record is cheap, 28 were not), flushed after every pass of at most T records.
template <int T, int W>
__global__ void __launch_bounds__(64 * W)
opacity_merge_kernel(RaysDev rays, RecLists L, const uint4* __restrict__ aux,
                     float* __restrict__ grad, int gstride, int col) {
    constexpr int NT = 64 * W;
    constexpr int kGroup = 2, kRound = kGroup * W, RPP = T / (64 * kRound);
    static_assert((T & (T - 1)) == 0 && RPP >= 1, "a pass of RPP rounds must fit the table");
    if (tile_never_recorded(L, blockIdx.x)) return;          // (r03: see grad_fused_kernel)
    __shared__ int32_t keys[T];
    __shared__ float vals[T];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t tabreg = rec_tab_reg(L, blockIdx.x, lane);
    const int64_t q = ray_of_thread(rays, (int64_t)blockIdx.x * 64 + lane);
    int nrec = 0;
    float t_ray = 0.f;
    if (q < rays.Q) {
        const uint4 a = aux[q];
        nrec = (int)(a.x & ~kRecOverflow);
        t_ray = __uint_as_float(a.w);
    }
    int maxn = nrec;
    for (int off = 32; off > 0; off >>= 1) maxn = max(maxn, __shfl_xor(maxn, off, 64));
    maxn = __builtin_amdgcn_readfirstlane(maxn);
    if (maxn == 0) return;
    for (int i = threadIdx.x; i < T; i += NT) { keys[i] = -1; vals[i] = 0.f; }
    __syncthreads();
    for (int k0 = 0; k0 < maxn; k0 += RPP * kRound) {
#pragma unroll 1
        for (int rd = 0; rd < RPP; ++rd) {
            const int kb = k0 + rd * kRound + wave * kGroup;
            if (kb >= maxn) break;
            uint2 e[kGroup];
#pragma unroll
            for (int u = 0; u < kGroup; ++u) {
                e[u] = make_uint2(0u, 0u);
                if (kb + u < nrec) e[u] = rec_get(L.rec + rec_index_in(rec_block_u(L, tabreg, blockIdx.x, (kb + u) >> 3), lane, kb + u));
            }
#pragma unroll
            for (int u = 0; u < kGroup; ++u) {
                if (kb + u < nrec) {
                    const int32_t idx = (int32_t)e[u].x;
                    uint32_t h = ((uint32_t)idx * 0x9E3779B1u) >> (32 - __builtin_ctz(T));
                    while (true) {
                        const int32_t old = atomicCAS(keys + h, -1, idx);
                        if (old == -1 || old == idx) break;
                        h = (h + 1u) & (uint32_t)(T - 1);
                    }
                    atomicAdd(vals + h, __uint_as_float(e[u].y) * t_ray);     // ((delta_t * ds) * g) * T_ray
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < T; i += NT) {
            const int32_t key = keys[i];
            if (key >= 0) {
                atomicAdd(grad + (int64_t)key * gstride + col, vals[i]);
                keys[i] = -1;
                vals[i] = 0.f;
            }
        }
        __syncthreads();
    }
}

template <bool N2>
__global__ void __launch_bounds__(kBlock)
depth_kernel(TreeDev tr, RaysDev rays, Opts opt, float* __restrict__ depth) {
    const int64_t q = ray_of_thread(rays, (int64_t)blockIdx.x * kBlock + threadIdx.x);
    if (q >= rays.Q) return;
    Ray r;
    float d = 0.f;
    if (setup_ray(tr, rays, opt, q, r)) {
        const int K = tr.K;
        float t = r.tmin;
        while (t < r.tmax) {
            Sample s;
            march_step<N2>(tr, r, opt.step_size, t, s);
            if (s.valid) {
                const float sigma = tr.features[(int64_t)s.idx * K + (K - 1)];
                if (sigma > opt.sigma_thresh) { d = r.delta_scale * t; break; }
            }
            t = march_advance(t, s.delta_t);
        }
    }
    depth[q] = d;
}

// ---------------------------------------------------------------------------
// Roofline counters
// ---------------------------------------------------------------------------


template <bool N2>
__global__ void __launch_bounds__(kBlock)
count_fwd_kernel(TreeDev tr, RaysDev rays, Opts opt, unsigned long long* __restrict__ counters) {
    const int64_t q = ray_of_thread(rays, (int64_t)blockIdx.x * kBlock + threadIdx.x);
    unsigned long long hit = 0, steps = 0, levels = 0, valid = 0, active = 0;
    Ray r;
    if (q < rays.Q && setup_ray(tr, rays, opt, q, r)) {
        hit = 1;
        const int K = tr.K;
        float light = 1.f, t = r.tmin;
        while (t < r.tmax) {
            Sample s;
            march_step<N2>(tr, r, opt.step_size, t, s);
            ++steps;
            levels += s.leaf.levels;
            if (s.valid) {
                ++valid;
                const float sigma = tr.features[(int64_t)s.idx * K + (K - 1)];
                if (sigma > opt.sigma_thresh) {
                    ++active;
                    light *= pexpf(-s.delta_t * r.delta_scale * sigma);
                    if (light <= opt.stop_thresh) break;
                }
            }
            t = march_advance(t, s.delta_t);
        }
    }
    hit = wave_sum(hit); steps = wave_sum(steps); levels = wave_sum(levels);
    valid = wave_sum(valid); active = wave_sum(active);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(counters + 0, hit);
        atomicAdd(counters + 1, steps);
        atomicAdd(counters + 2, levels);
        atomicAdd(counters + 3, valid);
        atomicAdd(counters + 4, active);
    }
}

// What one forward march touches (svoxt_count_touched; for the roofline's compulsory bytes):
// row_mask[idx] = 1 for every valid leaf's feature row (the forward reads it), row_mask[M + idx] = 1
// if a sample there is composited (the backward reads the row again); tree_mask: grid cells and (child, data)
// pairs with the acceleration grid, child and data words without (march_step<..., MARK>);
// longest[0] = the most leaf crossings any ray makes.  The march is the real one.
template <bool N2>
__global__ void __launch_bounds__(kBlock)
count_touched_kernel(TreeDev tr, RaysDev rays, Opts opt, uint8_t* __restrict__ row_mask,
                     uint8_t* __restrict__ tree_mask, uint32_t n_slots, unsigned long long* __restrict__ longest) {
    const int64_t q = ray_of_thread(rays, (int64_t)blockIdx.x * kBlock + threadIdx.x);
    unsigned long long steps = 0;
    Ray r;
    if (q < rays.Q && setup_ray(tr, rays, opt, q, r)) {
        const int K = tr.K;
        float light = 1.f, t = r.tmin;
        while (t < r.tmax) {
            Sample s;
            march_step<N2, -1, true>(tr, r, opt.step_size, t, s, tree_mask, n_slots);
            ++steps;
            if (s.valid) {
                const float sigma = tr.features[(int64_t)s.idx * K + (K - 1)];
                row_mask[s.idx] = 1;                      // (every writer stores the same value: no race to lose)
                if (sigma > opt.sigma_thresh) {
                    row_mask[tr.M + s.idx] = 1;
                    light *= pexpf(-s.delta_t * r.delta_scale * sigma);
                    if (light <= opt.stop_thresh) break;
                }
            }
            t = march_advance(t, s.delta_t);
        }
    }
    for (int off = 32; off > 0; off >>= 1) steps = max(steps, (unsigned long long)__shfl_down(steps, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(longest, steps);
}

// ---------------------------------------------------------------------------
// Point query (svox_kernel.cu:45-94)
// ---------------------------------------------------------------------------

// (r04) The rows of a wavefront's 64 points, moved by the wavefront together: two points per instruction for rows of up
// to 32 floats (lanes 0-31 the columns of one, 32-63 of the next), one for wider rows -- a point's row is then ONE
// contiguous segment per load / store / atomic instruction, where a lane that walks its own row touches 64 different
// rows with every instruction (the backward: 28 memory-side atomic requests per point instead of 2-3; 1 M points on the
// depth-8 shell: query + backward 2.17 ms before: profiles/r04_query_timing.txt).  ridx: the lane's feature row, < 0: none.
template <bool BWD>
__device__ __forceinline__ void query_rows(const TreeDev& tr, int32_t ridx, int64_t q0, int64_t Q,
                                           float* __restrict__ values, const float* __restrict__ grad_out,
                                           float* __restrict__ grad) {
    const int lane = threadIdx.x & 63;
    const int K = tr.K;
    const int per = K <= 32 ? 2 : 1;                 // points per instruction
    const int lpr = 64 / per;                        // lanes per point
    const int sub = per == 2 ? (lane >> 5) : 0, j0 = per == 2 ? (lane & 31) : lane;
    for (int base = 0; base < 64; base += per) {     // (uniform: every lane takes part in the cross-lane read)
        const int p = base + sub;
        const int32_t r = __shfl(ridx, p, 64);
        const int64_t qq = q0 + p;
        if (qq >= Q) continue;
        for (int col = j0; col < K; col += lpr) {
            if constexpr (BWD) {
                if (r >= 0) atomicAdd(grad + (int64_t)r * K + col, grad_out[qq * K + col]);
            } else {
                values[qq * K + col] = r >= 0 ? tr.features[(int64_t)r * K + col] : 0.f;
            }
        }
    }
}

// the leaf of point q (q < Q): its slot, and its feature row or -1
template <bool N2>
__device__ __forceinline__ int32_t query_locate(const TreeDev& tr, const float* __restrict__ points, int64_t q, uint32_t& slot) {
    const float* p = points + 3 * q;
    const float px = tr.offset[0] + tr.scaling[0] * p[0];
    const float py = tr.offset[1] + tr.scaling[1] * p[1];
    const float pz = tr.offset[2] + tr.scaling[2] * p[2];
    Leaf lf;
    locate<N2>(tr, px, py, pz, lf);
    slot = lf.slot;
    const int32_t idx = tr.data[lf.slot];
    return (idx >= 0 && (int64_t)idx < tr.M) ? idx : -1;
}

template <bool N2>
__global__ void __launch_bounds__(kBlock)
query_fwd_kernel(TreeDev tr, const float* __restrict__ points, int64_t Q,
                 float* __restrict__ values, int64_t* __restrict__ node_ids,
                 int64_t* __restrict__ data_ids, uint8_t* __restrict__ hit_mask) {
    static_assert(kBlock == 64, "query_rows: one wavefront per workgroup");
    const int64_t q0 = (int64_t)blockIdx.x * kBlock, q = q0 + threadIdx.x;
    int32_t idx = -1;
    if (q < Q) {
        uint32_t slot;
        idx = query_locate<N2>(tr, points, q, slot);
        node_ids[q] = (int64_t)slot;
        if (hit_mask != nullptr) hit_mask[slot] = 1;
        data_ids[q] = idx;
    }
    query_rows<false>(tr, idx, q0, Q, values, nullptr, nullptr);
}

template <bool N2>
__global__ void __launch_bounds__(kBlock)
query_bwd_kernel(TreeDev tr, const float* __restrict__ points, int64_t Q,
                 const float* __restrict__ grad_out, float* __restrict__ grad) {
    const int64_t q0 = (int64_t)blockIdx.x * kBlock, q = q0 + threadIdx.x;
    int32_t idx = -1;
    if (q < Q) {
        uint32_t slot;
        idx = query_locate<N2>(tr, points, q, slot);
    }
    query_rows<true>(tr, idx, q0, Q, nullptr, grad_out, grad);
}

// ---------------------------------------------------------------------------
// Unique-leaf list of a point query: compaction of the hit mask into
// leaf_node[U, 4] = (node, u, v, w), sorted by packed leaf id.  The reference
// numbers the hits with a float atomic counter (svox_kernel.cu:260-269: order
// undefined, exact only below 2^24 leaves); here a three-step integer prefix
// sum gives a deterministic order: per-segment counts, scan of the counts,
// ranked scatter.
// ---------------------------------------------------------------------------

constexpr int kSeg = 1024;     // mask entries per workgroup (4 per thread)

__global__ void __launch_bounds__(kBlock)
leaves_count_kernel(const uint8_t* __restrict__ mask, int64_t n, int32_t* __restrict__ seg_count) {
    __shared__ int32_t wsum[kBlock / 64];
    const int64_t base = (int64_t)blockIdx.x * kSeg;
    int c = 0;
    for (int i = threadIdx.x; i < kSeg; i += kBlock) c += (base + i < n && mask[base + i]) ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t tot = 0;
        for (int w = 0; w < kBlock / 64; ++w) tot += wsum[w];
        seg_count[blockIdx.x] = tot;
    }
}

// exclusive scan of seg_count[0..nseg) in place; total -> *count (one workgroup)
__global__ void __launch_bounds__(kBlock)
leaves_scan_kernel(int32_t* __restrict__ seg_count, int nseg, int64_t* __restrict__ count) {
    __shared__ int32_t part[kBlock];
    const int per = (nseg + kBlock - 1) / kBlock;
    const int lo = threadIdx.x * per, hi = min(lo + per, nseg);
    int32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += seg_count[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t run = 0;
        for (int i = 0; i < kBlock; ++i) { const int32_t v = part[i]; part[i] = run; run += v; }
        *count = run;
    }
    __syncthreads();
    int32_t run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { const int32_t v = seg_count[i]; seg_count[i] = run; run += v; }
}

__global__ void __launch_bounds__(kBlock)
leaves_scatter_kernel(const uint8_t* __restrict__ mask, int64_t n, int N, const int32_t* __restrict__ seg_offset,
                      int64_t* __restrict__ leaf_node) {
    __shared__ int32_t wbase[kBlock / 64];
    const int64_t base = (int64_t)blockIdx.x * kSeg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t running = seg_offset[blockIdx.x];
    // 4 rounds of 256 consecutive entries keep the output in increasing slot order
    for (int rd = 0; rd < kSeg / kBlock; ++rd) {
        const int64_t i = base + rd * kBlock + threadIdx.x;
        const bool hit = i < n && mask[i] != 0;
        const unsigned long long b = __ballot(hit);
        if (lane == 0) wbase[wave] = __popcll(b);
        __syncthreads();
        int32_t before = 0, total = 0;
        for (int w = 0; w < kBlock / 64; ++w) { if (w < wave) before += wbase[w]; total += wbase[w]; }
        if (hit) {
            const int64_t dst = running + before + __popcll(b & ((1ull << lane) - 1ull));
            int64_t tmp = i;
            const int64_t w3 = tmp % N; tmp /= N;
            const int64_t v3 = tmp % N; tmp /= N;
            const int64_t u3 = tmp % N; tmp /= N;
            leaf_node[4 * dst + 0] = tmp;
            leaf_node[4 * dst + 1] = u3;
            leaf_node[4 * dst + 2] = v3;
            leaf_node[4 * dst + 3] = w3;
        }
        running += total;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// [M, stride] -> dense [M, K] (the backward accumulates into 64-byte-aligned rows)
// ---------------------------------------------------------------------------

// streaming copy: non-temporal both ways, the data is not re-read by these kernels
// CLEAR (svoxt_compact_rows_clear): what was read is left zeroed -- the padded gradient scratch is then ready for the
// next backward's atomics and that backward needs no fill of its own (the pad columns are never written by anyone)
template <typename V, bool CLEAR = false>
__global__ void __launch_bounds__(kBlock)
compact_rows_kernel(typename std::conditional<CLEAR, V, const V>::type* __restrict__ src, int64_t n, int Kv, int stride_v,
                    V* __restrict__ dst) {
    if constexpr (CLEAR) {
        // n = rows * stride_v here: the threads cover the PADDED rows, so that the zeros go out as whole lines (zeroing
        // 112 of a row's 128 bytes leaves every second line a partial write: 57 us against 27 + 14 for copy + fill, r03)
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
            const int64_t r = i / stride_v;
            const int c = (int)(i - r * stride_v);
            if (c < Kv) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + r * Kv + c);
            __builtin_nontemporal_store(V{}, src + i);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
            const int64_t r = i / Kv;
            const int c = (int)(i - r * Kv);
            __builtin_nontemporal_store(__builtin_nontemporal_load(src + r * stride_v + c), dst + i);
        }
    }
}

// (r04) The clearing form for the shape the per-tile backwards use -- float4 elements, a padded row of SV of them (a power
// of two: 8 = 128 bytes for K = 28) -- with 32-bit element indices (shift and mask for row and column where the general
// kernel divides 64-bit numbers: ~100 instructions per element, 14 us of vector arithmetic at 800 x 800 / depth 8 beside
// 30 us of memory time) and U elements per thread in flight: all loads, then the stores.
template <int SV, int U>
__global__ void __launch_bounds__(256)
compact_rows_clear_pow2_kernel(float4* __restrict__ src, uint32_t n /* rows * SV */, int Kv, float4* __restrict__ dst) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f* s = reinterpret_cast<v4f*>(src);
    v4f* d = reinterpret_cast<v4f*>(dst);
    const uint32_t i0 = (blockIdx.x * (uint32_t)U) * 256u + threadIdx.x;        // (the host keeps n + U * 256 below 2^32)
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t i = i0 + (uint32_t)u * 256u;
        v[u] = v4f{0.f, 0.f, 0.f, 0.f};
        if (i < n && (int)(i & (SV - 1)) < Kv) v[u] = __builtin_nontemporal_load(s + i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t i = i0 + (uint32_t)u * 256u;
        if (i < n) {
            const uint32_t r = i / SV, c = i & (SV - 1);
            if ((int)c < Kv) __builtin_nontemporal_store(v[u], d + (size_t)r * (size_t)Kv + c);
            __builtin_nontemporal_store(v4f{0.f, 0.f, 0.f, 0.f}, s + i);
        }
    }
}

// ---------------------------------------------------------------------------
// Acceleration grid build (N == 2): one thread per cell, see locate_accel()
// ---------------------------------------------------------------------------

__global__ void __launch_bounds__(kBlock)
accel_build_kernel(TreeDev tr, int G, bool bricks, uint32_t* __restrict__ cells) {
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= (1u << (3 * G))) return;
    uint32_t cx, cy, cz;
    accel_cell_coords(c, G, bricks, cx, cy, cz);
    int32_t node = 0;
    for (int k = 1; k <= G; ++k) {
        const int sh = G - k;
        const uint32_t c3 = (((cx >> sh) & 1u) << 2) | (((cy >> sh) & 1u) << 1) | ((cz >> sh) & 1u);
        const uint32_t slot = ((uint32_t)node << 3) + c3;
        const int32_t skip = tr.child[slot];
        if (skip == 0) {
            const uint32_t row = (uint32_t)tr.data[slot];
            cells[c] = kAccelLeaf | ((uint32_t)k << 27) | ((int64_t)row < tr.M ? row : kAccelIdx);
            return;
        }
        node += skip;
    }
    cells[c] = (uint32_t)node;
}

// ... and the (child, data) pairs the descent below the grid reads
__global__ void __launch_bounds__(kBlock)
accel_nodes_kernel(const int32_t* __restrict__ child, const int32_t* __restrict__ data, int64_t n,
                   uint2* __restrict__ nodes) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) nodes[i] = make_uint2((uint32_t)child[i], (uint32_t)data[i]);
}

}  // namespace svoxt
