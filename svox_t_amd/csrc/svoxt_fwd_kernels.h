// svoxt_fwd_kernels.h -- volume_render forward (trace_ray, rt_kernel.cu:222-328): the one-kernel
// forward (optionally recording sample lists), the generic fallback, and the forward as two kernels
// (march, then shade per tile / with channels on lanes, plus their tail launches).  See the file
// header of svoxt_kernels.hip and DESIGN.md 4 (the measurements: NOTEBOOK.md 5).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "svoxt_device.h"
#include "svoxt_lists.h"

#pragma clang fp contract(off)

#ifndef SVOXT_ROLES_CSUM
#define SVOXT_ROLES_CSUM 1               // 0 (exp/build_rev.sh ... -DSVOXT_ROLES_CSUM=0): no checksum is formed or compared -- what it costs
#endif

namespace svoxt {

// The packed leaf id (node * N^3 + u * N^2 + v * N + w, common.cuh:90-93) of the crossing at t, for
// tree.weight_accum (rt_kernel.cu:266-267, 309-311).  The acceleration grid does not carry it
// for leaves it resolves by itself (slot = ~0): those take the root descent -- only for samples
// that are composited, i.e. for the few coarse leaves that hold data.
template <bool N2>
__device__ __forceinline__ uint32_t leaf_slot(const TreeDev& tr, const Ray& r, float t, uint32_t slot) {
    if (slot != 0xffffffffu) return slot;
    Leaf lf;
    locate<N2>(tr, r.ox + t * r.dx, r.oy + t * r.dy, r.oz + t * r.dz, lf);
    return lf.slot;
}

// XF (SH only): per-leaf view rotations (tree.xform): the basis is re-evaluated
// for every composited sample with the leaf's matrix (rt_kernel.cu:283-291).
// RESUME (tail launch of the two-kernel forward, see shade_tile_kernel): only rays whose sample
// list overflowed (aux[q].x bit 31) do anything; they pick up the compositing state the shade
// kernel left in `out` (colour sums, transmittance in the alpha slot) and march on from
// aux[q].y, then finalise the pixel and the recorded final transmittance.
// LOBES (FMT_SH instances): the basis values are those of opt.format = SG or ASG with BD lobes (precalc_lobes).
// render_fwd_ray: one ray, tid = its launch thread (render_fwd_kernel's; fwd_finish_kernel walks tiles with it).
template <int FMT, int C, int BD, bool N2, bool REC, bool XF = false, bool RESUME = false, bool LOBES = false>
__device__ __forceinline__ void render_fwd_ray(const TreeDev& tr, const RaysDev& rays, const Opts& opt, float* __restrict__ out,
                                               const RecLists& L, uint4* __restrict__ aux, const int64_t tid) {
    static_assert(!XF || FMT == FMT_SH, "view rotations only matter for view-dependent formats");
    static_assert(!LOBES || (FMT == FMT_SH && !XF), "lobes stand in for an SH basis");
    static_assert(!(RESUME && REC), "the tail launch does not record");
    constexpr int K = (FMT == FMT_RGBA) ? (C + 1) : (C * BD + 1);
    __shared__ uint2 rstage[REC ? kRecBlock * kBlock : 1];
    __shared__ int32_t ltab[REC ? kMaxRecBlocks : 1];
    // (att, e_0, e_1, e_2) of the last <= 4 recorded samples of each ray, for the backward (C == 3)
    __shared__ float4 tstage[(REC && C == 3 && !XF) ? 4 * kBlock : 1];
    if constexpr (REC) rec_tab_init(ltab);
    const int S = L.S;
    int64_t cur_block = 0;
    const int64_t q = ray_of_thread(rays, tid);
    if (q >= rays.Q) return;
    float* o = out + q * (C + 1);
    float t_start = 0.f;
    if constexpr (RESUME) {
        const uint4 a = aux[q];
        if ((a.x & kRecOverflow) == 0u || a.w == kAuxStopped) return;   // (no tail, or the shade ended the ray: stop rule)
        t_start = __uint_as_float(a.y);
    }

    Ray r;
    if (!setup_ray(tr, rays, opt, q, r)) {
#pragma unroll
        for (int j = 0; j < C; ++j) o[j] = opt.background_brightness;
        o[C] = 0.f;
        if constexpr (REC) aux[q] = make_uint4(0u, 0u, __float_as_uint(1.f), 0u);
        return;
    }
    int nrec = 0;
    bool over = false;
    float t_resume = 0.f;
    float basis[BD > 0 ? BD : 1];
    float vd[3] = {0.f, 0.f, 0.f};
    if constexpr (FMT == FMT_SH) {
        load_vdir(rays, q, vd);
        if constexpr (LOBES) precalc_lobes<BD>(opt.format, tr, vd[0], vd[1], vd[2], basis);
        else if constexpr (!XF) precalc_basis<BD>(FMT_SH, BD, tr, vd[0], vd[1], vd[2], basis);
    }
    float acc[C];
#pragma unroll
    for (int j = 0; j < C; ++j) acc[j] = 0.f;

    float light = 1.f;
    float t = r.tmin;
    if constexpr (RESUME) {
#pragma unroll
        for (int j = 0; j < C; ++j) acc[j] = o[j];
        light = o[C];
        t = t_start;
    }
    bool stopped = false;
    // One composited sample (rt_kernel.cu:279-319); true: the ray ends here (early termination)
    // REC (r05): the lists serve the backward, which takes every sample with sigma > 0 and never stops early
    // (rt_kernel.cu:382, 456) -- so every such sample is RECORDED (and its hand-over formed), while it is COMPOSITED
    // only if the forward's own rules say so (sigma > sigma_thresh, :279; not after T <= stop_thresh, :313-319): the
    // march goes on to the end of the ray, `stopped` only ends the compositing.  With both thresholds 0 nothing
    // changes: the two sets are the same, and the stop rule then fires at T == 0 exactly, where every later weight is 0.
    auto shade = [&](const float (&row)[K], int32_t idx, float delta_t, float t_cur, uint32_t slot) -> bool {
        const float sigma = row[K - 1];
        if (!(sigma > (REC ? 0.f : opt.sigma_thresh))) return false;
        const bool comp = !REC || (sigma > opt.sigma_thresh && !stopped);
        bool recorded = false;
        if constexpr (REC) {
            bool room = nrec < S;
            if (room && (nrec & 7) == 0) {
                cur_block = rec_block_begin(L, ltab, tid >> 6, nrec >> 3);
                room = cur_block >= 0;
            }
            recorded = room;
            if (room) {
                rec_stage_put(rstage, (int)threadIdx.x, L.rec, cur_block, nrec, (uint32_t)idx, delta_t);
                ++nrec;
            } else if (!over) {
                over = true;
                t_resume = t_cur;
            }
        }
        const float att = pexpf(-delta_t * r.delta_scale * sigma);
        const float weight = light * (1.f - att);
        float ex[C];
        if constexpr (FMT == FMT_SH) {
            if constexpr (XF) rotated_sh_basis<BD>(tr, idx, vd, basis);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                ex[c] = pexpf(neg_sh_dot<BD>(basis, row + c * BD));
                if (comp) acc[c] = (float)((double)acc[c] + (double)weight / (1.0 + (double)ex[c]));
            }
        } else {
#pragma unroll
            for (int j = 0; j < C; ++j) {
                ex[j] = pexpf(-row[j]);
                if (comp) acc[j] = (float)((double)acc[j] + (double)weight / (1.0 + (double)ex[j]));
            }
        }
        if constexpr (REC && C == 3 && !XF) {
            // what the backward would otherwise gather the row and form again (both of its sweeps):
            // its own attenuation (exponent associated as in rt_kernel.cu:397) and the three exponentials.
            // (Measured r02: handing over sigma instead and letting the backward form its attenuation,
            // with cached instead of non-temporal stores: the forward saves nothing, the backward loses 0.02 ms.)
            if (recorded && L.terms != nullptr) {
                const int k = nrec - 1;
                tstage[(k & 3) * kBlock + threadIdx.x] =
                    make_float4(pexpf(-delta_t * sigma * r.delta_scale), ex[0], ex[1], ex[2]);
                if ((k & 3) == 3) terms_flush(tstage, (int)threadIdx.x, L.terms, cur_block, k);
            }
        }
        if (comp) {
            light *= att;
            if (tr.weight_accum != nullptr) atomicAdd(tr.weight_accum + leaf_slot<N2>(tr, r, t_cur, slot), weight);
            if (light <= opt.stop_thresh) {
                if constexpr (!REC) return true;
                stopped = true;                  // (REC: the march goes on recording)
            }
        }
        return false;
    };
    // (Measured r02 and removed: a loop two crossings ahead -- crossing k+1 located and its row
    // requested, the grid cell of crossing k+2 requested, THEN sample k shaded -- so that a
    // wavefront alone on its SIMD, which is what the last 100 us of this kernel consist of, has
    // loads in flight while it shades.  Bit-identical, 132 registers, 3 wavefronts per SIMD:
    // 0.246 -> 0.314 ms.  What it gains in the tail it loses, and more, while the CUs are full.)
    // Software pipeline: where the ray goes next depends on the leaf geometry only,
    // not on the leaf's features, so the descent of step k+1 is issued right after
    // the row load of step k and the two latencies overlap (memory operations of a
    // wavefront return in order: waiting for the younger descent load covers the row).
    Sample s;
    bool have = t < r.tmax;
    if (have) march_step<N2>(tr, r, opt.step_size, t, s);
    while (have) {
        float row[K];
        const bool valid = s.valid;
        if (valid) load_row<K>(tr.features + (int64_t)s.idx * K, row);   // whole row at once: sigma is its last element
        const float t_cur = t, delta_t = s.delta_t;
        const int32_t idx = s.idx;
        const uint32_t slot = s.leaf.slot;
        t = march_advance(t, delta_t);
        have = t < r.tmax;
        if (have) march_step<N2>(tr, r, opt.step_size, t, s);           // next descent, in flight with the row
        if (valid && shade(row, idx, delta_t, t_cur, slot)) { stopped = true; break; }
    }
    if (stopped) {
        const float scale = (float)(1.0 / (1.0 - (double)light));
#pragma unroll
        for (int j = 0; j < C; ++j) o[j] = acc[j] * scale;
    } else {
        const float bg = light * opt.background_brightness;
#pragma unroll
        for (int j = 0; j < C; ++j) o[j] = acc[j] + bg;
    }
    o[C] = 1.f - light;
    if constexpr (REC) {  // + the final transmittance, for the single-march backward
        rec_stage_finish(rstage, (int)threadIdx.x, L.rec, cur_block, nrec);
        if constexpr (C == 3 && !XF) {
            if (L.terms != nullptr && (nrec & 3)) terms_flush(tstage, (int)threadIdx.x, L.terms, cur_block, nrec - 1);
        }
        aux[q] = make_uint4((uint32_t)nrec | (over ? kRecOverflow : 0u), __float_as_uint(t_resume),
                            __float_as_uint(light), 0u);
        if (over) note_overflow(L);
    }
    if constexpr (RESUME) aux[q].z = __float_as_uint(light);
}

template <int FMT, int C, int BD, bool N2, bool REC, bool XF = false, bool RESUME = false, bool LOBES = false>
__global__ void __launch_bounds__(kBlock)
render_fwd_kernel(TreeDev tr, RaysDev rays, Opts opt, float* __restrict__ out,
                  RecLists L, uint4* __restrict__ aux) {
    if constexpr (RESUME) {
        if (no_ray_overflowed(L)) return;                    // (one scalar load: see kPoolOverflowWord)
    }
    render_fwd_ray<FMT, C, BD, N2, REC, XF, RESUME, LOBES>(tr, rays, opt, out, L, aux,
                                                           ((int64_t)blockIdx.x + rays.tile0) * kBlock + threadIdx.x);
}

// Generic fallback: any K, any format, component sub-range; accumulators in
// global memory exactly as the reference keeps them.
template <bool N2>
__global__ void __launch_bounds__(kBlock)
render_fwd_generic_kernel(TreeDev tr, RaysDev rays, Opts opt, int C, float* __restrict__ out) {
    const int64_t q = ray_of_thread(rays, (int64_t)blockIdx.x * kBlock + threadIdx.x);
    if (q >= rays.Q) return;
    float* o = out + q * (C + 1);
    const int K = tr.K;

    Ray r;
    if (!setup_ray(tr, rays, opt, q, r)) {
        for (int j = 0; j < C; ++j) o[j] = opt.background_brightness;
        o[C] = 0.f;
        return;
    }
    for (int j = 0; j < C; ++j) o[j] = 0.f;
    float basis[25];
    float vd[3];
    load_vdir(rays, q, vd);
    precalc_basis<0>(opt.format, opt.basis_dim, tr, vd[0], vd[1], vd[2], basis);
    float light = 1.f;
    float t = r.tmin;
    while (t < r.tmax) {
        Sample s;
        march_step<N2>(tr, r, opt.step_size, t, s);
        if (s.valid) {
            const float* row = tr.features + (int64_t)s.idx * K;
            const float sigma = row[K - 1];
            if (sigma > opt.sigma_thresh) {
                const float att = pexpf(-s.delta_t * r.delta_scale * sigma);
                const float weight = light * (1.f - att);
                if (tr.xform != nullptr) rotated_basis(tr, opt.format, opt.basis_dim, s.idx, vd, basis);
                if (opt.format != FMT_RGBA) {
                    for (int c = 0; c < C; ++c) {
                        const int off = c * opt.basis_dim;
                        float tmp = 0.f;
                        for (int i = opt.min_comp; i <= opt.max_comp; ++i) tmp += basis[i] * row[off + i];
                        o[c] = (float)((double)o[c] + (double)weight / (1.0 + (double)pexpf(-tmp)));
                    }
                } else {
                    for (int j = 0; j < C; ++j)
                        o[j] = (float)((double)o[j] + (double)weight / (1.0 + (double)pexpf(-row[j])));
                }
                light *= att;
                if (tr.weight_accum != nullptr) atomicAdd(tr.weight_accum + leaf_slot<N2>(tr, r, t, s.leaf.slot), weight);
                if (light <= opt.stop_thresh) {
                    const float scale = (float)(1.0 / (1.0 - (double)light));
                    for (int j = 0; j < C; ++j) o[j] *= scale;
                    o[C] = 1.f - light;
                    return;
                }
            }
        }
        t = march_advance(t, s.delta_t);
    }
    for (int j = 0; j < C; ++j) o[j] += light * opt.background_brightness;
    o[C] = 1.f - light;
}

// ---------------------------------------------------------------------------
// Forward as two kernels: march, then shade per tile (trace_ray, rt_kernel.cu:222-328)
// ---------------------------------------------------------------------------
//
// render_fwd_kernel is as long as its longest wavefront: the 8x8 tile whose rays graze the
// shell makes ~140 leaf crossings, and every crossing carries the whole shading of a sample
// (row gather, basis products, four exponentials, three double-precision divisions: ~450
// instructions, r02 ISA) in one dependent chain -- 1.8 us per crossing, 250 us for that
// wavefront while the bulk of the grid is done after 70 us.  Where a ray goes next depends on
// the leaf geometry alone, so the chain that must be sequential is the stepping: it gets a
// kernel of its own, and the shading becomes throughput work.
//
//   march_rec_kernel   one ray per lane: locate leaf, step, nothing else.  The sigma of a
//                      crossing (one 4-byte gather) is requested and looked at one crossing
//                      later -- memory operations of a wavefront return in order, so it has
//                      arrived with the next crossing's tree words and costs the chain nothing.
//                      Samples that pass (sigma > sigma_thresh) are recorded as (feature row,
//                      delta_t) in rec[k][q], the same lists the backward replays.
//   shade_tile_kernel  one workgroup of eight wavefronts per 64 rays (lane l of each = ray l).
//                      Per round, wavefronts 1..7 each take one list position of the 64 rays and
//                      form what depends on the sample alone: att = exp(-delta_t ds sigma) and
//                      e_c = exp(-x_c) (row gather, basis products, four exponentials); wavefront
//                      0 then runs what is sequential along a ray -- weight = T (1 - att),
//                      acc_c = float(double(acc_c) + double(weight) / (1.0 + double(e_c))),
//                      T *= att -- through the round's positions in list order, one round behind
//                      the others (double-buffered LDS, one barrier per round).  Operation for
//                      operation render_fwd_kernel: outputs are bit-identical.
//   render_fwd_kernel<..., RESUME>   rays whose list overflowed continue from where it ends.

// STOP: apply the early-termination rule (T <= stop_thresh ends the ray, rt_kernel.cu:313-319)
// while marching, with the transmittance formed exactly as the shade kernel forms it.  Off when
// the lists are for a backward, which wants every sample with sigma > 0 (:382,456).
// One bit per feature row: sigma > thresh (svoxt_sigma_mask_build).  A wavefront takes four groups of 64
// rows -- four independent loads per lane in flight, one 8-byte word per group -- 1024 rows per workgroup.
// (r04) fill (svoxt_sigma_mask_build_fill): the workgroups behind the first mask_blocks set fill_vec 16-byte words to all
// ones -- the -1 the block table, pool counters and tile states of a recording forward's lists start from: the fill that
// forward would otherwise launch by itself (a launch is ~4.5 us however little it does; the two are independent).
__global__ void __launch_bounds__(256)
sigma_mask_kernel(const float* __restrict__ features, int64_t M, int K, float thresh, unsigned long long* __restrict__ mask,
                  unsigned mask_blocks = ~0u, uint4* __restrict__ fill = nullptr, int64_t fill_vec = 0) {
    if (blockIdx.x >= mask_blocks) {
        const int64_t i0 = ((int64_t)(blockIdx.x - mask_blocks) * 256 + threadIdx.x) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j < fill_vec) fill[i0 + j] = make_uint4(~0u, ~0u, ~0u, ~0u);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int64_t word0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;      // first of this wavefront's four words
    const int64_t words = (M + 63) / 64;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t row = (word0 + j) * 64 + lane;
        v[j] = row < M ? features[row * K + (K - 1)] : thresh;       // (thresh > thresh is false: no bit)
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned long long b = __ballot(v[j] > thresh);
        if (lane == j && word0 + j < words) mask[word0 + j] = b;
    }
}

// (r04) The exponentials of RGBA-style rows of K = 8 / 16 / 32 floats, once per ROW: etab[row][c] = pexpf(-features[row][c])
// for the C = K - 1 feature columns, sigma itself in the last one.  The sigmoids of such a payload are view-independent
// (rt_kernel.cu:304, 420, 476: 1 / (1 + expf(-row[c]))), so what every sample of a row needs is a function of the row:
// at 1024 x 1024 / depth 9 the forward formed 13.4 M x 31 of these exponentials, the backward's sweeps 13.4 M x 31 and
// 5.9 M x 31 more -- for 4.7 M rows.  One streaming pass (the pass that builds the sigma bitmask reads every line of the
// table anyway: this kernel builds the mask too) writes the table; the shade kernel and both sweeps of the per-tile
// backward then read IT instead of the features -- the same bits, by construction.  A thread takes four floats of a row.
template <int K>
__global__ void __launch_bounds__(256)
exp_table_kernel(const float* __restrict__ features, int64_t M, float thresh, uint8_t* __restrict__ mask_bytes,
                 float* __restrict__ etab) {
    static_assert(K == 8 || K == 16 || K == 32, "row widths with a table");
    constexpr int P = K / 4, RW = 64 / P;            // threads per row, rows per wavefront
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t / P;
    const int part = (int)(t % P);
    bool bit = false;
    if (row < M) {
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(features) + t);      // read once: nothing of it is needed again
        v4f e;
        e.x = pexpf(-v.x); e.y = pexpf(-v.y); e.z = pexpf(-v.z);
        if (part == P - 1) { e.w = v.w; bit = v.w > thresh; }
        else e.w = pexpf(-v.w);
        reinterpret_cast<v4f*>(etab)[t] = e;
    }
    const unsigned long long b = __ballot(bit);
    if (mask_bytes != nullptr && (threadIdx.x & 63) == 0 && row < M) {
        uint32_t bits = 0u;                          // bit i: row (row + i) of this wavefront's RW rows
#pragma unroll
        for (int i = 0; i < RW; ++i) bits |= (uint32_t)((b >> (i * P + P - 1)) & 1ull) << i;
        uint8_t* dst = mask_bytes + (row >> 3);      // (row is a multiple of RW here, RW a multiple of 8)
#pragma unroll
        for (int j = 0; j < RW / 8; ++j) dst[j] = (uint8_t)(bits >> (8 * j));
    }
}

// MASK (no stop rule): whether a row's sigma exceeds sigma_thresh comes from one bit per feature row
// (svoxt_sigma_mask_build: M / 8 bytes, resident in L2) instead of a 4-byte gather that pulls a
// 64-byte line of the feature table -- half of this kernel's traffic, and HBM traffic once the table
// has left the Infinity Cache (r02, depth 9 / 32-float rows: forward 1.29 -> 1.15 ms with no gather at all).
// The march of ONE tile by one wavefront (lane = threadIdx.x & 63); rstage / ltab: this wavefront's LDS
// staging ([kRecBlock * 64] records) and block table ([kMaxRecBlocks]).
// Returns what the lane left in aux[q].x (list length | overflow flag); 0 for a lane without a ray or without samples.
// CSUM: *csum <- XOR of rec_hash over the records this lane wrote (tile_checksum_part folds the lanes; fwd_roles_kernel).
template <bool N2, bool STOP, int ACC, bool MASK, bool CSUM = false>
__device__ __forceinline__ uint32_t march_rec_tile(const TreeDev& tr, const RaysDev& rays, const Opts& opt, const RecLists& L,
                                                   uint4* __restrict__ aux, const uint32_t* __restrict__ sigma_mask,
                                                   int64_t tile, uint2* __restrict__ rstage, int32_t* __restrict__ ltab,
                                                   uint32_t* csum = nullptr) {
    static_assert(!(MASK && STOP), "the stop rule needs sigma itself");
    const int lane = (int)(threadIdx.x & 63);
    uint32_t cs = 0u;
    if (csum != nullptr) *csum = 0u;
    rec_tab_init(ltab);
    const int S = L.S;
    int64_t cur_block = 0;
    const int64_t tid = tile * 64 + lane;
    const int64_t q = ray_of_thread(rays, tid);
    if (q >= rays.Q) return 0u;
    Ray r;
    if (!setup_ray(tr, rays, opt, q, r)) {
        aux[q] = make_uint4(0u, 0u, __float_as_uint(1.f), 0u);
        return 0u;
    }
    const int K = tr.K;
    const float* __restrict__ sig_col = tr.features + (K - 1);
    int nrec = 0;
    uint32_t over = 0u;                  // kRecOverflow once the list is full
    float t_resume = 0.f;
    float light = 1.f;
    float t = r.tmin;
    // The crossing whose sigma is in flight.  No boolean lives across iterations (each would be a
    // lane mask merged with scalar instructions at every branch: r02, 87 of the loop's 186
    // instructions per crossing were such mask arithmetic): "nothing pending" is sigma = -inf,
    // "stop marching" is t = +inf.
    // MASK: what is in flight is the mask WORD of the crossing's row, and its bit is taken where the sample is
    // looked at -- behind the next crossing's cell, which was requested after it.  (Taken where the word is
    // requested, as r02 had it, the wavefront waits for the word before it can ask for the next cell: a second
    // trip to memory in the chain of every crossing that has a leaf with data.)  "Nothing pending" is word 0.
    const float kNone = -__builtin_inff();
    float p_sigma = kNone, p_dt = 0.f, p_t = 0.f;
    int32_t p_idx = 0;
    uint32_t p_word = 0u;
    auto pending = [&]() -> bool {
        if constexpr (MASK) return ((p_word >> (p_idx & 31)) & 1u) != 0u;
        // (!STOP without a bitmask: lists for a backward -- every sample with sigma > 0, whatever the forward's threshold)
        else return p_sigma > (STOP ? opt.sigma_thresh : 0.f);
    };
    while (t < r.tmax) {
        Sample s;
        march_step<N2, ACC>(tr, r, opt.step_size, t, s);
        const float t_cur = t;
        t = march_advance(t, s.delta_t);
        bool keep = true;
        if (pending()) {
            bool room = nrec < S;
            if (room && (nrec & 7) == 0) {
                cur_block = rec_block_begin(L, ltab, tid >> 6, nrec >> 3);
                room = cur_block >= 0;
            }
            if (room) {
                rec_stage_put(rstage, lane, L.rec, cur_block, nrec, (uint32_t)p_idx, p_dt);
                if constexpr (CSUM) cs ^= rec_hash(nrec, (uint32_t)p_idx, __float_as_uint(p_dt));
                ++nrec;
                if constexpr (STOP) {
                    light *= pexpf(-p_dt * r.delta_scale * p_sigma);
                    if (light <= opt.stop_thresh) { t = __builtin_inff(); keep = false; }
                }
            } else {        // list full: whoever consumes it marches on from this crossing
                over = kRecOverflow;
                t_resume = p_t;
                t = __builtin_inff();
                keep = false;
            }
        }
        p_sigma = kNone;
        p_word = 0u;
        if (keep && s.valid) {
            if constexpr (MASK) p_word = sigma_mask[s.idx >> 5];
            else p_sigma = sig_col[(int64_t)s.idx * K];
            p_idx = s.idx;
            p_dt = s.delta_t;
            p_t = t_cur;
        }
    }
    if (pending()) {    // the last crossing's sample
        bool room = nrec < S;
        if (room && (nrec & 7) == 0) {
            cur_block = rec_block_begin(L, ltab, tid >> 6, nrec >> 3);
            room = cur_block >= 0;
        }
        if (room) {
            rec_stage_put(rstage, lane, L.rec, cur_block, nrec, (uint32_t)p_idx, p_dt);
            if constexpr (CSUM) cs ^= rec_hash(nrec, (uint32_t)p_idx, __float_as_uint(p_dt));
            ++nrec;
        } else {
            over = kRecOverflow;
            t_resume = p_t;
        }
    }
    rec_stage_finish(rstage, lane, L.rec, cur_block, nrec);
    aux[q] = make_uint4((uint32_t)nrec | over, __float_as_uint(t_resume), __float_as_uint(1.f), 0u);
    if (over != 0u) note_overflow(L);
    if constexpr (CSUM) *csum = cs;
    return (uint32_t)nrec | over;
}

template <bool N2, bool STOP, int ACC, bool MASK = false>
__global__ void __launch_bounds__(kBlock)
march_rec_kernel(TreeDev tr, RaysDev rays, Opts opt, RecLists L, uint4* __restrict__ aux,
                 const uint32_t* __restrict__ sigma_mask = nullptr) {
    __shared__ uint2 rstage[kRecBlock * kBlock];
    __shared__ int32_t ltab[kMaxRecBlocks];
    const int64_t tile = (int64_t)blockIdx.x + rays.tile0;
    march_rec_tile<N2, STOP, ACC, MASK>(tr, rays, opt, L, aux, sigma_mask, tile, rstage, ltab);
}

// WTERMS (recording forwards, no view rotations): the wavefronts that form a sample's exponentials also
// leave them, with the attenuation in the backward's association (rt_kernel.cu:397), in L.terms
// (position-major: 1 KB per wavefront and list position) for the exact backward.
// COH: the lists are read with agent-scope loads (rec_get_coherent ...): for the workgroup that shades a
// tile inside the launch that marched it (fwd_roles_kernel).
constexpr int kShadeP = 7;                       // list positions per round: one per wavefront but the first
typedef float shade_v4f __attribute__((ext_vector_type(4)));
// LOBES (FMT_SH instances): the basis values are those of opt.format = SG or ASG with BD lobes (precalc_lobes).
// COH also: returns the XOR of rec_hash over the records this LANE loaded (its wavefront's list positions of its ray),
// for the caller to combine per ray and compare with the march's checksum; what it loads is made harmless first (a block id
// inside the pool, a row inside the table), so that a stale line costs a re-shade by the fallback launch, never a
// fault; `stale_test`: (test only) treat the first record of every ray as if a stale line had been read.
template <int FMT, int BD, bool XF, bool STOP, bool WTERMS, bool COH, bool LOBES = false>
__device__ __forceinline__ uint32_t shade_tile_body(const TreeDev& tr, const RaysDev& rays, const Opts& opt, const RecLists& L,
                                                    uint4* __restrict__ aux, float* __restrict__ out, int64_t tile,
                                                    shade_v4f (*terms)[kShadeP][64] /* [2]: (att, e_0, e_1, e_2) of a list position, per ray */,
                                                    bool stale_test = false, uint32_t* ax_used = nullptr /* COH: <- the aux.x this lane worked with */) {
    constexpr int C = 3, W = 8, P = kShadeP;
    static_assert(P == W - 1, "one wavefront runs along the rays");
    constexpr int K = (FMT == FMT_RGBA) ? (C + 1) : (C * BD + 1);
    constexpr int NB = (FMT == FMT_SH) ? BD : 1;
    typedef shade_v4f v4f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = ray_of_thread(rays, tile * 64 + lane);
    const bool inb = q < rays.Q;
    // (only the count lives through the rounds; wavefront 0 reads the entry again when it finalises the ray)
    uint32_t a_x = 0u;
    if (inb) a_x = COH ? aux_get_coherent(aux + q).x : aux[q].x;
    int nrec = (int)(a_x & ~kRecOverflow);
    if constexpr (COH) {
        nrec = min(nrec, L.S);                       // (a stale count must not run the loops away)
        if (ax_used != nullptr) *ax_used = a_x;
    }
    int maxn = nrec;
    for (int off = 32; off > 0; off >>= 1) maxn = max(maxn, __shfl_xor(maxn, off, 64));
    maxn = __builtin_amdgcn_readfirstlane(maxn);     // what decides the barrier count is scalar
    const int nround = (maxn + P - 1) / P;           // the same in every wavefront of the workgroup
    const int32_t tabreg = rec_tab_reg<COH>(L, tile, lane);
    uint32_t cs = 0u;                                // COH: XOR of rec_hash over the records this lane loaded

    float delta_scale = 0.f;
    float basis[NB];
    float vd[3] = {0.f, 0.f, 0.f};
    if (wave > 0 && nrec > 0) {
        Ray r;
        setup_ray(tr, rays, opt, q, r);               // a ray with samples hits the cube
        delta_scale = r.delta_scale;
        if constexpr (FMT == FMT_SH) {
            load_vdir(rays, q, vd);
            if constexpr (LOBES) precalc_lobes<BD>(opt.format, tr, vd[0], vd[1], vd[2], basis);
            else if constexpr (!XF) precalc_basis<BD>(FMT_SH, BD, tr, vd[0], vd[1], vd[2], basis);
        }
    }
    float light = 1.f, acc[C] = {0.f, 0.f, 0.f};
    bool stopped = false;

    for (int rd = 0; rd <= nround; ++rd) {
        if (wave > 0) {
            const int k = rd * P + (wave - 1);
            if (rd < nround && k < nrec) {
                int64_t blk = rec_block_u(L, tabreg, tile, k >> 3);
                if constexpr (COH) blk = min(max(blk, (int64_t)0), L.pool_blocks - 1);
                uint2 e = COH ? rec_get_coherent(L.rec + rec_index_in(blk, lane, k)) : rec_get(L.rec + rec_index_in(blk, lane, k));
                if constexpr (COH) {
                    if (stale_test && k == 0) e.x ^= 1u;
                    if constexpr (SVOXT_ROLES_CSUM != 0) cs ^= rec_hash(k, e.x, e.y);
                    e.x = min(e.x, (uint32_t)(tr.M - 1));
                }
                const int32_t idx = (int32_t)e.x;
                float row[K];
                load_row<K>(tr.features + (int64_t)idx * K, row);
                v4f tv;
                tv.x = pexpf(-__uint_as_float(e.y) * delta_scale * row[K - 1]);
                // (r05) lists that hold every sigma > 0 (a backward follows): a sample the forward's own threshold leaves
                // out (rt_kernel.cu:279) goes through the chain as one that does nothing -- att = 1: weight = T (1 - 1) = 0,
                // acc + 0 / (1 + e) = acc, T * 1 = T, bit for bit -- while its hand-over below is the backward's, untouched
                const bool below = !(row[K - 1] > opt.sigma_thresh);
                if constexpr (FMT == FMT_SH) {
                    if constexpr (XF) rotated_sh_basis<BD>(tr, idx, vd, basis);
                    float ex[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        ex[c] = pexpf(neg_sh_dot<BD>(basis, row + c * BD));
                    }
                    tv.y = ex[0]; tv.z = ex[1]; tv.w = ex[2];
                } else {
                    tv.y = pexpf(-row[0]); tv.z = pexpf(-row[1]); tv.w = pexpf(-row[2]);
                }
                {
                    v4f tl = tv;
                    tl.x = below ? 1.f : tv.x;
                    terms[rd & 1][wave - 1][lane] = tl;
                }
                if constexpr (WTERMS) {
                    typedef float v4g __attribute__((ext_vector_type(4)));
                    const float att_b = pexpf(-__uint_as_float(e.y) * row[K - 1] * delta_scale);
                    __builtin_nontemporal_store(v4g{att_b, tv.y, tv.z, tv.w},
                                                reinterpret_cast<v4g*>(L.terms + terms_index_pm(blk, lane, k)));
                }
            }
        } else if (rd > 0) {
            const int kb = (rd - 1) * P;
#pragma unroll
            for (int j = 0; j < P; ++j) {
                if (kb + j < nrec && !stopped) {
                    const v4f tv = terms[(rd - 1) & 1][j][lane];
                    const float weight = light * (1.f - tv.x);
                    acc[0] = (float)((double)acc[0] + (double)weight / (1.0 + (double)tv.y));
                    acc[1] = (float)((double)acc[1] + (double)weight / (1.0 + (double)tv.z));
                    acc[2] = (float)((double)acc[2] + (double)weight / (1.0 + (double)tv.w));
                    light *= tv.x;
                    // (the stop rule, rt_kernel.cu:313-319 -- also where the march did not apply it (!STOP: lists for a
                    // backward, or a threshold of 0): it then ends the COMPOSITING here; at 0 it fires at T == 0 exactly,
                    // where every later weight is 0 and the rescale is by 1)
                    if (light <= opt.stop_thresh) stopped = true;
                }
            }
        }
        lds_barrier();       // (the rounds hand over through LDS alone: the hand-over stores to memory stay in flight)
    }
    if (wave == 0 && inb) {
        // (the entry's other words stay as the march wrote them: only what the shade adds is stored -- a shade that
        // read a stale list length must not write it back)
        uint32_t* aw = reinterpret_cast<uint32_t*>(aux + q);
        float* o = out + q * (C + 1);
        if (stopped) {
            const float scale = (float)(1.0 / (1.0 - (double)light));
#pragma unroll
            for (int j = 0; j < C; ++j) o[j] = acc[j] * scale;
            o[C] = 1.f - light;
            // nothing left for the FORWARD's tail launch: scratch lists (the march stopped with the ray) drop the flag;
            // lists for a backward keep it -- the backward's tail marches what the list could not hold -- and say so in .w
            if constexpr (STOP) aw[0] = a_x & ~kRecOverflow;
        } else if (a_x & kRecOverflow) {             // state for render_fwd_kernel<..., RESUME>
#pragma unroll
            for (int j = 0; j < C; ++j) o[j] = acc[j];
            o[C] = light;
        } else {
            const float bg = light * opt.background_brightness;
#pragma unroll
            for (int j = 0; j < C; ++j) o[j] = acc[j] + bg;
            o[C] = 1.f - light;
        }
        aw[2] = __float_as_uint(light);              // the final transmittance, for the single-march backward
        if constexpr (!STOP) aw[3] = stopped ? kAuxStopped : 0u;
    }
    return cs;
}

// (r04) A queue entry is 64 bits, written and read whole: the tile id (| kTileEmpty) below, the checksum of the tile's
// lists as the march wrote them above (rec_hash / tile_checksum_part).  The shading workgroup folds what it LOADED the
// same way; a tile whose checksum does not match is not marked shaded, so the fallback launch shades it again from
// memory the kernel boundary has made visible: a stale read -- which this hand-over excludes by measured cache
// behaviour, not by the memory model (below) -- costs a re-shade instead of wrong pixels, lists' hand-over and gradients.
// Counters at words [ntiles_even + 256, + 264) of tile_state (value + 1: the fill leaves -1): tiles shaded in the launch,
// checksum mismatches, workgroups whose poll ran out, tiles dropped by the test flag, tiles the fallback launch shaded.
constexpr int kRolePolls = 20000;
constexpr int64_t kTileEmpty = 1 << 30;          // queue entry tile | kTileEmpty: no ray of the tile has a sample and its pixels are written
// (kRoleXcds, roles_even, roles_state_words -- the size of svoxt_sample_lists.tile_state -- live in svoxt_lists.h: svoxt_step.hip sizes workspaces with them)
__device__ __forceinline__ int my_xcc() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & (kRoleXcds - 1)); }   // HW_REG_XCC_ID, bits 3:0
enum { kRoleCtrShaded = 256, kRoleCtrMismatch = 257, kRoleCtrGaveUp = 258, kRoleCtrDropped = 259, kRoleCtrFallback = 260 };
// test-only behaviour of the shading workgroups (svoxt_sample_lists.flags >> 8: SVOXT_LISTS_TEST_*) and the fence form
enum { kRoleTestDrop = 1, kRoleTestNoPoll = 2, kRoleTestStale = 4, kRoleAgentFence = 8 };

// The marching wavefront's side: every store of the wavefront (records, block table, aux) acknowledged by the L2
// first, then a position from the queue's tail counter and the tile id into that entry.
// tile_state (optional): tiles whose entry is kTileShaded were shaded inside fwd_roles_kernel -- nothing to do
constexpr int32_t kTileShaded = 0x200;
template <int FMT, int BD, bool XF, bool STOP, bool WTERMS = false, bool LOBES = false>
__global__ void __launch_bounds__(512)
shade_tile_kernel(TreeDev tr, RaysDev rays, Opts opt, RecLists L,
                  uint4* __restrict__ aux, float* __restrict__ out, const int32_t* __restrict__ tile_state = nullptr) {
    static_assert(!LOBES || (FMT == FMT_SH && !XF), "lobes stand in for an SH basis");
    __shared__ shade_v4f terms[2][kShadeP][64];
    const int64_t tile = (int64_t)blockIdx.x + rays.tile0;
    if (tile_state != nullptr && tile_state[tile] == kTileShaded) return;       // (uniform: one tile per workgroup)
    shade_tile_body<FMT, BD, XF, STOP, WTERMS, false, LOBES>(tr, rays, opt, L, aux, out, tile, terms);
    if (tile_state != nullptr && threadIdx.x == 0)                              // (fallback of fwd_roles_kernel: counted)
        atomicAdd(const_cast<int32_t*>(tile_state) + roles_even(gridDim.x + rays.tile0) + kRoleCtrFallback, 1);
}

// (r04) What follows fwd_roles_kernel, as ONE small launch: the fallback shade of every tile the launch left unshaded
// and the tails of the rays whose lists overflowed.  As two launches over all tiles -- 80 000 wavefronts to find every
// tile shaded, 10 000 to find no ray overflowed -- they were 6.3 + 4.4 us of the 800 x 800 forward's 0.25 ms (VERDICT r03
// item 5).  A workgroup takes a contiguous run of tiles: its threads read the run's states at once and collect the
// unshaded ones (none, normally), it shades those (shade_tile_body: the whole workgroup per tile), then -- one scalar
// load says whether any ray of the batch overflowed -- its wavefronts walk the run's tiles with render_fwd_ray<RESUME>.
template <int FMT, int BD, bool WTERMS, bool LOBES = false, bool XF = false>
__global__ void __launch_bounds__(512)
fwd_finish_kernel(TreeDev tr, RaysDev rays, Opts opt, RecLists L, uint4* __restrict__ aux, float* __restrict__ out,
                  int32_t* __restrict__ tile_state, int ntiles) {
    __shared__ shade_v4f terms[2][kShadeP][64];
    constexpr int CAP = 64;
    __shared__ int32_t s_list[CAP];
    __shared__ int32_t s_n;
    const int run = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t0 = (int)blockIdx.x * run, t1 = min(t0 + run, ntiles);
    if (t0 >= t1) return;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (int i = t0 + (int)threadIdx.x; i < t1; i += 512) {
        if (tile_state[i] != kTileShaded) {
            const int pos = atomicAdd(&s_n, 1);
            if (pos < CAP) s_list[pos] = i;
        }
    }
    __syncthreads();
    const int n = s_n;                                       // (the same in every thread)
    int32_t* const fallback_ctr = tile_state + roles_even(ntiles) + kRoleCtrFallback;
    if (n > CAP) {                                           // more than the list holds: the run, tile by tile
        for (int tile = t0; tile < t1; ++tile) {
            if (tile_state[tile] == kTileShaded) continue;   // (uniform)
            shade_tile_body<FMT, BD, XF, false, WTERMS, false, LOBES>(tr, rays, opt, L, aux, out, tile, terms);
            if (threadIdx.x == 0) atomicAdd(fallback_ctr, 1);
            __syncthreads();
        }
    } else {
        for (int j = 0; j < n; ++j) {
            shade_tile_body<FMT, BD, XF, false, WTERMS, false, LOBES>(tr, rays, opt, L, aux, out, s_list[j], terms);
            if (threadIdx.x == 0) atomicAdd(fallback_ctr, 1);
            __syncthreads();
        }
    }
    if (no_ray_overflowed(L)) return;
    __syncthreads();                                         // (a tail resumes from what the shade of its tile left)
    for (int tile = t0 + (int)(threadIdx.x >> 6); tile < t1; tile += 8)
        render_fwd_ray<FMT, 3, BD, true, false, XF, true, LOBES>(tr, rays, opt, out, L, aux, (int64_t)tile * 64 + (threadIdx.x & 63));
}

// ---------------------------------------------------------------------------
// March and shade in ONE launch (r03): the forward as two kernels is two serial
// phases of ~0.14 ms each at 800 x 800 / depth 8 -- a march that is as long as the dependent chain of
// its longest ray and leaves the vector ALUs nearly idle while its last wavefronts finish, then a shade
// kernel that is throughput work -- and two kernels on two streams only overlap what is independent
// (exp/two_stream_halves.py: two half images 0.41 ms serial, 0.30 on two streams, the whole image 0.28 in
// one call).  Here one grid carries both roles:
//   workgroups [0, n_march)   8 wavefronts = the marches of 8 consecutive tiles (march_rec_tile, unchanged).
//                             A wavefront that has written its tile's lists publishes the tile: all its
//                             stores acknowledged (s_waitcnt vmcnt(0): they are in its XCD's L2), then the
//                             tile id goes into the READY QUEUE OF ITS XCD (a position from an atomic counter,
//                             the id stored there).
//   the workgroups behind     each shades ONE tile (shade_tile_body; lists read with agent-scope loads: from the
//                             L2, never from a stale line of the CU's vector cache) -- the next one of the
//                             queue of the XCD IT RUNS ON (hardware register XCC_ID, not an assumption about
//                             the dispatcher): thread 0 takes a position from the queue's second counter and
//                             polls that entry (bounded: kRolePolls polls of ~1 us) until a march has filled
//                             it.  Tiles are shaded in the order their marches finish, so no workgroup ever
//                             waits for a particular long march while finished tiles queue behind it
//                             (the first version -- shading workgroup i waits for tile i -- gained 0.018 ms
//                             of the 0.14 it could; r03 measurement).
// Termination: workgroups are dispatched in index order, so every march workgroup is resident or done before
// the first shading workgroup exists -- a waiting workgroup can never keep a march off the machine -- and the
// poll is bounded, so every wavefront reaches its end whatever happens.  Completeness: an XCD gets exactly as
// many shading workgroups as tiles were marched on it when workgroup b runs on XCD b mod (number of XCDs) --
// n_march is a multiple of 8 and shading workgroup i is active iff i / 8 < (tiles of march workgroups congruent
// to i mod 8).  If the dispatcher does anything else, a queue sees more consumers than entries (they give up
// after the poll budget) and another fewer (tiles stay unshaded): every tile not marked kTileShaded in
// tile_state is shaded by the fallback launch that follows (shade_tile_kernel with tile_state) -- slower,
// never wrong.  Same lists, terms and pixels bit for bit: the same device functions do the work.
//
// tile_state layout (int32, all -1 before the launch: lists_begin's fill): [0, T) per-tile state (T rounded up to
// even); then 16 counter slots of 32 words (slot x < 8: queue x's tail at +0, its head at +16: 64 bytes apart; "next
// position" - 1, like the block pool's counters; slot 8: the kRoleCtr* counters); then 8 queues of T 64-bit entries.
// What makes the lists visible to the consumer (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup
// visibility"): the wavefront's own stores are waited for -- asm volatile, which no compiler pass removes or moves: the
// guide's "Compiler hazard" -- so they have reached the L2 of the XCD this wavefront runs on; the queue is that XCD's
// (HW_REG_XCC_ID on both sides), so the consumer's CU sits behind the SAME L2, and every load of the handed-over bytes
// on its side is a global_load ... sc1 (never served by its own vector cache).  That is the guide's "sc1 loads in place
// of the acquire" with plain / nt producer stores, which the guide lists as valid only for sc1 (write-through) stores:
// the difference is cross-XCD visibility, which this hand-over never needs.  It remains measured behaviour, not an
// architectural guarantee -- hence the checksum.  agent_fence (test / measurement: SVOXT_LISTS_FWD_AGENT_FENCE) adds the
// guide's valid form in full: release fence at agent scope (buffer_wbl2 sc1) before the entry is stored, acquire fence
// (buffer_inv sc1) on the consumer before its first load.
__device__ __forceinline__ void publish_tile(int32_t* __restrict__ tile_state, int ntiles, int64_t tile, uint32_t csum,
                                             bool agent_fence) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (agent_fence) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if ((threadIdx.x & 63) == 0) {
        const int xcc = my_xcc();
        const int64_t base = roles_even(ntiles);
        int32_t* ctr = tile_state + base + xcc * 32;
        unsigned long long* queue = reinterpret_cast<unsigned long long*>(tile_state + base + 16 * 32) + (int64_t)xcc * ntiles;
        const int pos = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        __hip_atomic_store(queue + pos, (unsigned long long)(uint32_t)tile | ((unsigned long long)csum << 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}
// The shading workgroup's side (thread 0): the next entry of the queue of the XCD it runs on, or ~0 (nothing arrived).
__device__ __forceinline__ unsigned long long pop_tile(int32_t* __restrict__ tile_state, int ntiles, int polls) {
    const int xcc = my_xcc();
    const int64_t base = roles_even(ntiles);
    int32_t* ctr = tile_state + base + xcc * 32;
    unsigned long long* queue = reinterpret_cast<unsigned long long*>(tile_state + base + 16 * 32) + (int64_t)xcc * ntiles;
    unsigned long long ent = ~0ull;
    const int idx = __hip_atomic_fetch_add(ctr + 16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (idx < ntiles) {
        for (int p = 0; p < polls; ++p) {
            ent = __hip_atomic_load(queue + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ent != ~0ull) break;
            __builtin_amdgcn_s_sleep(16);
        }
    }
    return ent;
}

// (r04) Which tiles a marching workgroup takes.  Workgroup b runs on XCD b mod 8 (observed; nothing depends on it but
// speed), and left alone marching workgroup b takes tiles 8 b .. 8 b + 7: every XCD gets every eighth 64 x 8 pixel strip
// of every tile row, so no two vertically neighbouring strips -- whose rays cross the same leaves -- share an L2, and
// each of the eight L2s fetches (almost) every feature row and tree word of the view (r03 PMC: 373 MB fetched for
// ~110 MB of distinct lines).  RolesMap.gb > 0: the groups of 8 tiles are cut into BANDS of gb consecutive groups (two
// tile rows of an 800-pixel image), band B goes to XCD B mod 8, and the workgroups of an XCD walk its bands in order:
// a band's neighbours in the image are on the same L2, bands of one XCD are spread over the whole image (a contiguous
// eighth per XCD leaves seven XCDs waiting for the one that has the shell's middle: r03).  cnt[x]: the tiles XCD x's
// marching workgroups hold = the shading workgroups its queue needs.  gb == 0: the plain order.
// (the last round of bands is cut to gb_last groups each, so that every XCD ends with the same share of what is left)
struct RolesMap {
    int gb, full, gb_last;       // groups per band, rounds of 8 full bands, groups per band of the last round
    int cnt[8];
};
__host__ __device__ inline int64_t roles_group_of(const RolesMap& m, int b) {      // marching workgroup b -> its group of 8 tiles
    if (m.gb <= 0) return b;
    const int x = b & 7, s = b >> 3;
    if (s < m.full * m.gb) return ((int64_t)(s / m.gb) * 8 + x) * m.gb + s % m.gb;
    return (int64_t)m.full * 8 * m.gb + (int64_t)x * m.gb_last + (s - m.full * m.gb);
}

// XF (r04; SH up to 9 basis functions): per-leaf view rotations -- the shading role evaluates a record's basis from its
// leaf's matrix (shade_tile_body<..., XF>) and its hand-over holds the exponentials of THAT basis (grad_fused_kernel<..., 3, ..., XF>).
// (r05, measured and not kept -- exp/r05/*.diff.txt, NOTEBOOK.md 4.1: (a) the shading role PERSISTENT, at most as many shading
// workgroups as the chip holds, each taking tiles until a stop entry, tiles without samples not queued at all: forward
// 0.2485 -> 0.2554 ms -- a pop, two dependent round trips, costs what the dispatch of a workgroup cost, and as a loop the
// body keeps ~40 registers of invariants alive unless they are formed per tile behind asm barriers (79 -> 120 registers,
// two workgroups per CU: 0.33 ms); (b) the gathering wavefronts' record of the next round requested a round ahead:
// 0.2476 -> 0.2494; (c) wave 0's chain with one of its three double-precision quotients (wrong pixels, timing only):
// 0.2559 -> 0.2543.  None of the three is what the kernel waits for: it is as long as its longest march (~155 us: 133
// crossings of ~1.16 us) plus that tile's own shade (~45-58 us: ~19 rounds of gather -> barrier).  (d) The march itself: the
// grid cell of a LATER crossing requested on speculation (where that crossing starts if the leaves keep their size), one or
// two crossings ahead, in front of or right behind the crossing's own request: forward 0.247 -> 0.294-0.306 ms -- the ~15
// instructions of the guess sit in every crossing's chain and the speculative line competes with the real one.  (e) The
// march TWO crossings at a time: the step across the crossing's leaf formed from the level of the leaf before it while the
// cell is on its way, the next point's cell requested right behind it and USED when the level was the one assumed (bit for
// bit the reference's march either way; exp/r05/march_two_crossings_at_a_time.diff.txt): forward 0.237 -> 0.270 ms here (89
// registers), and 0.80 -> 0.83 at 1024 x 1024 / depth 9 where the march is a kernel of its own at 64 -- runs of equal
// levels are short on these trees, and every iteration pays two points and two requests.)
template <int FMT, int BD, int ACC, bool WTERMS, bool LOBES = false, bool XF = false>
__global__ void __launch_bounds__(512)
fwd_roles_kernel(TreeDev tr, RaysDev rays, Opts opt, RecLists L, uint4* __restrict__ aux, float* __restrict__ out,
                 const uint32_t* __restrict__ sigma_mask, int32_t* __restrict__ tile_state, int n_march, int ntiles,
                 int tflags /* kRoleTest* | kRoleAgentFence: 0 in production */, RolesMap map) {
    constexpr int kMarchBytes = 8 * (kRecBlock * 64 * (int)sizeof(uint2) + kMaxRecBlocks * (int)sizeof(int32_t));
    constexpr int kShadeBytes = 2 * kShadeP * 64 * (int)sizeof(shade_v4f);
    __shared__ __attribute__((aligned(16))) unsigned char lds[kMarchBytes > kShadeBytes ? kMarchBytes : kShadeBytes];
    __shared__ unsigned long long s_ent;
    const int wave = threadIdx.x >> 6;
    const bool agent_fence = (tflags & kRoleAgentFence) != 0;
    int32_t* const ctrs = tile_state + roles_even(ntiles);
    if ((int)blockIdx.x < n_march) {
        const int64_t tile = roles_group_of(map, (int)blockIdx.x) * 8 + wave;
        if (tile >= ntiles) return;
        uint2* rstage = reinterpret_cast<uint2*>(lds) + wave * (kRecBlock * 64);
        int32_t* ltab = reinterpret_cast<int32_t*>(lds + 8 * kRecBlock * 64 * sizeof(uint2)) + wave * kMaxRecBlocks;
        // the kernel is as long as its longest march plus that tile's shade: the marching wavefronts' few
        // instructions per crossing go ahead of the shading wavefronts' many (issue priority 3 of 0..3)
        __builtin_amdgcn_s_setprio(3);
        uint32_t cs_lane;
        const uint32_t ax = march_rec_tile<true, false, ACC, true, SVOXT_ROLES_CSUM != 0>(tr, rays, opt, L, aux, sigma_mask, tile, rstage, ltab, &cs_lane);
        // Two tiles in three have no sample at all (800 x 800, depth-8 shell).  What the shade would leave for such a
        // tile -- the background in every pixel, by the operations shade_tile_body performs for a ray without records
        // (light = 1, acc = 0) -- the marching wavefront leaves itself, and the queue entry says so: the shading
        // workgroup that takes it ends after its pop instead of fetching 64 list lengths to find nothing to do.
        // The kernel's time is the time its workgroups hold their slots (three per CU: 768 -- measured r03: 1 250
        // marching workgroups x ~45 us + 3 439 shadings x ~30 us + 6 561 shadings of nothing x ~4 us = 768 x 0.24 ms),
        // so what an empty tile no longer costs is what the kernel returns: 0.243 -> 0.230 ms.
        const bool empty = __ballot(ax != 0u) == 0ull;
        if (empty) {
            const int64_t q = ray_of_thread(rays, tile * 64 + (threadIdx.x & 63));
            if (q < rays.Q) {
                const float light = 1.f, acc = 0.f;
                const float bg = light * opt.background_brightness;
                float* o = out + q * 4;
                o[0] = acc + bg; o[1] = acc + bg; o[2] = acc + bg;
                o[3] = 1.f - light;
            }
            if ((threadIdx.x & 63) == 0) tile_state[tile] = kTileShaded;
        }
        // (a lane without a ray, or whose ray misses the cube, returned ax = 0 and recorded nothing: it folds in as zero)
        uint32_t csum = 0u;
        if constexpr (SVOXT_ROLES_CSUM != 0) csum = empty ? 0u : tile_checksum(ax != 0u ? cs_lane : 0u, ax, (int)(threadIdx.x & 63));
        publish_tile(tile_state, ntiles, empty ? (tile | kTileEmpty) : tile, csum, agent_fence);      // (the queue addresses are formed behind the march: nothing of them lives across it)
        return;
    }
    // is this shading workgroup one of those its XCD needs?  (see above; the same in every wavefront)
    const int i = (int)blockIdx.x - n_march, a = i & 7, b = i >> 3;
    if (b >= map.cnt[a]) return;                                 // (cnt: tiles of the marching workgroups congruent to a mod 8)
    if (threadIdx.x == 0) s_ent = pop_tile(tile_state, ntiles, (tflags & kRoleTestNoPoll) ? 1 : kRolePolls);
    __syncthreads();
    const unsigned long long ent = s_ent;
    if (ent == ~0ull) {                                          // nothing arrived: the tile this entry will name is left to the fallback launch
        if (threadIdx.x == 0) atomicAdd(ctrs + kRoleCtrGaveUp, 1);
        return;
    }
    const int64_t tile = (int64_t)(uint32_t)ent;
    if (tile >= ntiles) return;                                  // tile | kTileEmpty: finished by its march
    if ((tflags & kRoleTestDrop) && tile % 3 == 0) {             // (test: a consumer that takes its tile and does nothing)
        if (threadIdx.x == 0) atomicAdd(ctrs + kRoleCtrDropped, 1);
        return;
    }
    if (agent_fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    uint32_t ax = 0u;
    static_assert(!XF || (FMT == FMT_SH && !LOBES && BD <= 9), "view rotations: SH rows");
    const uint32_t part = shade_tile_body<FMT, BD, XF, false, WTERMS, true, LOBES>(
        tr, rays, opt, L, aux, out, tile, reinterpret_cast<shade_v4f (*)[kShadeP][64]>(lds), (tflags & kRoleTestStale) && tile % 5 == 0, &ax);
    // what this workgroup loaded, folded like the march folded what it wrote: per ray first (the eight wavefronts'
    // words of a lane; the shade's double buffer is free behind its last barrier), then over the tile
    uint32_t* s_part = reinterpret_cast<uint32_t*>(lds);
    s_part[threadIdx.x] = part;
    lds_barrier();
    if (wave == 0) {
        const int lane = (int)threadIdx.x;
        uint32_t x = 0u;
#pragma unroll
        for (int w = 0; w < 8; ++w) x ^= s_part[w * 64 + lane];
        const uint32_t c = SVOXT_ROLES_CSUM != 0 ? tile_checksum(ax != 0u ? x : 0u, ax, lane) : (uint32_t)(ent >> 32);
        if (lane == 0) {
        if (c == (uint32_t)(ent >> 32)) {
            tile_state[tile] = kTileShaded;                      // (read by the fallback launch: after this kernel)
            atomicAdd(ctrs + kRoleCtrShaded, 1);
        } else {
            atomicAdd(ctrs + kRoleCtrMismatch, 1);               // stale (or, with the test flag, made to look so): the fallback launch shades it again
        }
        }
    }
}

// The shade kernel for RGBA-style rows of K = 8, 16 or 32 floats (C = K - 1 feature channels and
// sigma: BASELINE configs[3] is K = 32): CHANNELS on lanes.  A wavefront takes 64 / K rays; lane
// (g, c) is channel c of ray g.  Per list position a lane reads ITS float of the sample's row --
// the K lanes of a ray read one contiguous row -- and forms one exponential; the sigma lane
// (c = K - 1) forms the attenuation instead and hands it to its group (one cross-lane read), then
// every channel lane runs its own chain  acc = float(double(acc) + double(T (1 - att)) / (1.0 +
// double(e)))  along the ray: the same operations in the same order as render_fwd_kernel, bit for
// bit, with every lane busy, ~30 registers, no LDS, no barrier.  (render_fwd_kernel<RGBA, 31>
// keeps 31 accumulators and the 32-float row per lane: 1.1 wavefronts per SIMD on average and the
// VALU half idle at 1024 x 1024, depth 9 -- r02 PMC -- because a wavefront shades all 31
// channels of whichever of its 64 rays have a sample.)
// FAST (opt-in tolerance mode, SVOXT_LISTS_NATIVE_MATH): the lists are the exact march's; the shading
// takes its exponentials with v_exp_f32 (nexpf) and the quotient with the hardware reciprocal,
// acc = fma(w, rcp(1 + e), acc) -- a dozen instructions per channel and sample where the bit-exact
// replica of expf and the double-precision divide are about fifty (r02 PMC: this kernel 100 % VALU-bound).
#ifndef SVOXT_CHAN_SWIZZLE
#define SVOXT_CHAN_SWIZZLE 1
#endif
// ETAB (exact mode, r04): the rows are read from tr.etab (exp_table_kernel): a channel lane's float IS its exponential,
// and the attenuations of a block's eight records -- the only exponentials left -- are formed by eight lanes of the
// ray's group at once (lane c takes record c mod 8: sigma of that record from the sigma lane, delta_t from the lane's
// own copy of the record line) instead of eight times by every lane; the quotients w / (1 + e) by div_unit_range.
// Per record and lane ~30 vector instructions where the exact instance without the table executes 74 (r03 PMC).
// (r05, measured and not kept: the next block's record line requested behind this block's row gathers -- 74 registers, 6
// wavefronts per SIMD instead of 7: config 4 forward 1.047 -> 1.067 ms.  The kernel does not wait for its records.)
template <int K, bool STOP, bool FAST, bool ETAB = false>
__global__ void __launch_bounds__(256)
shade_chan_kernel(TreeDev tr, RaysDev rays, Opts opt, RecLists L,
                  uint4* __restrict__ aux, float* __restrict__ out, int keep_over = 0) {
    static_assert(K == 8 || K == 16 || K == 32, "row widths with a channel-lane instance");
    static_assert(!(ETAB && FAST), "the table holds the exact exponentials");
    constexpr int RPW = 64 / K;                                  // rays per wavefront
    const int lane = threadIdx.x & 63;
    const int c = lane & (K - 1);
    const int sig_lane = lane | (K - 1);                         // the sigma lane of this lane's ray
    // t: the launch thread of march_rec_kernel that holds this ray (tile t >> 6, lane t & 63)
    // (r03) Workgroup b runs on XCD b mod 8, and a tile's 64 rays are 64 / RPW wavefronts = 8 consecutive workgroups
    // for K = 32: left alone, the rays of ONE tile -- which share their feature rows -- are shaded on all eight XCDs,
    // each behind its own L2.  Within every 8 tiles' worth of workgroups the XCD field and the field of the workgroup
    // within its tile change places (K = 32: the two 3-bit fields of every 64 workgroups): XCD x then takes tile x of the
    // group, whole, and neighbouring tiles still go to different XCDs.  (A contiguous eighth of the image per XCD: 1.00 -> 1.37 ms --
    // the image's middle rows hold most of the samples.  Two / four tiles in a row per XCD, SVOXT_CHAN_SWIZZLE = 2 / 4:
    // the kernel's fetches 751 -> 719 / 704 MB, its time 0.669 -> 0.685 / 0.72 ms native: one tile it is.)
    constexpr unsigned kRun = (unsigned)(K / 4) * SVOXT_CHAN_SWIZZLE;   // consecutive workgroups one XCD takes: a tile's (64 rays / (4 wavefronts x 64 / K rays))
    const unsigned b0 = blockIdx.x;
    unsigned wg = b0;
    if (kRun != 0u && (b0 | (8u * kRun - 1u)) < gridDim.x) {
        const unsigned base = b0 & ~(8u * kRun - 1u), x = b0 & 7u, m = (b0 >> 3) & (kRun - 1u);
        wg = base + x * kRun + m;
    }
    const int64_t t = (int64_t)rays.tile0 * 64 + ((int64_t)wg * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RPW + (lane / K);
    const int64_t q = ray_of_thread(rays, t);
    const bool inb = q < rays.Q;
    uint4 a = make_uint4(0u, 0u, 0u, 0u);
    if (inb) a = aux[q];
    const int nrec = (int)(a.x & ~kRecOverflow);
    int maxn = nrec;
    for (int off = 32; off >= K; off >>= 1) maxn = max(maxn, __shfl_xor(maxn, off, 64));
    maxn = __builtin_amdgcn_readfirstlane(maxn);
    float ds = 0.f;
    if (nrec > 0) {
        Ray r;
        setup_ray(tr, rays, opt, q, r);                          // a ray with samples hits the cube
        ds = r.delta_scale;
    }
    const int32_t tabreg = rec_tab_reg(L, t >> 6, lane);
    float light = 1.f, acc = 0.f;
    bool stopped = false;
    // A block of 8 records is the ray's own 64-byte line: fetch it whole, request the 8 rows it
    // names back to back (the K lanes of a ray: one contiguous row each), form the 8 exponentials --
    // all independent -- and only then run the chain along the ray.  (One record, one row, one step
    // at a time the kernel was a chain of two dependent loads per sample: 1.12 ms at 1024 x 1024,
    // depth 9, K = 32, and no faster with the float quotient.)
    // (r03) Branch-free: a list position past the ray's last record is made a sample that does nothing --
    // its row index is clamped to row 0 (a load nobody waits for the value of), its exponent to 0, so that
    // e = att = 1, weight = T (1 - 1) = 0, acc + 0 = acc and T * 1 = T, bit for bit (acc never holds -0: it
    // starts at +0 and only receives terms >= 0) -- instead of a compare, an exec-mask save / restore and a
    // branch around every load and every step of the chain (r02 ISA: ~15 of the FAST loop's 31 vector
    // instructions per record).  The row address is ONE v_mad_u64_u32 (row index x row bytes + the lane's
    // column address).
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const bool is_sig = c == K - 1;
    const char* __restrict__ col_base = reinterpret_cast<const char*>(ETAB ? tr.etab : tr.features) + 4 * c;
    constexpr float kNegLog2e = -1.44269504088896341f;
    // FAST: exponent of 2 = x * sc with sc = fma(dt, mult, base): sigma lane dt * (-ds log2 e), channel lanes -log2 e
    const float sc_mult = is_sig ? ds * kNegLog2e : 0.f, sc_base = is_sig ? 0.f : kNegLog2e;
    for (int kb = 0; kb < maxn; kb += kRecBlock) {
        uint32_t idx[kRecBlock];
        float dt[kRecBlock], ex[kRecBlock];
        const int n_here = nrec - kb;                            // records of this ray from here on (<= 0: none)
        const int64_t blk = rec_block_u(L, tabreg, t >> 6, kb >> 3);
        // The ray's line of this block, read whether or not the ray still has records here: kb < maxn, so
        // some ray of this wavefront -- of the same tile -- started the block, i.e. the block exists and
        // every lane's line of it is memory of the lists (stale bits where nothing was recorded).
        const v4u* line = reinterpret_cast<const v4u*>(L.rec + rec_index_in(blk, t, kb));
#pragma unroll
        for (int j = 0; j < kRecBlock / 2; ++j) {
            const v4u w = __builtin_nontemporal_load(line + j);
            idx[2 * j] = w.x; dt[2 * j] = __uint_as_float(w.y);
            idx[2 * j + 1] = w.z; dt[2 * j + 1] = __uint_as_float(w.w);
        }
        float x[kRecBlock];
#pragma unroll
        for (int j = 0; j < kRecBlock; ++j) {
            const uint32_t row = j < n_here ? idx[j] : 0u;       // (slots past the count hold stale bits)
            x[j] = *reinterpret_cast<const float*>(col_base + (size_t)row * (size_t)(K * 4));
        }
        float att_mine = 1.f;                                    // ETAB: the attenuation of record (c mod 8) of this lane's ray
        if constexpr (ETAB) {
            // sigma of record j sits in the sigma lane's x[j]; lane c needs the one of record c mod 8 (and its own delta_t)
            const int jm = c & (kRecBlock - 1);
            float sg = 0.f, dm = 0.f;
#pragma unroll
            for (int j = 0; j < kRecBlock; ++j) {
                const float sj = __shfl(x[j], sig_lane, 64);
                sg = jm == j ? sj : sg;
                dm = jm == j ? dt[j] : dm;
            }
            // (r05: sg > sigma_thresh -- lists that hold every sigma > 0 for a backward: a sample below the forward's own
            // threshold is one that does nothing, like a position past the list)
            att_mine = pexpf((jm < n_here && sg > opt.sigma_thresh) ? -dm * ds * sg : 0.f);
#pragma unroll
            for (int j = 0; j < kRecBlock; ++j) ex[j] = j < n_here ? x[j] : 1.f;      // (a position past the list: e = 1, as pexpf(0))
        } else {
#pragma unroll
        for (int j = 0; j < kRecBlock; ++j) {
            if constexpr (FAST) {
                const float e2 = x[j] * __builtin_fmaf(dt[j], sc_mult, sc_base);
                ex[j] = __builtin_amdgcn_exp2f((j < n_here && !(is_sig && !(x[j] > opt.sigma_thresh))) ? e2 : 0.f);
            } else {
                const float arg = is_sig ? ((x[j] > opt.sigma_thresh) ? -dt[j] * ds * x[j] : 0.f) : -x[j];
                ex[j] = pexpf(j < n_here ? arg : 0.f);
            }
        }
        }
#pragma unroll
        for (int j = 0; j < kRecBlock; ++j) {
            const float att = ETAB ? __shfl(att_mine, (lane & ~(K - 1)) + j, 64)       // (K >= 8: lane j of the group exists)
                                   : __shfl(ex[j], sig_lane, 64);                      // every lane takes part
            if (!STOP || !stopped) {
                const float weight = light * (1.f - att);
                if constexpr (FAST) acc = __builtin_fmaf(weight, __builtin_amdgcn_rcpf(1.f + ex[j]), acc);
                else if constexpr (ETAB) acc = (float)((double)acc + div_unit_range((double)weight, 1.0 + (double)ex[j]));
                else acc = (float)((double)acc + (double)weight / (1.0 + (double)ex[j]));
                light *= att;
                if constexpr (STOP) {
                    if (j < n_here && light <= opt.stop_thresh) stopped = true;
                }
            }
        }
    }
    if (!inb) return;
    const bool over = (a.x & kRecOverflow) != 0u && !stopped;    // state for render_fwd_kernel<..., RESUME>
    float v;
    if (c < K - 1) {
        if (stopped) v = acc * (float)(1.0 / (1.0 - (double)light));
        else if (over) v = acc;
        else v = acc + light * opt.background_brightness;
    } else {
        v = over ? light : 1.f - light;
        // (keep_over: lists for a backward -- the stop rule ended the forward's compositing only; the backward's tail
        // still marches what the list could not hold, the forward's tail is told to leave the ray alone)
        if (stopped && !keep_over) a.x &= ~kRecOverflow;
        a.z = __float_as_uint(light);
        a.w = (stopped && keep_over) ? kAuxStopped : 0u;
        aux[q] = a;
    }
    out[q * K + c] = v;
}

// The tail launch for those rows: rays whose list overflowed (1.3 % at 1024 x 1024, depth 9,
// S = 96 -- but as render_fwd_kernel<RGBA, 31, ..., RESUME> they cost 0.25 ms, a lane shading 31
// channels per sample) continue with the same lane layout as shade_chan_kernel: the K lanes of a
// ray march it together (the same steps in every lane: redundant, but a march is latency, not
// work) and each shades its own channel.  State in and out as for the RESUME launch.
template <int K, bool N2, bool FAST>
__global__ void __launch_bounds__(256)
tail_chan_kernel(TreeDev tr, RaysDev rays, Opts opt, uint4* __restrict__ aux, float* __restrict__ out, RecLists L) {
    // One workgroup (four wavefronts) per 64-ray tile; with lists that hold every sample -- the
    // usual case since they are pooled -- the launch is one scalar load (kPoolOverflowWord), else one look at the
    // tile's 64 aux entries.
    if (no_ray_overflowed(L)) return;
    constexpr int RPW = 64 / K;
    __shared__ int any_over;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x + rays.tile0;
    if (wave == 0) {
        const int64_t q0 = ray_of_thread(rays, tile * 64 + lane);
        bool ov = false;
        if (q0 < rays.Q) { const uint4 a0 = aux[q0]; ov = (a0.x & kRecOverflow) != 0u && a0.w != kAuxStopped; }
        const bool a0 = __any(ov);
        if (lane == 0) any_over = a0 ? 1 : 0;
    }
    __syncthreads();
    if (!any_over) return;
    const int c = lane & (K - 1);
    const int sig_lane = lane | (K - 1);
    for (int grp = wave; grp < 64 / RPW; grp += 4) {          // RPW rays per wavefront and turn
        const int64_t t0 = tile * 64 + grp * RPW + (lane / K);
        const int64_t q = ray_of_thread(rays, t0);
        uint4 a = make_uint4(0u, 0u, 0u, 0u);
        if (q < rays.Q) a = aux[q];
        const bool tail = (a.x & kRecOverflow) != 0u && a.w != kAuxStopped;     // (kAuxStopped: the shade ended the ray by the stop rule)
        bool alive = tail;
        if (!__any(alive)) continue;
        Ray r;
        float light = 1.f, acc = 0.f, t = 0.f, tmax = -1.f;
        bool stopped = false;
        if (alive) {
            setup_ray(tr, rays, opt, q, r);
            t = __uint_as_float(a.y);
            tmax = r.tmax;
            light = out[q * K + (K - 1)];
            if (c < K - 1) acc = out[q * K + c];
        }
        while (__any(alive && t < tmax)) {
            const bool go = alive && t < tmax;
            float x = 0.f, dt = 0.f;
            bool valid = false;
            if (go) {
                Sample s;
                march_step<N2>(tr, r, opt.step_size, t, s);
                dt = s.delta_t;
                valid = s.valid;
                if (valid) x = tr.features[(int64_t)s.idx * K + c];
                t = march_advance(t, s.delta_t);
            }
            const float sigma = __shfl(x, sig_lane, 64);
            const bool active = go && valid && sigma > opt.sigma_thresh;
            float ex = 1.f;
            if (active) {
                const float arg = c == K - 1 ? -dt * r.delta_scale * x : -x;
                ex = FAST ? nexpf(arg) : pexpf(arg);
            }
            const float att = __shfl(ex, sig_lane, 64);
            if (active) {
                const float weight = light * (1.f - att);
                if constexpr (FAST) acc = __builtin_fmaf(weight, __builtin_amdgcn_rcpf(1.f + ex), acc);
                else acc = (float)((double)acc + (double)weight / (1.0 + (double)ex));
                light *= att;
                if (light <= opt.stop_thresh) { stopped = true; alive = false; }
            }
        }
        if (!tail) continue;
        float v;
        if (c < K - 1) {
            v = stopped ? acc * (float)(1.0 / (1.0 - (double)light)) : acc + light * opt.background_brightness;
        } else {
            v = 1.f - light;
            aux[q].z = __float_as_uint(light);
        }
        out[q * K + c] = v;
    }
}

}  // namespace svoxt
