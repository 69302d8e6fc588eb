// svoxt_bwd_kernels.h -- volume_render backward (trace_ray_backward, rt_kernel.cu:331-496): the
// per-ray kernel (marching or replaying sample lists, shaped atomics through LDS staging; its
// two-kernel, tail-only and one-sigmoid-pass forms), the per-tile kernels (grad_merge_kernel,
// grad_fused_kernel for 3-channel payloads, grad_wide_kernel for RGBA rows of 8 / 16 / 32 floats)
// and the generic fallbacks.  See the file header of svoxt_kernels.hip and DESIGN.md 4 (the measurements: NOTEBOOK.md 5).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "svoxt_device.h"
#include "svoxt_lists.h"

#pragma clang fp contract(off)

#ifndef SVOXT_WIDE_STAGE_ROWS
#define SVOXT_WIDE_STAGE_ROWS 2          // rows per lane and stage of grad_wide_kernel's reduce (1: 1.59 ms, 2: 1.55, 3: 1.77, 4: 1.82 backward)
#endif
#ifndef SVOXT_WIDE_XCD_ROWS
#define SVOXT_WIDE_XCD_ROWS 1
#endif
#ifndef SVOXT_WIDE_ETAB_WAVES
#define SVOXT_WIDE_ETAB_WAVES 8          // wavefronts per SIMD the table instance of grad_wide_kernel is compiled for (8: four workgroups per CU)
#endif

namespace svoxt {

// ---------------------------------------------------------------------------
// Backward: trace_ray_backward (rt_kernel.cu:331-496) + kernel (:675-694)
// ---------------------------------------------------------------------------

// Specialised backward, arranged for CDNA4's memory-side float atomics
// (MI355X_MICROARCH.md "Global float atomics": a wave-instruction that adds one
// dword per lane into 64 different rows runs ~17x below the rate of one that
// covers contiguous row segments -- 9.8 ms for this kernel written the
// reference's way).
//
//   pass 1  marches the ray once (rt_kernel.cu:365-437 without the atomics):
//           builds `accum` and the final transmittance and, when a workspace is
//           given, records every composited sample as (feature row, delta_t)
//           in a per-ray list rec[k][q] (up to S entries).
//   pass 2  runs wave-synchronously over the samples.  With the list it does
//           not traverse the tree again: it replays the recorded samples (a
//           ray whose list overflowed continues by marching from where the
//           list ends; without a workspace every ray marches, as in the
//           reference's second pass, :439-494).  Every lane with a sample
//           writes its K gradient values -- the colour terms the reference
//           scatters in pass 1 (:410-425), recomputed from the same operands,
//           and the sigma term (:486-490) -- to an LDS staging row; the wave
//           then flushes the staged rows cooperatively: lanes 0..31 and 32..63
//           each take one row per round and issue ONE atomic instruction
//           covering two contiguous K-float segments.  (Rows of one iteration
//           that hit the same leaf are not merged first -- see flush_staged.)
//
// Per-contribution values are bit-identical to the reference formulas; only
// the order in which floats are accumulated differs (as it does between any
// two runs of the reference's own atomics).

// Colour / sigma contributions of one sample -> staging row `st`; advances the
// ray's transmittance and the running `accum` exactly as pass 2 of the
// reference does.
// XF: `basis` is the sample's own (rotated) basis, used for the colour terms;
// `basis_sig` is the one the reference's second pass sees for total_color -- the
// basis its first pass ended with (rt_kernel.cu:439-494 never re-evaluates it).
template <int FMT, int C, int BD, int K, bool XF = false>
__device__ __forceinline__ void stage_sample(const float (&row)[K], const float* basis, const float* g,
                                             float delta_t, float delta_scale, float light_ray,
                                             float& light, float& accum, float* __restrict__ st,
                                             const float* basis_sig = nullptr) {
    const float sigma = row[K - 1];
    const float att = pexpf(-delta_t * sigma * delta_scale);
    const float weight = light * (1.f - att);
    float total_color = 0.f;
    if constexpr (FMT == FMT_SH) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float tmp = 0.f;
#pragma unroll
            for (int i = 0; i < BD; ++i) tmp += basis[i] * row[c * BD + i];
            const double sd = sigmoid_d(tmp);
            const float sig = (float)sd;
            const float gsig = (float)((double)sig * (1.0 - (double)sig));
#pragma unroll
            for (int i = 0; i < BD; ++i) st[c * BD + i] = weight * basis[i] * gsig * g[c];
            if constexpr (XF) {
                float tmp2 = 0.f;
#pragma unroll
                for (int i = 0; i < BD; ++i) tmp2 += basis_sig[i] * row[c * BD + i];
                total_color = (float)((double)total_color + sigmoid_d(tmp2) * (double)g[c]);
            } else {
                total_color = (float)((double)total_color + sd * (double)g[c]);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const double sd = sigmoid_d(row[j]);
            const float sig = (float)sd;
            st[j] = weight * sig * (1.f - sig) * g[j];
            total_color = (float)((double)total_color + sd * (double)g[j]);
        }
    }
    light *= att;
    accum -= weight * total_color;
    st[K - 1] = delta_t * delta_scale * (total_color * light - accum)
              + delta_t * delta_scale * g[C] * light_ray;
}

// Pass-1 bookkeeping of one sample (rt_kernel.cu:397-428 without the atomics).
template <int FMT, int C, int BD, int K>
__device__ __forceinline__ void accum_sample(const float (&row)[K], const float* basis, const float* g,
                                             float delta_t, float delta_scale, float& light, float& accum) {
    const float sigma = row[K - 1];
    const float att = pexpf(-delta_t * sigma * delta_scale);
    const float weight = light * (1.f - att);
    float total_color = 0.f;
    if constexpr (FMT == FMT_SH) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float tmp = 0.f;
#pragma unroll
            for (int i = 0; i < BD; ++i) tmp += basis[i] * row[c * BD + i];
            total_color += (float)sigmoid_d(tmp) * g[c];
        }
    } else {
#pragma unroll
        for (int j = 0; j < C; ++j) total_color += (float)sigmoid_d(row[j]) * g[j];
    }
    light *= att;
    accum += weight * total_color;
}

// ONEPASS (RGBA-style rows, lists): everything of a listed sample that needs its sigmoids, formed
// ONCE -- the colour entries the reference scatters in its first pass (rt_kernel.cu:419-425), that
// pass's total_color (float sigmoids, :424) and the second pass's (double quotients, :470) --
// instead of once per pass: with 31 channels the two passes are 62 double-precision quotients and
// exponentials per sample and bound the kernel (r02: 3.4 ms at 1024 x 1024, depth 9).
template <int C, int K>
__device__ __forceinline__ void stage_colour(const float (&row)[K], const float* g, float delta_t,
                                             float delta_scale, float& light, float& accum,
                                             float* __restrict__ st, float& total2) {
    const float att = pexpf(-delta_t * row[K - 1] * delta_scale);
    const float weight = light * (1.f - att);
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const double sd = sigmoid_d(row[j]);
        const float sig = (float)sd;
        st[j] = weight * sig * (1.f - sig) * g[j];
        t1 += sig * g[j];
        t2 = (float)((double)t2 + sd * (double)g[j]);
    }
    st[K - 1] = 0.f;               // the sigma entry follows in the second sweep, when accum is complete
    light *= att;
    accum += weight * t1;
    total2 = t2;
}

// A sample's contribution in factored form, for the two-kernel backward: the colour
// entry (c, i) is ((weight * basis_i) * coef_c) * g_c with coef_c = sigmoid'(.) for SH,
// or coef_c itself for RGBA; `sg` is the sigma entry.  Same operations as stage_sample.
template <int FMT, int C, int BD, int K, bool XF = false, bool ILP = false>
__device__ __forceinline__ void coef_sample(const float (&row)[K], const float* basis, const float* g,
                                            float delta_t, float delta_scale, float light_ray,
                                            float& light, float& accum, float& weight_out,
                                            float (&coef)[C], float& sg, const float* basis_sig = nullptr) {
    const float sigma = row[K - 1];
    const float att = pexpf<ILP>(-delta_t * sigma * delta_scale);
    const float weight = light * (1.f - att);
    float total_color = 0.f;
    if constexpr (FMT == FMT_SH) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float tmp = 0.f;
#pragma unroll
            for (int i = 0; i < BD; ++i) tmp += basis[i] * row[c * BD + i];
            const double sd = sigmoid_d<ILP>(tmp);
            const float sig = (float)sd;
            coef[c] = (float)((double)sig * (1.0 - (double)sig));
            if constexpr (XF) {      // pass 2 of the reference evaluates total_color with the stale basis
                float tmp2 = 0.f;
#pragma unroll
                for (int i = 0; i < BD; ++i) tmp2 += basis_sig[i] * row[c * BD + i];
                total_color = (float)((double)total_color + sigmoid_d<ILP>(tmp2) * (double)g[c]);
            } else {
                total_color = (float)((double)total_color + sd * (double)g[c]);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const double sd = sigmoid_d<ILP>(row[j]);
            const float sig = (float)sd;
            coef[j] = weight * sig * (1.f - sig) * g[j];
            total_color = (float)((double)total_color + sd * (double)g[j]);
        }
    }
    light *= att;
    accum -= weight * total_color;
    sg = delta_t * delta_scale * (total_color * light - accum)
       + delta_t * delta_scale * g[C] * light_ray;
    weight_out = weight;
}

// coef_sample() in two parts: what depends on the sample alone ...
template <int FMT, int C, int BD, int K>
__device__ __forceinline__ void sample_terms(const float (&row)[K], const float* basis, const float* g,
                                             float delta_t, float delta_scale, float& att, float& total_color,
                                             float (&coef)[C]) {
    att = pexpf<true>(-delta_t * row[K - 1] * delta_scale);
    total_color = 0.f;
    if constexpr (FMT == FMT_SH) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float tmp = 0.f;
#pragma unroll
            for (int i = 0; i < BD; ++i) tmp += basis[i] * row[c * BD + i];
            const double sd = sigmoid_d<true>(tmp);
            const float sig = (float)sd;
            coef[c] = (float)((double)sig * (1.0 - (double)sig));
            total_color = (float)((double)total_color + sd * (double)g[c]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const double sd = sigmoid_d<true>(row[j]);
            coef[j] = (float)sd;                     // sample_advance forms weight * sig * (1 - sig) * g_j
            total_color = (float)((double)total_color + sd * (double)g[j]);
        }
    }
}

// ... and what runs from sample to sample along the ray (RGBA: coef_j <- weight * sig_j * (1 - sig_j) * g_j)
template <int FMT, int C>
__device__ __forceinline__ void sample_advance(float att, float total_color, float (&coef)[C], const float* g,
                                               float delta_t, float delta_scale, float light_ray,
                                               float& light, float& accum, float& weight_out, float& sg) {
    const float weight = light * (1.f - att);
    if constexpr (FMT == FMT_RGBA) {
#pragma unroll
        for (int j = 0; j < C; ++j) coef[j] = weight * coef[j] * (1.f - coef[j]) * g[j];
    }
    light *= att;
    accum -= weight * total_color;
    sg = delta_t * delta_scale * (total_color * light - accum)
       + delta_t * delta_scale * g[C] * light_ray;
    weight_out = weight;
}

constexpr uintptr_t kOnlyOverflowed = 1;      // render_bwd_kernel<..., XF> (lists, no GATHER): coef_out = this value, see there
// REPLAY: rec / aux were filled by render_fwd_kernel<..., REC=true> for the
// same tree, rays and options: pass 1 walks the list instead of the tree.
// GATHER (C == 3, lists only): the listed samples are not sent to the gradient table
// here; their factored contributions overwrite the list -- rec[k][q] = (row, sigma
// entry), coef[k][q] = (weight, c0, c1, c2) -- and grad_merge_kernel adds them up per
// 8x8 tile.  Samples past the list (overflowed rays) still go out as shaped atomics.
// ONEPASS (RGBA-style rows, lists, L.terms = one float per list slot): see stage_colour.  Sweep 1
// walks the lists wave-synchronously and sends the colour entries out as shaped atomic rows; it
// leaves the second pass's total_color of every listed sample in L.terms (position-major, 256
// contiguous bytes per wavefront and list position).  Sweep 2 is scalar work per listed sample --
// sigma gather, attenuation, accum -= weight * total_color -- and one atomic on the sigma column.
// Per-contribution values are the reference's; only where the additions happen differs.
// LOBES (FMT_SH instances): the basis values are those of opt.format = SG or ASG with BD lobes (precalc_lobes).
template <int FMT, int C, int BD, bool N2, bool REPLAY, bool XF = false, bool GATHER = false, bool ONEPASS = false,
          bool LOBES = false>
__global__ void __launch_bounds__(kBlock, (GATHER && !XF && C == 3 && BD <= 9) ? 4 : 1)
render_bwd_kernel(TreeDev tr, RaysDev rays, Opts opt, const float* __restrict__ grad_out,
                  float* __restrict__ grad, int gstride, RecLists L,
                  const uint4* __restrict__ aux, const float* __restrict__ fwd_out,
                  float4* __restrict__ coef_out = nullptr) {
    // (C > 3: only as the tail-only launch, coef_out == NULL, in front of grad_wide_kernel)
    static_assert(!GATHER || (REPLAY && (C == 3 || FMT == FMT_RGBA)), "per-tile backward: lists; 3 channels or RGBA-style rows");
    static_assert(!ONEPASS || (REPLAY && FMT == FMT_RGBA && !XF && !GATHER), "one sigmoid pass: RGBA-style rows, lists");
    constexpr int K = (FMT == FMT_RGBA) ? (C + 1) : (C * BD + 1);
    constexpr int KS = K | 1;                         // odd LDS row stride: conflict-free column writes
    __shared__ float stage_all[(kBlock / 64) * 64 * KS];
    __shared__ int32_t sidx_all[kBlock];
    __shared__ uint2 rstage[REPLAY ? 1 : kRecBlock * kBlock];     // pass 1 records into the workspace lists
    __shared__ int32_t ltab[REPLAY ? 1 : kMaxRecBlocks];
    if constexpr (!REPLAY) rec_tab_init(ltab);
    uint2* __restrict__ rec = L.rec;
    const int S = L.S;
    int64_t cur_block = 0;

    if constexpr (GATHER) {
        // tail-only launch (coef_out == NULL): nothing to do when no ray of the batch overflowed (one scalar load)
        if (coef_out == nullptr && no_ray_overflowed(L)) return;
    }
    // (r04) XF, lists, no GATHER: coef_out == kOnlyOverflowed restricts the launch to the rays whose list overflowed --
    // grad_fused_kernel<..., XF> leaves exactly those to it (their last sample, whose basis the second pass keeps, lies
    // past the list)
    const bool only_overflowed = XF && REPLAY && !GATHER && reinterpret_cast<uintptr_t>(coef_out) == kOnlyOverflowed;
    if (only_overflowed && no_ray_overflowed(L)) return;
    const int lane = threadIdx.x & 63;
    float* stage = stage_all + (threadIdx.x >> 6) * (64 * KS);
    int32_t* sidx = sidx_all + (threadIdx.x >> 6) * 64;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t q = ray_of_thread(rays, tid);
    Ray r;
    bool alive = q < rays.Q;
    if (only_overflowed) alive = alive && (aux[alive ? q : 0].x & kRecOverflow) != 0u;
    if constexpr (GATHER) {
        // tail-only launch (coef_out == NULL): most wavefronts have no overflowed ray and leave here
        if (coef_out == nullptr && !__any(alive && (aux[alive ? q : 0].x & kRecOverflow) != 0u)) return;
    }
    if (alive) alive = setup_ray(tr, rays, opt, q, r);
    if (!__any(alive)) return;

    static_assert(!XF || FMT == FMT_SH, "view rotations only matter for view-dependent formats");
    float basis[BD > 0 ? BD : 1];          // the unrotated basis; with XF: scratch for the current sample's
    float basis_last[XF ? BD : 1];         // XF: the basis pass 1 ends with, which pass 2's total_color uses
    float vd[3] = {0.f, 0.f, 0.f};
    float g[C + 1];
    if (alive) {
        if constexpr (FMT == FMT_SH) {
            load_vdir(rays, q, vd);
            if constexpr (LOBES) precalc_lobes<BD>(opt.format, tr, vd[0], vd[1], vd[2], basis);
            else precalc_basis<BD>(FMT_SH, BD, tr, vd[0], vd[1], vd[2], basis);
        }
#pragma unroll
        for (int j = 0; j <= C; ++j) g[j] = grad_out[q * (C + 1) + j];
    }
    if constexpr (XF) {
#pragma unroll
        for (int i = 0; i < BD; ++i) basis_last[i] = basis[i];     // a ray without samples never evaluates another
    }

    float accum = 0.f;
    float light_ray = 1.f;
    float light1 = 1.f;           // ONEPASS: the transmittance behind the listed samples
    float* __restrict__ tot2 = reinterpret_cast<float*>(L.terms);
    // a record's slot in the position-major hand-over, from its place in rec[block][lane][k % 8]
    auto pm_of = [](int64_t ri) { return ((ri >> 9) << 9) + ((ri & 7) << 6) + ((ri >> 3) & 63); };
    if constexpr (ONEPASS) {      // sweep 1 over the lists, all lanes of the wavefront in step
        int n1 = 0;
        if (alive) n1 = (int)(aux[q].x & ~kRecOverflow);
        int maxn = n1;
        for (int off = 32; off > 0; off >>= 1) maxn = max(maxn, __shfl_xor(maxn, off, 64));
        maxn = __builtin_amdgcn_readfirstlane(maxn);
        for (int k1 = 0; k1 < maxn; ++k1) {
            const bool active = k1 < n1;
            const unsigned long long amask = __ballot(active);
            if (active) {
                const int64_t ri = rec_index(L, tid, k1);
                const uint2 e = rec_get(rec + ri);
                float row[K];
                load_row<K>(tr.features + (int64_t)(int32_t)e.x * K, row);
                const int slot = __popcll(amask & lane_lt);
                sidx[slot] = (int32_t)e.x;
                float t2;
                stage_colour<C, K>(row, g, __uint_as_float(e.y), r.delta_scale, light1, accum, stage + slot * KS, t2);
                tot2[pm_of(ri)] = t2;
            }
            flush_staged<K, KS>(stage, sidx, __popcll(amask), lane, grad, gstride);
        }
    }
    int nrec = 0;                 // samples recorded for this ray
    float t_resume = 0.f;         // where pass 2 resumes marching
    float tmax2 = -1.f;           // ... and until where (-1: nothing left to march)
    if (alive) {   // pass 1
        bool skip_pass1 = false;
        int32_t last_idx = -1;    // XF: feature row of the ray's last composited sample
        float light = 1.f, t = r.tmin;
        t_resume = r.tmin;
        tmax2 = (S > 0) ? -1.f : r.tmax;
        if constexpr (REPLAY) {
            const uint4 a = aux[q];
            nrec = (int)(a.x & ~kRecOverflow);
            if (a.x & kRecOverflow) { t_resume = __uint_as_float(a.y); tmax2 = r.tmax; }
            if constexpr (GATHER) {
                // tail-only launch (coef_out == NULL, in front of grad_fused_kernel): nothing to
                // do for a ray whose list holds all of its samples
                if (coef_out == nullptr && tmax2 < 0.f) nrec = 0;
            }
            if (fwd_out != nullptr) {
                // Single march: what pass 1 would compute is already in the forward's
                // output.  accum = sum_j w_j sum_c s_jc g_c + T bg sum_c g_c equals
                // sum_c g_c out_c (out_c = sum_j w_j s_jc + T bg, thresholds are 0), and
                // the final transmittance was recorded.  Differs from the two-pass
                // value by float rounding only (~1e-7 of the summed magnitudes); it
                // enters the sigma terms alone, the colour terms stay bit-identical.
                const float* o = fwd_out + q * (C + 1);
#pragma unroll
                for (int c = 0; c < C; ++c) accum += g[c] * o[c];
                light_ray = __uint_as_float(a.z);
                skip_pass1 = true;
                if constexpr (XF) {
                    if (nrec > 0) last_idx = (int32_t)rec_get(rec + rec_index(L, tid, nrec - 1)).x;
                }
            } else if constexpr (ONEPASS) {
                light = light1;                  // the lists were walked above; a tail marches on from here
            } else {
                for (int k = 0; k < nrec; ++k) {
                    const uint2 e = rec_get(rec + rec_index(L, tid, k));
                    float row[K];
                    load_row<K>(tr.features + (int64_t)(int32_t)e.x * K, row);
                    if constexpr (XF) { rotated_sh_basis<BD>(tr, (int32_t)e.x, vd, basis); last_idx = (int32_t)e.x; }
                    accum_sample<FMT, C, BD, K>(row, basis, g, __uint_as_float(e.y), r.delta_scale, light, accum);
                }
            }
            // march only what the list does not cover (with the forward's output at hand
            // that is nothing -- unless XF still has to find the last sample past the list)
            const bool tail = tmax2 >= 0.f && (!skip_pass1 || XF);
            t = tail ? t_resume : r.tmax;
        }
        while (t < r.tmax) {
            Sample s;
            march_step<N2>(tr, r, opt.step_size, t, s);
            if (s.valid) {
                float row[K];
                load_row<K>(tr.features + (int64_t)s.idx * K, row);   // whole row: sigma is its last element
                if (row[K - 1] > 0.f) {
                    if constexpr (!REPLAY) {
                        if (S > 0) {
                            bool room = nrec < S;
                            if (room && (nrec & 7) == 0) {
                                cur_block = rec_block_begin(L, ltab, tid >> 6, nrec >> 3);
                                room = cur_block >= 0;
                            }
                            if (room) {
                                rec_stage_put(rstage, (int)threadIdx.x, rec, cur_block, nrec, (uint32_t)s.idx, s.delta_t);
                                ++nrec;
                            } else if (tmax2 < 0.f) {   // list full: pass 2 marches from this step on
                                t_resume = t;
                                tmax2 = r.tmax;
                            }
                        }
                    }
                    if constexpr (XF) last_idx = s.idx;
                    if (!skip_pass1) {
                        if constexpr (XF) rotated_sh_basis<BD>(tr, s.idx, vd, basis);
                        accum_sample<FMT, C, BD, K>(row, basis, g, s.delta_t, r.delta_scale, light, accum);
                    }
                }
            }
            t = march_advance(t, s.delta_t);
        }
        if (!skip_pass1) {
            float total_grad = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) total_grad += g[j];
            accum += light * opt.background_brightness * total_grad;
            light_ray = light;
        }
        if constexpr (XF) {
            if (last_idx >= 0) rotated_sh_basis<BD>(tr, last_idx, vd, basis_last);
        }
        if constexpr (!REPLAY) {
            if (S > 0) rec_stage_finish(rstage, (int)threadIdx.x, rec, cur_block, nrec);
        }
    }

    if constexpr (GATHER) {
        // tail-only launch in front of grad_fused_kernel<..., EXACT>: an overflowed ray's pass 1
        // (list + tail) is complete only here; its results travel in aux[q].z / .w
        if (coef_out == nullptr && fwd_out == nullptr && alive && tmax2 >= 0.f) {
            uint4* aw = const_cast<uint4*>(aux) + q;
            aw->z = __float_as_uint(light_ray);
            aw->w = __float_as_uint(accum);
        }
    }
    // pass 2, wave-synchronous: replay the recorded samples, then (rays with
    // unrecorded samples only) march the rest.
    float light = 1.f;
    int k = 0;
    float t = t_resume;
    if constexpr (GATHER) {
        if (coef_out == nullptr && !__any(tmax2 >= 0.f)) return;       // tail-only launch, no overflowed ray here
        // lane-independent: no wavefront-wide synchronisation while walking the list
        if (alive) {
            for (; k < nrec; ++k) {
                uint2* slot = rec + rec_index(L, tid, k);
                const uint2 e = rec_get(slot);
                float row[K];
                load_row<K>(tr.features + (int64_t)(int32_t)e.x * K, row);
                float w, cf[C], sg;
                typedef float v4f __attribute__((ext_vector_type(4)));
                if constexpr (XF) {
                    // the sample's own (rotated) direction goes along: coef[S + k][q]
                    float rd[3];
                    rotated_dir(tr, (int32_t)e.x, vd, rd);
                    precalc_basis<BD>(FMT_SH, BD, tr, rd[0], rd[1], rd[2], basis);
                    coef_sample<FMT, C, BD, K, true>(row, basis, g, __uint_as_float(e.y), r.delta_scale, light_ray,
                                                     light, accum, w, cf, sg, basis_last);
                    if (coef_out != nullptr)
                        __builtin_nontemporal_store(v4f{rd[0], rd[1], rd[2], 0.f},
                                                    reinterpret_cast<v4f*>(coef_out + ((int64_t)(S + k) * rays.Q + q)));
                } else {
                    coef_sample<FMT, C, BD, K, false, true>(row, basis, g, __uint_as_float(e.y), r.delta_scale,
                                                            light_ray, light, accum, w, cf, sg);
                }
                if (coef_out != nullptr) {               // (tail-only launch: the running values alone)
                    rec_put(slot, e.x, sg);
                    __builtin_nontemporal_store(v4f{w, cf[0], cf[1], cf[2]},
                                                reinterpret_cast<v4f*>(coef_out + ((int64_t)k * rays.Q + q)));
                }
            }
        }
        k = nrec;
    }
    if constexpr (ONEPASS) {
        // sweep 2 over the lists: the sigma entries (rt_kernel.cu:456-490), no sigmoid formed again
        if (alive) {
            for (; k < nrec; ++k) {
                const int64_t ri = rec_index(L, tid, k);
                const uint2 e = rec_get(rec + ri);
                const int32_t idx = (int32_t)e.x;
                const float delta_t = __uint_as_float(e.y);
                const float sigma = tr.features[(int64_t)idx * K + (K - 1)];
                const float att = pexpf(-delta_t * sigma * r.delta_scale);
                const float weight = light * (1.f - att);
                const float total_color = tot2[pm_of(ri)];
                light *= att;
                accum -= weight * total_color;
                const float sg = delta_t * r.delta_scale * (total_color * light - accum)
                               + delta_t * r.delta_scale * g[C] * light_ray;
                atomicAdd(grad + (int64_t)idx * gstride + (K - 1), sg);
            }
        }
        k = nrec;
    }
    while (__any(k < nrec || t < tmax2)) {
        // which lanes have a sample this iteration, and which feature row it is
        bool active = false;
        int32_t idx = -1;
        float delta_t = 0.f;
        float row[K];
        if (k < nrec) {
            const uint2 e = rec_get(rec + rec_index(L, tid, k));
            ++k;
            idx = (int32_t)e.x;
            delta_t = __uint_as_float(e.y);
            load_row<K>(tr.features + (int64_t)idx * K, row);
            active = true;
        } else if (t < tmax2) {
            Sample s;
            march_step<N2>(tr, r, opt.step_size, t, s);
            delta_t = s.delta_t;
            t = march_advance(t, s.delta_t);
            if (s.valid) {
                load_row<K>(tr.features + (int64_t)s.idx * K, row);
                if (row[K - 1] > 0.f) { active = true; idx = s.idx; }
            }
        }
        const unsigned long long amask = __ballot(active);
        if (amask == 0ull) continue;
        if (active) {
            const int slot = __popcll(amask & lane_lt);       // compact: staging row = rank among active lanes
            sidx[slot] = idx;
            if constexpr (XF) {
                rotated_sh_basis<BD>(tr, idx, vd, basis);
                stage_sample<FMT, C, BD, K, true>(row, basis, g, delta_t, r.delta_scale, light_ray,
                                                  light, accum, stage + slot * KS, basis_last);
            } else {
                stage_sample<FMT, C, BD, K>(row, basis, g, delta_t, r.delta_scale, light_ray,
                                            light, accum, stage + slot * KS);
            }
        }
        flush_staged<K, KS>(stage, sidx, __popcll(amask), lane, grad, gstride);
    }
}

// Second kernel of the two-kernel backward.  One wavefront per 64 rays (the same
// ray <-> lane map as the march: an 8x8 tile).  Neighbouring rays hit the same leaves
// (5.9x on the headline workload), so the tile's records are grouped by feature row
// and a row leaves the CU once per tile instead of once per sample:
//   load     lane = ray: the tile's records go to LDS; a hash table (atomicCAS on the
//            key) maps each feature row to a slot, a counter per slot counts its records
//   sort     exclusive scan of the counters, scatter of the record numbers (counting sort)
//   reduce   64 sorted records at a time.  lane = record: expand it into its K gradient
//            values -- from the factored form ((weight * basis_i) * coef_c) * g_c, the
//            reference's own order of operations, with the rays' bases and upstream
//            gradients in LDS -- and stage them; then lane = gradient column: each
//            half-wavefront adds up 32 staged rows in order and, when the feature row
//            changes, sends the sum out: one atomic instruction per row, K contiguous floats.
// LDS float atomics are not used: ds_add_f32 retires about one lane per 4-5 clocks
// (measured: 1.2 ms for this kernel written with a table of ds_add_f32 rows).  Also
// measured: a row per lane summed in registers, 64 rows at a time (most lanes idle behind
// the longest row, 0.55 ms); column lanes reading record fields straight from LDS
// (a dependent read chain per record, 1.16 ms).
// XF: the basis is per record (view rotations): evaluated in the expand step from the
// rotated direction the list walk stored in coef[S + k][q].
template <int FMT, int BD, int T, int R, int W, bool XF = false>
__global__ void __launch_bounds__(64 * W)
grad_merge_kernel(TreeDev tr, RaysDev rays, const float* __restrict__ grad_out, RecLists L,
                  const float4* __restrict__ coef, const uint4* __restrict__ aux,
                  float* __restrict__ grad, int gstride) {
    if (tile_never_recorded(L, blockIdx.x)) return;          // (r03: see grad_fused_kernel)
    const int S = L.S;
    const int32_t tabreg = rec_tab_reg(L, blockIdx.x, threadIdx.x & 63);
    // W wavefronts share one tile (64 rays) and its LDS: the phases below are latency
    // chains of LDS operations, and LDS -- not registers -- limits how many tiles a CU
    // holds, so the way to more wavefronts per CU is more wavefronts per tile.  Lane l of
    // every wavefront stands for ray l of the tile; list positions (load) and batches of
    // 64 sorted records (reduce) are dealt round-robin to the wavefronts.
    constexpr int C = 3;
    constexpr int K = (FMT == FMT_RGBA) ? (C + 1) : (C * BD + 1);
    constexpr int BDS = (FMT == FMT_SH) ? (BD | 1) : 1;      // odd stride: conflict-free basis rows
    constexpr int KS = K | 1;
    constexpr int NT = 64 * W;
    static_assert(K <= 32 && (T & (T - 1)) == 0 && T >= 128 && T <= 1024 && R >= 128 && R <= 1024 && T * 64 <= 65536, "sizes");
    __shared__ int32_t keys[T];
    __shared__ int32_t cnt[T];                   // records per slot; after the scan: where the slot's next record goes
    __shared__ uint16_t order[R];
    __shared__ uint16_t r_sl[R];                 // slot << 6 | lane
    __shared__ float r_sg[R], r_w[R], r_c[3 * R];
    __shared__ float r_d[XF ? 3 * R : 1];        // XF: the record's rotated view direction
    __shared__ float bases[XF ? 1 : 64 * BDS];
    __shared__ float gl[64 * 3];
    __shared__ float stage_all[W * 64 * KS];
    __shared__ int32_t seg_all[W * 64];
    __shared__ int32_t s_nb;                     // records in the buffer
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* stage = stage_all + wave * 64 * KS;
    int32_t* seg = seg_all + wave * 64;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const int64_t q = ray_of_thread(rays, (int64_t)blockIdx.x * 64 + lane);
    const bool alive = q < rays.Q;
    int nrec = 0;
    if (alive) nrec = (int)(aux[q].x & ~kRecOverflow);
    int maxn = nrec;
    for (int off = 32; off > 0; off >>= 1) maxn = max(maxn, __shfl_xor(maxn, off, 64));
    // wavefront-uniform by construction; said so to the compiler, because everything that
    // decides how many workgroup barriers a wavefront executes must be scalar control flow
    maxn = __builtin_amdgcn_readfirstlane(maxn);
    if (maxn == 0) return;                       // the same in every wavefront of the workgroup
    if (wave == 0 && alive && nrec > 0) {
        if constexpr (FMT == FMT_SH && !XF) {
            float vd[3], b[BD];
            load_vdir(rays, q, vd);
            precalc_basis<BD>(FMT_SH, BD, tr, vd[0], vd[1], vd[2], b);
#pragma unroll
            for (int i = 0; i < BD; ++i) bases[lane * BDS + i] = b[i];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) gl[lane * 3 + c] = grad_out[q * (C + 1) + c];
    }
    for (int i = threadIdx.x; i < T; i += NT) { keys[i] = -1; cnt[i] = 0; }
    if (threadIdx.x == 0) s_nb = 0;
    __syncthreads();

    // List positions are taken kGroup at a time per wavefront (their loads overlap), RPP
    // rounds of W * kGroup positions per pass: the schedule is fixed by maxn alone -- a
    // pass holds at most RPP * W * kGroup * 64 <= R records and as many distinct rows <= T,
    // so neither the buffer nor the table (which may run full: every key searched for is
    // then present) needs a data-dependent check between workgroup barriers.
    constexpr int kGroup = W >= 8 ? 1 : 2;
    constexpr int kRound = kGroup * W;                       // list positions per round
    constexpr int RPP = R / (64 * kRound);                   // rounds per pass
    static_assert(RPP >= 1 && RPP * 64 * kRound <= T, "a pass must fit the buffer and the table");
    typedef float v4f __attribute__((ext_vector_type(4)));
    // the records of a round are requested one round ahead (across the sort / reduce of the
    // pass in between as well), so their memory latency is not waited for
    uint2 e_n[kGroup];
    v4f c_n[kGroup], d_n[kGroup];
    auto request = [&](int kb) {
#pragma unroll
        for (int u = 0; u < kGroup; ++u) {
            e_n[u] = make_uint2(0u, 0u);
            c_n[u] = v4f{0.f, 0.f, 0.f, 0.f};
            d_n[u] = v4f{0.f, 0.f, 0.f, 0.f};
            if (kb + u < nrec) {
                e_n[u] = rec_get(L.rec + rec_index_in(rec_block_u(L, tabreg, blockIdx.x, (kb + u) >> 3), lane, kb + u));
                c_n[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(coef + ((int64_t)(kb + u) * rays.Q + q)));
                if constexpr (XF)
                    d_n[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(coef + ((int64_t)(S + kb + u) * rays.Q + q)));
            }
        }
    };
    request(wave * kGroup);
    for (int k0 = 0; k0 < maxn; k0 += RPP * kRound) {
        // ---- load: fill the record buffer and the hash table
#pragma unroll 1
        for (int rd = 0; rd < RPP; ++rd) {
            const int kb = k0 + rd * kRound + wave * kGroup;   // this wavefront's positions of the round
            if (kb >= maxn) break;                             // wavefront-uniform; no barrier inside this loop
            uint2 e[kGroup];
            v4f c4[kGroup], d4[kGroup];
#pragma unroll
            for (int u = 0; u < kGroup; ++u) { e[u] = e_n[u]; c4[u] = c_n[u]; d4[u] = d_n[u]; }
            if (kb + kRound < maxn) request(kb + kRound);      // this wavefront's positions of the next round
#pragma unroll
            for (int u = 0; u < kGroup; ++u) {
                const bool active = kb + u < nrec;
                const unsigned long long am = __ballot(active);
                if (am == 0ull) continue;
                uint32_t h = 0;
                if (active) {
                    const int32_t idx = (int32_t)e[u].x;
                    h = ((uint32_t)idx * 0x9E3779B1u) >> (32 - __builtin_ctz(T));
                    while (true) {
                        const int32_t old = atomicCAS(keys + h, -1, idx);
                        if (old == -1 || old == idx) break;
                        h = (h + 1u) & (uint32_t)(T - 1);
                    }
                    atomicAdd(cnt + h, 1);
                }
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_nb, __popcll(am));
                base = __shfl(base, 0, 64);
                if (active) {
                    const int pos = base + __popcll(am & lane_lt);
                    r_sl[pos] = (uint16_t)((h << 6) | (uint32_t)lane);
                    r_sg[pos] = __uint_as_float(e[u].y);
                    r_w[pos] = c4[u].x; r_c[pos] = c4[u].y; r_c[R + pos] = c4[u].z; r_c[2 * R + pos] = c4[u].w;
                    if constexpr (XF) { r_d[pos] = d4[u].x; r_d[R + pos] = d4[u].y; r_d[2 * R + pos] = d4[u].z; }
                }
            }
        }
        __syncthreads();
        // ---- sort: exclusive scan of the counters (wavefront 0), counting sort of the record numbers
        const int nb = __builtin_amdgcn_readfirstlane(s_nb);
        if (wave == 0) {
            constexpr int PER = T / 64;
            int mine[PER], sum = 0;
#pragma unroll
            for (int j2 = 0; j2 < PER; ++j2) { mine[j2] = cnt[lane * PER + j2]; sum += mine[j2]; }
            int incl = sum;
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off, 64);
                if (lane >= off) incl += v;
            }
            int run = incl - sum;
#pragma unroll
            for (int j2 = 0; j2 < PER; ++j2) { cnt[lane * PER + j2] = run; run += mine[j2]; }
        }
        __syncthreads();
        for (int rr = threadIdx.x; rr < nb; rr += NT) {
            const int sl = (int)r_sl[rr] >> 6;
            const int pos = atomicAdd(cnt + sl, 1);
            order[pos] = (uint16_t)rr;
        }
        __syncthreads();
        // ---- reduce: 64 sorted records at a time per wavefront.  lane = record expands it into
        // its K gradient values (staged in LDS); then four groups of 16 lanes each walk 16 staged
        // rows, a lane owning two columns (sub and sub + 16), and send a sum out whenever the
        // feature row changes: a row leaves as two atomic instructions (columns 0-15, 16-K).
        for (int base = wave * 64; base < nb; base += NT) {
            const int p = base + lane;
            int my_sl = -1;
            if (p < nb) {
                const int rr = (int)order[p];
                const int v = (int)r_sl[rr];
                my_sl = v >> 6;
                const int rl = v & 63;
                float* st = stage + lane * KS;
                if constexpr (FMT == FMT_SH) {
                    const float w = r_w[rr];
                    float bx[XF ? BD : 1];
                    const float* b = bases + rl * BDS;
                    if constexpr (XF) {
                        precalc_basis<BD>(FMT_SH, BD, tr, r_d[rr], r_d[R + rr], r_d[2 * R + rr], bx);
                        b = bx;
                    }
#pragma unroll
                    for (int c3 = 0; c3 < C; ++c3) {
                        const float cc = r_c[c3 * R + rr], gc = gl[rl * 3 + c3];
#pragma unroll
                        for (int i = 0; i < BD; ++i) st[c3 * BD + i] = w * b[i] * cc * gc;
                    }
                } else {
                    st[0] = r_c[rr]; st[1] = r_c[R + rr]; st[2] = r_c[2 * R + rr];
                }
                st[K - 1] = r_sg[rr];
            }
            seg[lane] = my_sl >= 0 ? keys[my_sl] : -1;          // the feature row staged row `lane` belongs to
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // columns 0-15 and 16-K: with rows that start on 128-byte boundaries (grad_stride 32
            // for K = 28) each of the two atomic instructions of a row touches one 64-byte line --
            // the kernel runs at the rate at which the memory side takes such requests
            constexpr int HALF = K < 16 ? K : 16;
            const int grp = lane >> 4, sub = lane & 15;
            const bool has0 = sub < HALF, has1 = sub + HALF < K;
            int cur = -1;
            float acc0 = 0.f, acc1 = 0.f;
            // all LDS reads of the 16 steps first (independent), then the sequential logic
            int keyv[16];
            float x0v[16], x1v[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int row = grp * 16 + t;
                keyv[t] = seg[row];
                x0v[t] = stage[row * KS + (has0 ? sub : 0)];
                x1v[t] = stage[row * KS + (has1 ? sub + HALF : 0)];
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int key = keyv[t];
                if (key != cur) {
                    if (cur >= 0) {
                        if (has0) atomicAdd(grad + (int64_t)cur * gstride + sub, acc0);
                        if (has1) atomicAdd(grad + (int64_t)cur * gstride + sub + HALF, acc1);
                    }
                    acc0 = 0.f; acc1 = 0.f;
                    cur = key;
                }
                acc0 += x0v[t];
                acc1 += x1v[t];
            }
            if (cur >= 0) {
                if (has0) atomicAdd(grad + (int64_t)cur * gstride + sub, acc0);
                if (has1) atomicAdd(grad + (int64_t)cur * gstride + sub + HALF, acc1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (k0 + RPP * kRound >= maxn) break;            // last pass (scalar condition)
        __syncthreads();
        for (int i = threadIdx.x; i < T; i += NT) { keys[i] = -1; cnt[i] = 0; }
        if (threadIdx.x == 0) s_nb = 0;
        __syncthreads();
    }
}

// The backward of an image in ONE kernel after the forward (3-channel payloads, K <= 32, N = 2,
// no view rotations, lists and the forward's output at hand): list walk and per-tile merge
// fused.  A workgroup of W = 8 wavefronts owns a tile (lane l of every wavefront = ray l).  Per
// round, wavefront w forms the per-sample terms (sample_terms: row gather, exponential, three
// double-precision sigmoids) of list position kb + w for the tile's 64 rays -- eight positions
// of a list at once, where a list walked by one lane costs 1.7 us per position -- and puts
// them, with the hash-table entry of their feature row, straight into the merge's record
// arrays in LDS; wavefront 0 then advances the two values that run along each ray
// (sample_advance: transmittance and accum, a dozen instructions per sample) through the
// round's positions in order, which turns (att, total_color) into (weight, sigma entry).  Two
// rounds fill a pass (16 positions, at most 1024 records, as grad_merge_kernel), which is
// sorted by feature row and reduced as there.  No record goes through memory (rec keeps
// (row, delta_t), coef is not used) and there is no separate list-walk kernel with its tail.
// Operation for operation coef_sample(); rays whose list overflowed have their tail handled
// by a tail-only launch of render_bwd_kernel<..., GATHER> in front.
// EXACT (fwd_out == NULL): nothing is taken from the forward's output.  A first sweep over the
// lists -- the same division of labour: eight wavefronts form (att, total_color) of eight list
// positions, wavefront 0 runs along the rays -- builds `accum` and the final transmittance exactly
// as the reference's first pass does (rt_kernel.cu:365-437: total_color summed in float there, in
// double in the second pass, :472-476 -- both are reproduced), so every gradient contribution is
// bit-identical to the reference's formulas.  Without it (fwd_out given) accum = sum_c g_c out_c:
// one sweep, but a third of the sigma entries then differ from the reference's by more than 1e-5
// of their own value (r02, tests/test_gpu_query_and_misc.py) -- opt-in.
// COUNT (instrumentation, svoxt_set_bwd_counters): counters[0] += 64-byte atomic requests sent,
// counters[1] += (tile, pass, feature row) groups; the work itself is unchanged.
// TERMS (EXACT only): 2 / 3 = the recording forward left (att, e_0, e_1, e_2) of every sample in L.terms
// (render_fwd_kernel: lane-major lines; shade_tile_kernel: position-major): neither sweep gathers a feature
// row or forms an exponential; 0 = both sweeps gather the rows.  (1, a hand-over from sweep 1 to sweep 2 for
// lists whose forward left none, existed until r04: no caller of the operator layer reached it.)
// Two workgroups per CU (77 KB of LDS each) need at most 128 registers: said to the compiler,
// because one branch too many costs exactly that (r02: 116 -> 130 registers, 0.38 -> 0.55 ms).
// LOBES (FMT_SH instances over the forward's hand-over): the 64 rays' basis values are those of opt.format = SG or ASG.
// CHECK (instrumentation, svoxt_set_bwd_check; r04): every LDS / pool / table index against its extent, violations
// counted per site in counters[kChkBase + site] (see chk()); the sites of this kernel: 0 sweep 1's hand-over slot,
// 1 the chain's read of it, 2 a list block outside the pool, 3 a record's slot in a pass, 4 its table entry, 5 the
// chain's slot, 6 a feature row outside the table; svoxt_tile_reduce.inc: 8 table entry of a record, 9 its place in
// the sorted order, 10 sorted record number, 11 table entry of a sorted record, 12 the row a sum is sent to,
// 13 records of the pass != sum of the table's counters.
// XF (r04; SH up to 9 basis functions, EXACT, TERMS 0): per-leaf view rotations (rt_kernel.cu:387-395).  A record's basis is
// its own -- the leaf's matrix applied to the ray's view direction -- in sweep 1 and for the colour entries; sweep 2's
// total_color takes the basis sweep 1 ENDED with (the reference's second pass never re-evaluates it: rt_kernel.cu:439-494),
// which the wavefront that forms a ray's last record leaves in `bases`; the reduce evaluates a record's basis again from
// the row's matrix and the ray's direction (vds) instead of holding nine more floats per record in LDS.  Rays whose list
// overflowed are not this kernel's: their last sample lies past the list, so a launch of render_bwd_kernel<..., XF> over
// those rays alone (kOnlyOverflowed) does all of their work in front.
template <int FMT, int BD, bool EXACT, bool COUNT = false, int TERMS = 0, bool LOBES = false, bool CHECK = false, bool XF = false>
__global__ void __launch_bounds__(512, 4)
grad_fused_kernel(TreeDev tr, RaysDev rays, Opts opt, const float* __restrict__ grad_out,
                  RecLists L, const uint4* __restrict__ aux, const float* __restrict__ fwd_out,
                  float* __restrict__ grad, int gstride, unsigned long long* __restrict__ counters = nullptr) {
    if (tile_never_recorded(L, blockIdx.x)) return;          // (r03: before anything else is requested or cleared)
#define SVOXT_CHK(i, n, site) chk<CHECK>(i, n, counters, site)
    float4* __restrict__ terms = L.terms;
    // P list positions per round, formed by wavefronts W - P .. W - 1 (pw = the wavefront's position in the round);
    // wavefront 0 only runs along the rays.  (r03: with P = W wavefront 0 also formed the terms of a position
    // and was every round's critical path -- its own terms, then the barrier, then the advance of the
    // round while the others already waited at the next barrier: 5 100 cycles per round of sweep 2.)
    // (r03) Rows wider than 32 floats -- SH16: 49, SH25: 76 -- take this kernel too when the forward left the
    // hand-over (TERMS 2 / 3): then neither sweep touches a feature row, the row's width only shows in the reduce
    // (ceil(K / 16) rounds of 16 columns) and in the 64 rays' basis values in LDS (SH25: a pass holds 896 records
    // -- one pair of rounds, the least a pass can be -- instead of 1024 so that two workgroups still share a CU's LDS).  800 x 800 / depth 8, forward+backward:
    // SH16 2.02 -> see profiles/r03_lobes_timing.txt (per ray: 49 float atomics per sample).
    constexpr int C = 3, W = 8, P = W - 1, NT = 64 * W, T = 1024, R = (FMT == FMT_SH && BD > 16) ? 896 : 1024;
    static_assert(2 * P * 64 <= R && P >= 1 && P <= W && R <= T, "a pair of rounds must fit the record arrays, a pass the hash table");
    constexpr int K = (FMT == FMT_RGBA) ? (C + 1) : (C * BD + 1);
    constexpr int NB = (FMT == FMT_SH) ? BD : 0;
    constexpr int BDS = (FMT == FMT_SH) ? (BD | 1) : 1;
    constexpr int HALF = K < 16 ? K : 16;                    // columns 0-15 / 16-31 / ...: see grad_merge_kernel
    constexpr int NH = (K + HALF - 1) / HALF;                // rounds of columns in the reduce
    constexpr int KS = HALF | 1;                             // staging row: one round of columns
    static_assert(TERMS == 0 || TERMS == 2 || TERMS == 3, "hand-over layouts");
    static_assert(K <= 32 || (EXACT && TERMS >= 2), "wide rows: only with the forward's hand-over (no row in registers)");
    static_assert(!LOBES || (FMT == FMT_SH && EXACT && TERMS >= 2), "lobes: only over the forward's hand-over");
    static_assert(!XF || (FMT == FMT_SH && EXACT && (TERMS == 0 || TERMS == 3) && !LOBES && BD <= 9), "view rotations: SH rows in registers, exact");
    __shared__ float vds[XF ? 64 * 3 : 1];       // XF: the 64 rays' view directions, for the reduce
    __shared__ int32_t keys[T];
    __shared__ int32_t cnt[T];
    __shared__ uint16_t order[R];
    __shared__ uint32_t r_sl[R];                 // slot << 6 | lane; ~0: no record (16 bits are all in use)
    __shared__ float r_sg[R], r_w[R], r_c[3 * R], r_dt[R];
    __shared__ float bases[64 * BDS];
    __shared__ float gl[64 * 3];
    __shared__ float stage_all[W * 64 * KS];
    __shared__ int32_t seg_all[W * 64];
    __shared__ int32_t s_nb;
    // (the wavefront's number as a SCALAR: what is decided by it -- who forms terms, who prefetches -- is then
    // uniform control flow for the compiler too, with no per-lane merge of the values defined under it)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pw = wave - (W - P);                           // < 0: this wavefront forms no terms
    float* stage = stage_all + wave * 64 * KS;
    int32_t* seg = seg_all + wave * 64;
    const int32_t tabreg = rec_tab_reg(L, blockIdx.x, lane);
    const int64_t q = ray_of_thread(rays, (int64_t)blockIdx.x * 64 + lane);
    // (r03) What the tile needs of its rays is requested together with their list lengths, by every lane at a ray that
    // exists -- the lengths decide who needs it, but a second and a third round trip to memory in front of the first
    // round cost every tile with samples a tenth of its time (exp/trace_fused.py) -- and the table is cleared under it.
    const bool in_batch = q < rays.Q;
    const int64_t qc = in_batch ? q : 0;
    const bool from_tensors = rays.c2w == nullptr;           // (scalar) else the rays come from the camera: arithmetic only
    uint4 a = aux[qc];
    float g[C + 1], dsrc[3] = {0.f, 0.f, 1.f}, vsrc[3] = {0.f, 0.f, 1.f};
    if (from_tensors) {
#pragma unroll
        for (int j = 0; j < 3; ++j) dsrc[j] = rays.dirs[3 * qc + j];
        if constexpr (FMT == FMT_SH) {
#pragma unroll
            for (int j = 0; j < 3; ++j) vsrc[j] = rays.vdirs[3 * qc + j];
        }
    }
#pragma unroll
    for (int j = 0; j <= C; ++j) g[j] = grad_out[qc * (C + 1) + j];
    for (int i = threadIdx.x; i < T; i += NT) { keys[i] = -1; cnt[i] = 0; }
    if (!in_batch) a = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (XF) { if (a.x & kRecOverflow) a.x = 0u; }  // (done whole by the launch in front)
    const int nrec = (int)(a.x & ~kRecOverflow);
    int maxn = nrec;
    for (int off = 32; off > 0; off >>= 1) maxn = max(maxn, __shfl_xor(maxn, off, 64));
    maxn = __builtin_amdgcn_readfirstlane(maxn);             // everything that decides barriers is scalar
    if (maxn == 0) return;                                   // the same in every wavefront of the workgroup
    if constexpr (CHECK) { if (threadIdx.x == 0) atomicAdd(counters + kChkBase + 29, 1ull); }    // (tiles this instance worked on)

    Ray r;
    float basis[NB > 0 ? NB : 1];
    float accum = 0.f, light = 1.f;
    float light_ray = __uint_as_float(a.z);
    r.delta_scale = 0.f;
    if (from_tensors) {
        float dx, dy, dz;
        r.delta_scale = dir_to_tree(tr, dsrc, dx, dy, dz);   // setup_ray's, from the direction already at hand
        if constexpr (FMT == FMT_SH && !LOBES && !XF) precalc_basis<BD>(FMT_SH, BD, tr, vsrc[0], vsrc[1], vsrc[2], basis);
    } else if (nrec > 0) {
        setup_ray(tr, rays, opt, q, r);                      // for delta_scale (a ray with samples hits the cube)
        if constexpr (FMT == FMT_SH) {
            float vd[3];
            load_vdir(rays, q, vd);
            if constexpr (!LOBES && !XF) precalc_basis<BD>(FMT_SH, BD, tr, vd[0], vd[1], vd[2], basis);
            else { vsrc[0] = vd[0]; vsrc[1] = vd[1]; vsrc[2] = vd[2]; }
        }
    }
    if (wave == 0) {
        if constexpr (!EXACT) {
            if (nrec > 0) {
                const float* o = fwd_out + q * (C + 1);      // see render_bwd_kernel: single march
#pragma unroll
                for (int c = 0; c < C; ++c) accum += g[c] * o[c];
            }
        }
        if constexpr (LOBES) {
            // (the lobes one at a time straight into LDS: 25 unrolled exponentials at this kernel's 128 registers spill 106)
#pragma unroll 1
            for (int i = 0; i < NB; ++i) bases[lane * BDS + i] = lobe_value(opt.format, tr, vsrc[0], vsrc[1], vsrc[2], i, NB);
        } else if constexpr (XF) {                           // (bases: by whoever forms the ray's last record, in sweep 1)
#pragma unroll
            for (int j = 0; j < 3; ++j) vds[lane * 3 + j] = vsrc[j];
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) bases[lane * BDS + i] = basis[i];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) gl[lane * 3 + c] = g[c];
    }
    if constexpr (EXACT) {
        // ---- sweep 1: pass 1 of the reference without its atomics (accum_sample), W positions per
        // round; r_w / r_sg carry (att, total_color) from the wavefront that formed them to wavefront 0,
        // double-buffered by the parity of the round (one barrier per round)
        float light1 = 1.f;
        const int nr1 = (maxn + P - 1) / P;                  // the same in every wavefront
        // (r03) The hand-over of a round is requested one round ahead and the rounds' barriers order LDS
        // traffic only (lds_barrier), so the request stays in flight across them: a round used to begin with
        // a load from memory that nothing overlapped -- 2 900 shader-clock cycles per round of sweep 1, 5 100
        // per round of sweep 2, most of it that latency (exp/trace_fused.py, r02).
        // D1 rounds ahead (the latency of a request under load is longer than a round), each of the D1 register
        // sets refilled by the round that used it: the loop body is written out D1 times per trip so that no set is
        // ever copied into another (a copy waits for the load it copies)
        constexpr int D1 = 3;
        float4 tq0 = make_float4(0.f, 0.f, 0.f, 0.f), tq1 = tq0, tq2 = tq0;
        auto terms_at = [&](int k) {
            const int64_t blk = SVOXT_CHK(rec_block_u(L, tabreg, blockIdx.x, k >> 3), L.pool_blocks, 2);
            return terms[TERMS == 2 ? terms_index(blk, lane, k) : terms_index_pm(blk, lane, k)];
        };
        if constexpr (TERMS >= 2) {
            tq0 = terms_at(min(max(pw, 0), maxn - 1));
            tq1 = terms_at(min(max(pw, 0) + P, maxn - 1));
            tq2 = terms_at(min(max(pw, 0) + 2 * P, maxn - 1));
        }
        auto round1 = [&](int rd, float4& tq) {
            const int k = rd * P + pw;
            // The request of round rd + D1, by EVERY lane of the wavefront under SCALAR conditions: a lane whose list
            // ends before that position reads a stale slot of the same block and never looks at it (a load under a per-lane
            // condition comes with a merge of old and new value that waits for it on the spot); the block of
            // a position exists iff some ray of the tile has that many records, and the block index handed to
            // rec_block_u must be the same in all lanes.
            const float4 tv_cur = tq;
            if (pw >= 0 && rd < nr1 && k < nrec) {
                const int64_t blk = SVOXT_CHK(rec_block_u(L, tabreg, blockIdx.x, k >> 3), L.pool_blocks, 2);
                float att, ex[C];                             // exp(-x_c): sigmoid_d(x) = 1.0 / (1.0 + double(exp(-x)))
                if constexpr (TERMS >= 2) {
                    const float4 tv = tv_cur;
                    att = tv.x; ex[0] = tv.y; ex[1] = tv.z; ex[2] = tv.w;
                    if constexpr (XF) {
                        // (the forward's exponentials are those of each record's own basis; what this sweep still owes is
                        // the basis of the ray's LAST record, for sweep 2's total_color)
                        if (k == nrec - 1) {
                            const uint2 e = rec_get(L.rec + rec_index_in(blk, lane, k));
                            rotated_sh_basis<BD>(tr, SVOXT_CHK((int32_t)e.x, tr.M, 6), vsrc, basis);
#pragma unroll
                            for (int i = 0; i < NB; ++i) bases[lane * BDS + i] = basis[i];
                        }
                    }
                } else {
                    const uint2 e = rec_get(L.rec + rec_index_in(blk, lane, k));
                    float row[K];
                    load_row<K>(tr.features + (int64_t)SVOXT_CHK((int32_t)e.x, tr.M, 6) * K, row);
                    if constexpr (XF) {
                        rotated_sh_basis<BD>(tr, SVOXT_CHK((int32_t)e.x, tr.M, 6), vsrc, basis);
                        if (k == nrec - 1) {                 // what the reference's second pass keeps using
#pragma unroll
                            for (int i = 0; i < NB; ++i) bases[lane * BDS + i] = basis[i];
                        }
                    }
                    att = pexpf<true>(-__uint_as_float(e.y) * row[K - 1] * r.delta_scale);
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        float x;
                        if constexpr (FMT == FMT_SH) {
                            x = 0.f;
#pragma unroll
                            for (int i = 0; i < BD; ++i) x += basis[i] * row[c * BD + i];
                        } else {
                            x = row[c];
                        }
                        ex[c] = pexpf<true>(-x);
                    }
                }
                float total_color = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) total_color += (float)(1.0 / (1.0 + (double)ex[c])) * g[c];
                const int sl = SVOXT_CHK(((rd & 1) * P + pw) * 64 + lane, R, 0);
                r_w[sl] = att; r_sg[sl] = total_color;
            }
            // (unconditional, at a clamped position, and behind the last use of the set it refills: no merge of old and
            // new value, no second register set, no copy)
            if constexpr (TERMS >= 2) tq = terms_at(min(max(k, 0) + D1 * P, maxn - 1));
            if (wave == 0 && rd > 0) {
                // (the operands of the round's eight positions first, then what depends on the step before)
                float av[P], tv[P];
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    const int sl = SVOXT_CHK((((rd - 1) & 1) * P + j) * 64 + lane, R, 1);
                    av[j] = r_w[sl]; tv[j] = r_sg[sl];
                }
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    if ((rd - 1) * P + j < nrec) {
                        const float weight = light1 * (1.f - av[j]);
                        light1 *= av[j];
                        accum += weight * tv[j];
                    }
                }
            }
            if constexpr (TERMS >= 2) lds_barrier();
            else __syncthreads();
        };
        for (int rd = 0;;) {                                 // rounds 0 .. nr1 (the last one only advances)
            round1(rd, tq0); if (++rd > nr1) break;
            round1(rd, tq1); if (++rd > nr1) break;
            round1(rd, tq2); if (++rd > nr1) break;
        }
        if (wave == 0) {
            if (a.x & kRecOverflow) {                        // list + tail: from the tail-only launch in front
                light_ray = __uint_as_float(a.z);
                accum = __uint_as_float(a.w);
            } else {
                float total_grad = 0.f;
#pragma unroll
                for (int j = 0; j < C; ++j) total_grad += g[j];
                accum += light1 * opt.background_brightness * total_grad;
                light_ray = light1;
            }
        }
    }
    __syncthreads();

    // ---- sweep 2 (rt_kernel.cu:439-494 + the colour entries :410-425).
    // this wavefront's record (and hand-over) of the NEXT round, requested a round ahead: rounds follow each
    // other at k + P across the passes' boundaries as well
    constexpr bool PRE = EXACT && TERMS != 0;
    // (two rounds ahead, two register sets that take turns, each refilled in place: see sweep 1)
    constexpr int D2 = 2;
    uint2 eq0 = make_uint2(0u, 0u), eq1 = eq0;
    float4 uq0 = make_float4(0.f, 0.f, 0.f, 0.f), uq1 = uq0;
    auto request = [&](int k, uint2& eq, float4& uq) {
        k = min(max(k, 0), maxn - 1);                             // (a position some ray of the tile has: its block exists)
        const int64_t blk = SVOXT_CHK(rec_block_u(L, tabreg, blockIdx.x, k >> 3), L.pool_blocks, 2);
        eq = rec_get(L.rec + rec_index_in(blk, lane, k));
        uq = terms[TERMS == 2 ? terms_index(blk, lane, k) : terms_index_pm(blk, lane, k)];
    };
    if constexpr (PRE) {
        request(pw, eq0, uq0);
        request(pw + P, eq1, uq1);
    }
    // (r03) A PASS is as many rounds as fit the record arrays, its records compact: the slot of (position k, ray l)
    // is the number of records of the pass in front of position k plus the number of rays below l that have
    // position k -- both from the ballot "nrec > k", which every wavefront forms for itself (lane = ray in all
    // of them).  With the fixed window of 14 positions the arrays were a third full on the headline workload
    // (~300 of 896 slots: lists are ragged), i.e. three times the scans, sorts, table clears and barriers, and a
    // feature row left the CU once per window instead of once per ~40 positions.
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    // one round: positions kb .. kb + P - 1 (kb uniform), records from slot `nslot` on; returns the round's records
    auto round = [&](int kb, int nslot, uint2& eq, float4& uq) -> int {
        int basep[P], total = 0;                                 // slots in front of position kb + j within the round
        unsigned long long maskp[P];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            maskp[j] = __ballot(nrec > kb + j);
            basep[j] = total;
            total += (int)__popcll(maskp[j]);
        }
        const int k = kb + pw;
        uint2 e = eq;
        const float4 tv_pre = uq;
        if (pw >= 0 && k < nrec) {
            int myb = 0;
            unsigned long long mym = 0ull;
#pragma unroll
            for (int j = 0; j < P; ++j) { if (j == pw) { myb = basep[j]; mym = maskp[j]; } }      // (pw is scalar)
            const int slot = SVOXT_CHK(nslot + myb + (int)__popcll(mym & lane_lt), R, 3);
            const int64_t blk = SVOXT_CHK(rec_block_u(L, tabreg, blockIdx.x, k >> 3), L.pool_blocks, 2);
            if constexpr (!PRE) e = rec_get(L.rec + rec_index_in(blk, lane, k));
            float att, tc, cf[C];
            if constexpr (EXACT && TERMS != 0) {
                const float4 tv = tv_pre;
                const float ex[C] = {tv.y, tv.z, tv.w};
                att = tv.x;
                tc = 0.f;
                float row[XF ? K : 1];
                if constexpr (XF) load_row<K>(tr.features + (int64_t)SVOXT_CHK((int32_t)e.x, tr.M, 6) * K, row);
#pragma unroll
                for (int c = 0; c < C; ++c) {                     // sample_terms from here on, operation for operation
                    const double sd = 1.0 / (1.0 + (double)ex[c]);
                    if constexpr (FMT == FMT_SH) {
                        const float sig = (float)sd;
                        cf[c] = (float)((double)sig * (1.0 - (double)sig));
                    } else {
                        cf[c] = (float)sd;
                    }
                    if constexpr (XF) {
                        // total_color of the second pass: the basis the first pass ended with (coef_sample<..., XF>)
                        float tmp2 = 0.f;
#pragma unroll
                        for (int i = 0; i < BD; ++i) tmp2 += bases[lane * BDS + i] * row[c * BD + i];
                        tc = (float)((double)tc + sigmoid_d<true>(tmp2) * (double)g[c]);
                    } else {
                        tc = (float)((double)tc + sd * (double)g[c]);
                    }
                }
            } else {
                float row[K];
                load_row<K>(tr.features + (int64_t)SVOXT_CHK((int32_t)e.x, tr.M, 6) * K, row);
                if constexpr (XF) {
                    // coef_sample<..., XF>'s operations: the colour coefficients from the record's own basis, total_color
                    // from the basis sweep 1 ended with (read from LDS column by column: nine registers less)
                    rotated_sh_basis<BD>(tr, SVOXT_CHK((int32_t)e.x, tr.M, 6), vsrc, basis);
                    att = pexpf<true>(-__uint_as_float(e.y) * row[K - 1] * r.delta_scale);
                    tc = 0.f;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        float tmp = 0.f, tmp2 = 0.f;
#pragma unroll
                        for (int i = 0; i < BD; ++i) tmp += basis[i] * row[c * BD + i];
                        const float sig = (float)sigmoid_d<true>(tmp);
                        cf[c] = (float)((double)sig * (1.0 - (double)sig));
#pragma unroll
                        for (int i = 0; i < BD; ++i) tmp2 += bases[lane * BDS + i] * row[c * BD + i];
                        tc = (float)((double)tc + sigmoid_d<true>(tmp2) * (double)g[c]);
                    }
                } else {
                    sample_terms<FMT, C, BD, K>(row, basis, g, __uint_as_float(e.y), r.delta_scale, att, tc, cf);
                }
            }
            const int32_t idx = SVOXT_CHK((int32_t)e.x, tr.M, 6);
            uint32_t h = ((uint32_t)idx * 0x9E3779B1u) >> (32 - __builtin_ctz(T));
            while (true) {
                const int32_t old = atomicCAS(keys + h, -1, idx);
                if (old == -1 || old == idx) break;
                h = (h + 1u) & (uint32_t)(T - 1);
            }
            h = (uint32_t)SVOXT_CHK((int)h, T, 4);
            atomicAdd(cnt + h, 1);
            r_sl[slot] = (h << 6) | (uint32_t)lane;
            r_w[slot] = att; r_sg[slot] = tc; r_dt[slot] = __uint_as_float(e.y);
            r_c[slot] = cf[0]; r_c[R + slot] = cf[1]; r_c[2 * R + slot] = cf[2];
        }
        if constexpr (PRE) {
            request(k + D2 * P, eq, uq);
            lds_barrier();
        } else {
            __syncthreads();
        }
        if (wave == 0) {
            float av[P], tv[P], dv[P];
            int sl[P];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                sl[j] = nslot + basep[j] + (int)__popcll(maskp[j] & lane_lt);
                if (kb + j < nrec) sl[j] = SVOXT_CHK(sl[j], R, 5);
                av[j] = 1.f; tv[j] = 0.f; dv[j] = 0.f;
                if (kb + j < nrec) { av[j] = r_w[sl[j]]; tv[j] = r_sg[sl[j]]; dv[j] = r_dt[sl[j]]; }
            }
#pragma unroll
            for (int j = 0; j < P; ++j) {
                if (kb + j < nrec) {
                    const int s2 = sl[j];
                    float cf[C] = {0.f, 0.f, 0.f};
                    if constexpr (FMT == FMT_RGBA) { cf[0] = r_c[s2]; cf[1] = r_c[R + s2]; cf[2] = r_c[2 * R + s2]; }
                    float wgt, sg;
                    sample_advance<FMT, C>(av[j], tv[j], cf, g, dv[j], r.delta_scale, light_ray, light, accum, wgt, sg);
                    r_w[s2] = wgt; r_sg[s2] = sg;
                    if constexpr (FMT == FMT_RGBA) { r_c[s2] = cf[0]; r_c[R + s2] = cf[1]; r_c[2 * R + s2] = cf[2]; }
                }
            }
        }
        return total;
    };
    // records of the round that starts at position kb (scalar)
    auto round_records = [&](int kb) {
        int n = 0;
#pragma unroll
        for (int j = 0; j < P; ++j) n += (int)__popcll(__ballot(nrec > kb + j));
        return n;
    };
    int kb = 0;
    while (kb < maxn) {
        // ---- terms + advance: rounds of P list positions, two at a time (the two register sets of the prefetch take
        // turns and a pass is a whole number of such pairs: the second round of a tile's last pair may be empty), while
        // the next pair still fits
        int npass = 0;                                           // records of this pass
        while (true) {
            npass += round(kb, npass, eq0, uq0);
            kb += P;
            npass += round(kb, npass, eq1, uq1);
            kb += P;
            if (kb >= maxn || npass + round_records(kb) + round_records(kb + P) > R) break;
        }
        const int nscan = npass;                                 // (svoxt_tile_reduce.inc: the slots in use are 0 .. nscan - 1)
#include "svoxt_tile_reduce.inc"
        if (kb >= maxn) break;                                   // last pass (scalar condition)
        lds_barrier();
        for (int i = threadIdx.x; i < T; i += NT) { keys[i] = -1; cnt[i] = 0; }
        lds_barrier();
    }
#undef SVOXT_CHK
}

// The backward of an image for RGBA-style rows of 8 / 16 / 32 floats (C = K - 1 = 7 / 15 / 31
// channels), exact, per 64-ray tile like grad_fused_kernel -- but a lane cannot keep 31 channels
// of a ray (render_bwd_kernel<RGBA, 31>: 238 registers, 1.2 wavefronts per SIMD in flight, 3 ms
// at 1024 x 1024 on a depth-9 tree, VALU 27 % busy: r02 PMC), and 40 M atomic requests (a row
// per sample) are 2 ms at the memory side's 22 G requests/s whatever the kernel does.  Here:
//   sweep 1  (rt_kernel.cu:365-437 without its atomics) per window of 16 list positions: the
//            records are compacted; lane = RECORD forms the C sigmoids of its row ONCE -- every
//            lane busy, the row as K / 4 loads, upstream gradients from LDS -- and from them both
//            total_colors the reference forms (float sigmoids :424, double quotients :470);
//            wavefront 0 then runs along the rays (weight, transmittance, accum).  The attenuation
//            and the second pass's total_color go to L.terms, 8 bytes per record (position-major).
//   sweep 2  (:439-494 + the colour entries :419-425) per window: lane = ray takes the attenuation
//            from the hand-over and enters the record's feature row in a hash table; wavefront
//            0 runs along the rays (accum -= weight * total_color; sigma entries); counting sort
//            by feature row; then lane = COLUMN: a group of K lanes takes one distinct row, forms
//            the sigmoid of its column once per (tile, window, row), adds up the row's records
//            -- ((weight * sig) * (1 - sig)) * g_c of the record's ray, the reference's order --
//            and sends ONE atomic row: sigmoid work and requests divided by the reuse (2.0-2.6x
//            at 1024 x 1024 / depth 9, exp/reuse_probe.py).
// Rays whose list overflowed: a tail-only launch of render_bwd_kernel<..., GATHER> in front
// (aux.z / .w carry its pass-1 results), as for grad_fused_kernel<EXACT>.
// FAST (opt-in tolerance mode, SVOXT_LISTS_NATIVE_MATH): the sigmoids and the attenuation of both sweeps
// with the hardware's exponential and reciprocal (nsigmoidf / nexpf: ~5 instructions where the exact
// replica of expf plus a double-precision divide are ~45 -- r02 PMC: 783 M vector instructions, 65 %
// VALU busy at 1024 x 1024 / depth 9), the second pass's total_color summed in float; lists, hash,
// sort and the order of the additions unchanged.  Gradients within 1e-5 of the tight scale.
// COUNT (instrumentation, svoxt_set_bwd_counters): counters[0] += 64-byte atomic requests sent (a row of
// K floats that starts on a K * 4-byte boundary: two for K = 32, one for K = 16 / 8), counters[1] +=
// (tile, window, feature row) groups.
// (r03: every barrier of this kernel is lds_barrier() -- its phases hand over through LDS alone, the hand-over
// between the sweeps through memory is written and read by the same lane -- so the gradient atomics and the
// hand-over stores stay in flight across them.)
// CHECK (svoxt_set_bwd_check; r04): as for grad_fused_kernel; the sites of this kernel: 16 a list block outside the
// pool, 17 sweep 1's compacted position, 18 a compacted record's slot, 19 a feature row outside the table, 20 sweep
// 2's table entry, 21 the list of occupied entries, 22 a record's place in the sorted order, 23 an occupied entry
// read by the reduce, 24 a sorted record read by the reduce, 25 the row a sum is sent to.
// (r05, measured and not kept -- exp/r05/grad_wide_split_sweeps.diff.txt: the two sweeps as two launches (sweep 1 leaves accum and the
// final transmittance in aux, as the tail-only launch does), each compiled for its own occupancy: backward 1.31 -> 1.41-1.42 ms at
// 8 / 8, 8 / 6 and 6 / 6 wavefronts per SIMD.  Sweep 1 alone needs 72 registers without scratch, sweep 2 alone 80 -- no gain there --
// and as one launch the workgroups' latency-bound sweeps overlap the atomic-rate-bound reduces of their neighbours, which two
// launches give up.  Profiled apart at config 4 (exp/prof_lib.sh): sweep 1 0.487 ms -- no atomics: its row gathers, lane = record, and
// ~17 instructions per column -- sweep 2 0.878 ms against the 0.54 ms its 11.8 M atomic requests take at the memory side.)
// ETAB (exact mode, r04): the rows are read from tr.etab -- etab[row][c] = pexpf(-features[row][c]), sigma in the last
// column (exp_table_kernel, built once per forward) -- so neither sweep forms the exponential of a feature again, and
// the double-precision reciprocals 1 / (1 + e) take rcp_unit_range (the compiler's division sequence minus what their
// operand range makes an identity).  The same bits.
template <int K, bool FAST = false, bool COUNT = false, bool CHECK = false, bool ETAB = false>
__global__ void __launch_bounds__(512, FAST ? 8 : ETAB ? SVOXT_WIDE_ETAB_WAVES : 6)
grad_wide_kernel(TreeDev tr, RaysDev rays, Opts opt, const float* __restrict__ grad_out,
                 RecLists L, const uint4* __restrict__ aux, float* __restrict__ grad, int gstride,
                 unsigned long long* __restrict__ counters = nullptr) {
    static_assert(K == 8 || K == 16 || K == 32, "row widths with an instance");
    static_assert(!(ETAB && FAST), "the table holds the exact exponentials");
    const float* __restrict__ const rows = ETAB ? tr.etab : tr.features;
    // (r04) Which tile: workgroup b runs on XCD b mod 8; of every 64 consecutive tiles -- a super-tile when the image is
    // walked in super-tiles, else 64 neighbours of a tile row -- XCD x takes tiles 8 x .. 8 x + 7, eight neighbours in a
    // row behind ONE L2, instead of every eighth tile (SVOXT_WIDE_XCD_ROWS 0: tile = workgroup)
    unsigned wtile = blockIdx.x;
    if (SVOXT_WIDE_XCD_ROWS != 0 && (blockIdx.x | 63u) < gridDim.x)
        wtile = (blockIdx.x & ~63u) | ((blockIdx.x & 7u) << 3) | ((blockIdx.x >> 3) & 7u);
    if (tile_never_recorded(L, wtile)) return;               // (r03: before anything else is requested)
#define SVOXT_CHK(i, n, site) chk<CHECK>(i, n, counters, site)
    constexpr int C = K - 1, W = 8, NT = 64 * W, T = 1024, R = 1024, RPP = R / (64 * W);
    constexpr int KG = K | 1;                                // odd stride: conflict-free gradient rows
    constexpr int SPW = 64 / K;                              // distinct rows a wavefront reduces at a time
    static_assert(RPP == 2, "sizes");
    __shared__ int32_t keys[T];
    __shared__ int32_t cnt[T];
    __shared__ uint16_t order[R];                // sweep 1: the window's records, compacted
    static_assert(T <= R, "slots shares order's storage");
    uint16_t* const slots = order;               // sweep 2: the occupied table entries (order is sweep 1's alone)
    __shared__ uint32_t r_sl[R];                 // sweep 1: feature row; sweep 2: table entry << 6 | lane, ~0: no record
    __shared__ float r_w[R], r_sg[R], r_t1[R], r_dt[R];   // (sweep 2: r_t1 / r_dt hold weight / sigma entry in sorted order)
    __shared__ uint8_t s_ray[R];                 // sweep 2: the ray of the sorted record
    __shared__ float gl[64 * KG];
    __shared__ float dsl[64];
    __shared__ int32_t s_nb, s_ns;
    float2* __restrict__ tot2 = reinterpret_cast<float2*>(L.terms);      // (attenuation, second-pass total_color) per record
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t tabreg = rec_tab_reg(L, wtile, lane);
    const int64_t q = ray_of_thread(rays, (int64_t)wtile * 64 + lane);
    uint4 a = make_uint4(0u, 0u, 0u, 0u);
    if (q < rays.Q) a = aux[q];
    const int nrec = (int)(a.x & ~kRecOverflow);
    int maxn = nrec;
    for (int off = 32; off > 0; off >>= 1) maxn = max(maxn, __shfl_xor(maxn, off, 64));
    maxn = __builtin_amdgcn_readfirstlane(maxn);             // everything that decides barriers is scalar
    if (maxn == 0) return;                                   // the same in every wavefront of the workgroup
    if constexpr (CHECK) { if (threadIdx.x == 0) atomicAdd(counters + kChkBase + 29, 1ull); }    // (tiles this instance worked on)

    float ds = 0.f, g_sig = 0.f;
    if (nrec > 0) {
        Ray r;
        setup_ray(tr, rays, opt, q, r);                      // for delta_scale (a ray with samples hits the cube)
        ds = r.delta_scale;
        g_sig = grad_out[q * K + C];
    }
    if (wave == 0) {
        dsl[lane] = ds;
#pragma unroll 1
        for (int j = 0; j < C; ++j) gl[lane * KG + j] = nrec > 0 ? grad_out[q * K + j] : 0.f;
    }
    if (threadIdx.x == 0) { s_nb = 0; s_ns = 0; }
    lds_barrier();

    // ---- sweep 1
    float light1 = 1.f, accum = 0.f, light_ray = 1.f;
    for (int k0 = 0; k0 < maxn; k0 += RPP * W) {
        // records of the window, compacted in (position, lane) order within a wavefront's round
        {
            uint2 e[RPP];
            bool have[RPP];
#pragma unroll
            for (int rd = 0; rd < RPP; ++rd) {
                const int k = k0 + rd * W + wave;
                have[rd] = k < nrec;
                e[rd] = make_uint2(0u, 0u);
                if (have[rd]) e[rd] = rec_get(L.rec + rec_index_in(SVOXT_CHK(rec_block_u(L, tabreg, wtile, k >> 3), L.pool_blocks, 16), lane, k));
            }
#pragma unroll
            for (int rd = 0; rd < RPP; ++rd) {
                const unsigned long long m = __ballot(have[rd]);
                if (m != 0ull) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&s_nb, __popcll(m));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (have[rd]) {
                        const int slot = (rd * W + wave) * 64 + lane;
                        r_sl[slot] = e[rd].x;
                        r_dt[slot] = __uint_as_float(e[rd].y);
                        order[SVOXT_CHK(base + (int)__popcll(m & ((1ull << lane) - 1ull)), R, 17)] = (uint16_t)slot;
                    }
                }
            }
        }
        lds_barrier();
        const int nb1 = __builtin_amdgcn_readfirstlane(s_nb);
        // G = K / 8 neighbouring lanes per record, 8 row columns each: four times the busy lanes and a
        // quarter of the dependent work per lane (a window holds a few hundred records for 512 lanes);
        // the sums over the channels run in the reference's order, handed from lane to lane
        constexpr int G = K / 8;
        const int gq = threadIdx.x & (G - 1);
        if constexpr (!FAST && ETAB) {
            // (r04) lane = RECORD, the whole row: with the exponentials in the table a column costs a reciprocal and two
            // products, and what dominated the G-lanes-per-record form below was its bookkeeping -- the two ordered sums
            // over the channels hopped from lane to lane G - 1 times, every hop's eight-step chain issued for all G lanes
            // with one of them active: 195 of the 328 instructions per record and lane (ISA, K = 32).  Here a lane runs
            // its record's 31 columns in order by itself: the same operations in the same order, ~17 instructions per
            // column, no shuffle; a window's first positions hold up to 1 024 records for the 512 lanes.
            for (int p = threadIdx.x; p < nb1; p += NT) {
                const int slot = SVOXT_CHK((int)order[SVOXT_CHK(p, R, 17)], R, 18);
                const int ray = slot & 63;
                float row[K];
                load_row<K>(rows + (int64_t)SVOXT_CHK((int32_t)r_sl[slot], tr.M, 19) * K, row);
                const float* __restrict__ gr = gl + ray * KG;
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const double sd = rcp_unit_range(1.0 + (double)row[c]);
                    const float gj = gr[c];
                    t1 += (float)sd * gj;
                    t2 = (float)((double)t2 + sd * (double)gj);
                }
                r_w[slot] = pexpf<true>(-r_dt[slot] * row[K - 1] * dsl[ray]);
                r_t1[slot] = t1;
                r_sg[slot] = t2;
            }
        } else
        for (int p0 = wave * (64 / G); p0 < nb1; p0 += NT / G) {      // (scalar bounds: every lane takes part in the shuffles)
            const int p = p0 + (lane / G);
            const bool on = p < nb1;
            const int slot = on ? SVOXT_CHK((int)order[SVOXT_CHK(p, R, 17)], R, 18) : 0;
            const int ray = slot & 63;
            float row[8];
            if constexpr (FAST) {
                // tolerance mode: no order of the additions to reproduce -- every lane sums its 8 columns,
                // two xor-shuffles add the G partial sums (where the exact form hops G - 1 times from lane to
                // lane with the float and the double sum), one float serves both passes' total_color
                float part = 0.f;
                if (on) {
                    load_row<8>(tr.features + (int64_t)SVOXT_CHK((int32_t)r_sl[slot], tr.M, 19) * K + 8 * gq, row);
                    const float* __restrict__ gr = gl + ray * KG + 8 * gq;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (8 * gq + j < C) part = __builtin_fmaf(nsigmoidf(row[j]), gr[j], part);
                    }
                }
#pragma unroll
                for (int off = 1; off < G; off <<= 1) part += __shfl_xor(part, off, 64);
                if (on && gq == G - 1) {
                    r_w[slot] = nexpf(-r_dt[slot] * row[7] * dsl[ray]);
                    r_t1[slot] = part;
                    r_sg[slot] = part;
                }
            } else {
            float a1[8];
            double a2[8];
            if (on) {
                load_row<8>(rows + (int64_t)SVOXT_CHK((int32_t)r_sl[slot], tr.M, 19) * K + 8 * gq, row);
                const float* __restrict__ gr = gl + ray * KG + 8 * gq;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (8 * gq + j < C) {                      // (the last lane's eighth column is sigma)
                        const double sd = ETAB ? rcp_unit_range(1.0 + (double)row[j]) : sigmoid_d<true>(row[j]);
                        const float gj = gr[j];
                        a1[j] = (float)sd * gj;
                        a2[j] = sd * (double)gj;
                    } else {
                        a1[j] = 0.f; a2[j] = 0.0;
                    }
                }
            }
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int rr = 0; rr < G; ++rr) {
                float c1 = 0.f, c2 = 0.f;
                if (rr > 0) { c1 = __shfl_up(t1, 1, 64); c2 = __shfl_up(t2, 1, 64); }
                if (on && gq == rr) {
                    t1 = c1; t2 = c2;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (rr * 8 + j < C) {
                            t1 += a1[j];
                            t2 = (float)((double)t2 + a2[j]);
                        }
                    }
                }
            }
            if (on && gq == G - 1) {
                r_w[slot] = pexpf<true>(-r_dt[slot] * row[7] * dsl[ray]);
                r_t1[slot] = t1;
                r_sg[slot] = t2;
            }
            }
        }
        lds_barrier();
        if (threadIdx.x == 0) s_nb = 0;
        // along the rays (wavefront 0); everybody: the hand-over, 256 contiguous bytes per position
#pragma unroll 1
        for (int rd = 0; rd < RPP; ++rd) {
            const int k = k0 + rd * W + wave;
            if (k < nrec) {
                const int64_t blk = SVOXT_CHK(rec_block_u(L, tabreg, wtile, k >> 3), L.pool_blocks, 16);
                const int sl = (rd * W + wave) * 64 + lane;
                tot2[terms_index_pm(blk, lane, k)] = make_float2(r_w[sl], r_sg[sl]);
            }
        }
        if (wave == 0) {
            // (the operands of eight positions at a time first: the dependent part is then two multiplies and an add per step)
#pragma unroll
            for (int j0 = 0; j0 < RPP * W; j0 += 8) {
                float av[8], tv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { av[j] = r_w[(j0 + j) * 64 + lane]; tv[j] = r_t1[(j0 + j) * 64 + lane]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (k0 + j0 + j < nrec) {
                        const float weight = light1 * (1.f - av[j]);
                        light1 *= av[j];
                        accum += weight * tv[j];
                    }
                }
            }
        }
        lds_barrier();
    }
    if (wave == 0) {
        if (a.x & kRecOverflow) {                            // list + tail: from the tail-only launch in front
            light_ray = __uint_as_float(a.z);
            accum = __uint_as_float(a.w);
        } else {
            float total_grad = 0.f;
#pragma unroll 1
            for (int j = 0; j < C; ++j) total_grad += gl[lane * KG + j];
            accum += light1 * opt.background_brightness * total_grad;
            light_ray = light1;
        }
    }

    // ---- sweep 2
    float light = 1.f;
    for (int i = threadIdx.x; i < T; i += NT) { keys[i] = -1; cnt[i] = 0; }
    for (int i = threadIdx.x; i < R; i += NT) r_sl[i] = 0xffffffffu;
    lds_barrier();
    for (int k0 = 0; k0 < maxn; k0 += RPP * W) {
        {   // the window's two rounds together: records and hand-over first, then the sigma gathers, then the table
            uint2 e[RPP];
            float2 h[RPP];
            bool have[RPP];
#pragma unroll
            for (int rd = 0; rd < RPP; ++rd) {
                const int k = k0 + rd * W + wave;
                have[rd] = k < nrec;
                e[rd] = make_uint2(0u, 0u);
                h[rd] = make_float2(0.f, 0.f);
                if (have[rd]) {
                    const int64_t blk = SVOXT_CHK(rec_block_u(L, tabreg, wtile, k >> 3), L.pool_blocks, 16);
                    e[rd] = rec_get(L.rec + rec_index_in(blk, lane, k));
                    h[rd] = tot2[terms_index_pm(blk, lane, k)];
                }
            }
#pragma unroll
            for (int rd = 0; rd < RPP; ++rd) {
                if (have[rd]) {
                    const int32_t idx = SVOXT_CHK((int32_t)e[rd].x, tr.M, 19);
                    uint32_t h32 = ((uint32_t)idx * 0x9E3779B1u) >> (32 - __builtin_ctz(T));
                    while (true) {
                        const int32_t old = atomicCAS(keys + h32, -1, idx);
                        if (old == -1 || old == idx) break;
                        h32 = (h32 + 1u) & (uint32_t)(T - 1);
                    }
                    h32 = (uint32_t)SVOXT_CHK((int)h32, T, 20);
                    atomicAdd(cnt + h32, 1);
                    const int slot = (rd * W + wave) * 64 + lane;
                    r_sl[slot] = (h32 << 6) | (uint32_t)lane;
                    r_w[slot] = h[rd].x;              // the attenuation sweep 1 formed: no sigma gather here
                    r_sg[slot] = h[rd].y;
                    r_dt[slot] = __uint_as_float(e[rd].y);
                }
            }
        }
        lds_barrier();
        if (wave == 0) {                                     // along the rays: (att, total_color) -> (weight, sigma entry)
#pragma unroll
            for (int j0 = 0; j0 < RPP * W; j0 += 8) {        // eight positions' operands at a time
                float av[8], tv[8], dv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int s2 = (j0 + j) * 64 + lane;
                    av[j] = r_w[s2]; tv[j] = r_sg[s2]; dv[j] = r_dt[s2];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (k0 + j0 + j < nrec) {
                        const int s2 = (j0 + j) * 64 + lane;
                        const float weight = light * (1.f - av[j]);
                        light *= av[j];
                        accum -= weight * tv[j];
                        r_w[s2] = weight;
                        r_sg[s2] = dv[j] * ds * (tv[j] * light - accum) + dv[j] * ds * g_sig * light_ray;
                    }
                }
            }
        } else if (wave == 1) {                              // meanwhile: where each row's records go, and which entries are in use
            constexpr int PER = T / 64;
            int mine[PER], sum = 0, used = 0;
#pragma unroll
            for (int j2 = 0; j2 < PER; ++j2) { mine[j2] = cnt[lane * PER + j2]; sum += mine[j2]; used += mine[j2] > 0 ? 1 : 0; }
            int incl = sum, uincl = used;
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off, 64);
                const int u = __shfl_up(uincl, off, 64);
                if (lane >= off) { incl += v; uincl += u; }
            }
            int run = incl - sum, urun = uincl - used;
#pragma unroll
            for (int j2 = 0; j2 < PER; ++j2) {
                cnt[lane * PER + j2] = run;
                run += mine[j2];
                if (mine[j2] > 0) slots[SVOXT_CHK(urun++, T, 21)] = (uint16_t)(lane * PER + j2);
            }
            if (lane == 63) { s_nb = incl; s_ns = uincl; }
        }
        lds_barrier();
        for (int rr = threadIdx.x; rr < R; rr += NT) {
            const uint32_t v = r_sl[rr];
            if (v != 0xffffffffu) {
                const int pos = SVOXT_CHK(atomicAdd(cnt + SVOXT_CHK((int)(v >> 6), T, 20), 1), R, 22);
                r_t1[pos] = r_w[rr];                         // (r_dt was consumed by the chain above)
                r_dt[pos] = r_sg[rr];
                s_ray[pos] = (uint8_t)(v & 63u);
            }
        }
        lds_barrier();
        // ---- reduce: lane = column; after the scatter cnt[h] is where the records of entry h END
        const int ns = SVOXT_CHK(__builtin_amdgcn_readfirstlane(s_ns), T + 1, 21);
        if constexpr (COUNT) {
            if (threadIdx.x == 0 && ns > 0) {
                atomicAdd(counters, (unsigned long long)ns * (K * 4 > 64 ? (K * 4) / 64 : 1));
                atomicAdd(counters + 1, (unsigned long long)ns);
            }
        }
        const int col = lane & (K - 1), sub = lane / K;
        // (r04) Two stages of D rows each: the rows of the NEXT stage are requested before this stage's sums are sent.
        // A wavefront's vector-memory operations complete in order, the atomics among them: rows requested behind a
        // stage's atomics arrive after the memory side has taken those atomics AND a trip to memory; requested a stage
        // ahead they have the stage's arithmetic to arrive in.
        constexpr int D = SVOXT_WIDE_STAGE_ROWS;             // rows in flight per lane and stage
        struct Stage { int32_t idxs[D]; int ps[D], pes[D]; float xs[D]; };
        auto fetch = [&](Stage& st, int i0) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int i = i0 + u * W * SPW + sub;
                st.idxs[u] = -1; st.ps[u] = 0; st.pes[u] = 0; st.xs[u] = 0.f;
                if (i < ns) {
                    const int h = SVOXT_CHK((int)slots[i], T, 23);
                    st.idxs[u] = SVOXT_CHK(keys[h], tr.M, 25);
                    st.pes[u] = SVOXT_CHK(cnt[h], R + 1, 24);
                    // the entry's records start where the previous occupied entry's end (table order = sorted order)
                    if (i > 0) st.ps[u] = SVOXT_CHK(cnt[SVOXT_CHK((int)slots[i - 1], T, 23)], R + 1, 24);
                    if constexpr (CHECK) { if (st.ps[u] > st.pes[u]) { atomicAdd(counters + kChkBase + 24, 1ull); st.ps[u] = st.pes[u]; } }
                    st.xs[u] = rows[(int64_t)st.idxs[u] * K + col];
                }
            }
        };
        auto send = [&](const Stage& st, int i0) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                if (i0 + u * W * SPW >= ns) break;           // (scalar)
                const int32_t idx = st.idxs[u];
                const int p = st.ps[u];
                float sig = 0.f, om = 0.f;
                if (idx >= 0 && col < C) {
                    sig = FAST ? nsigmoidf(st.xs[u]) : ETAB ? (float)rcp_unit_range(1.0 + (double)st.xs[u]) : (float)sigmoid_d<true>(st.xs[u]);
                    om = 1.f - sig;
                }
                float acc = 0.f;
                int n_here = st.pes[u] - p, n_max = n_here;
                for (int off = 32; off >= K; off >>= 1) n_max = max(n_max, __shfl_xor(n_max, off, 64));
                n_max = __builtin_amdgcn_readfirstlane(n_max);
                for (int t = 0; t < n_max; ++t) {
                    if (t < n_here) {
                        float val;
                        if (col == C) val = r_dt[p + t];
                        else val = r_t1[p + t] * sig * om * gl[(int)s_ray[p + t] * KG + col];
                        acc += val;
                    }
                }
                if (idx >= 0) atomicAdd(grad + (int64_t)idx * gstride + col, acc);
            }
        };
        constexpr int STEP = D * W * SPW;
        Stage sa, sb;
        fetch(sa, wave * SPW);
        for (int i0 = wave * SPW; i0 < ns; i0 += 2 * STEP) {
            fetch(sb, i0 + STEP);
            send(sa, i0);
            if (i0 + STEP >= ns) break;                      // (scalar)
            fetch(sa, i0 + 2 * STEP);
            send(sb, i0 + STEP);
        }
        if (k0 + RPP * W >= maxn) break;                     // last window (scalar condition)
        lds_barrier();
        for (int i = threadIdx.x; i < T; i += NT) { keys[i] = -1; cnt[i] = 0; }
        for (int i = threadIdx.x; i < R; i += NT) r_sl[i] = 0xffffffffu;
        lds_barrier();
    }
#undef SVOXT_CHK
}

// Generic backward: any K / format / component range / channel count
// (C == 0 is opacity_render_backward, rt_kernel.cu:1593-1616).
template <bool N2>
__global__ void __launch_bounds__(kBlock)
render_bwd_generic_kernel(TreeDev tr, RaysDev rays, Opts opt, int C,
                          const float* __restrict__ grad_out, float* __restrict__ grad, int gstride) {
    const int64_t q = ray_of_thread(rays, (int64_t)blockIdx.x * kBlock + threadIdx.x);
    if (q >= rays.Q) return;
    Ray r;
    if (!setup_ray(tr, rays, opt, q, r)) return;
    const int K = tr.K;
    const float* g = grad_out + q * (C + 1);
    float basis[25];
    float vd[3];
    load_vdir(rays, q, vd);
    precalc_basis<0>(opt.format, opt.basis_dim, tr, vd[0], vd[1], vd[2], basis);
    float accum = 0.f;
    float light_ray;
    {   // pass 1
        float light = 1.f, t = r.tmin;
        while (t < r.tmax) {
            Sample s;
            march_step<N2>(tr, r, opt.step_size, t, s);
            if (s.valid) {
                const float* row = tr.features + (int64_t)s.idx * K;
                const float sigma = row[K - 1];
                if (sigma > 0.f) {
                    float* grow = grad + (int64_t)s.idx * gstride;
                    // pass 1 re-evaluates the rotated basis; pass 2 keeps the last one
                    if (tr.xform != nullptr) rotated_basis(tr, opt.format, opt.basis_dim, s.idx, vd, basis);
                    const float att = pexpf(-s.delta_t * sigma * r.delta_scale);
                    const float weight = light * (1.f - att);
                    float total_color = 0.f;
                    if (opt.format != FMT_RGBA) {
                        for (int c = 0; c < C; ++c) {
                            const int off = c * opt.basis_dim;
                            float tmp = 0.f;
                            for (int i = opt.min_comp; i <= opt.max_comp; ++i) tmp += basis[i] * row[off + i];
                            const float sig = (float)(1.0 / (1.0 + (double)pexpf(-tmp)));
                            const float gsig = (float)((double)sig * (1.0 - (double)sig));
                            for (int i = opt.min_comp; i <= opt.max_comp; ++i)
                                atomicAdd(grow + off + i, weight * basis[i] * gsig * g[c]);
                            total_color += sig * g[c];
                        }
                    } else {
                        for (int j = 0; j < C; ++j) {
                            const float sig = (float)(1.0 / (1.0 + (double)pexpf(-row[j])));
                            atomicAdd(grow + j, weight * sig * (1.f - sig) * g[j]);
                            total_color += sig * g[j];
                        }
                    }
                    light *= att;
                    accum += weight * total_color;
                }
            }
            t = march_advance(t, s.delta_t);
        }
        float total_grad = 0.f;
        for (int j = 0; j < C; ++j) total_grad += g[j];
        accum += light * opt.background_brightness * total_grad;
        light_ray = light;
    }
    {   // pass 2
        float light = 1.f, t = r.tmin;
        while (t < r.tmax) {
            Sample s;
            march_step<N2>(tr, r, opt.step_size, t, s);
            if (s.valid) {
                const float* row = tr.features + (int64_t)s.idx * K;
                const float sigma = row[K - 1];
                if (sigma > 0.f) {
                    float total_color = 0.f;
                    if (opt.format != FMT_RGBA) {
                        for (int c = 0; c < C; ++c) {
                            const int off = c * opt.basis_dim;
                            float tmp = 0.f;
                            for (int i = opt.min_comp; i <= opt.max_comp; ++i) tmp += basis[i] * row[off + i];
                            total_color = (float)((double)total_color + 1.0 / (1.0 + (double)pexpf(-tmp)) * (double)g[c]);
                        }
                    } else {
                        for (int j = 0; j < C; ++j)
                            total_color = (float)((double)total_color + 1.0 / (1.0 + (double)pexpf(-row[j])) * (double)g[j]);
                    }
                    const float att = pexpf(-s.delta_t * sigma * r.delta_scale);
                    const float weight = light * (1.f - att);
                    light *= att;
                    accum -= weight * total_color;
                    const float toadd = s.delta_t * r.delta_scale * (total_color * light - accum)
                                      + s.delta_t * r.delta_scale * g[C] * light_ray;
                    atomicAdd(grad + (int64_t)s.idx * gstride + (K - 1), toadd);
                }
            }
            t = march_advance(t, s.delta_t);
        }
    }
}

// Generic backward with shaped atomics: the same two steps as the specialised
// kernel -- a first march that only builds `accum`, a wave-synchronous second
// march that stages each sample's K gradient values in LDS and flushes rows
// cooperatively -- for any K <= 61, any format, component sub-range and per-leaf
// view rotation.  Staging lives in dynamic LDS: (kBlock/64) * 64 * (K|1) floats
// + kBlock row indices.
template <bool N2>
__global__ void __launch_bounds__(kBlock)
render_bwd_generic_staged_kernel(TreeDev tr, RaysDev rays, Opts opt, int C,
                                 const float* __restrict__ grad_out, float* __restrict__ grad, int gstride) {
    extern __shared__ float dyn_lds[];
    const int K = tr.K;
    const int KS = K | 1;
    const int lane = threadIdx.x & 63;
    float* stage = dyn_lds + (threadIdx.x >> 6) * (64 * KS);
    int32_t* sidx = reinterpret_cast<int32_t*>(dyn_lds + (kBlock / 64) * 64 * KS) + (threadIdx.x >> 6) * 64;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const int64_t q = ray_of_thread(rays, (int64_t)blockIdx.x * kBlock + threadIdx.x);
    Ray r;
    bool alive = q < rays.Q;
    if (alive) alive = setup_ray(tr, rays, opt, q, r);
    if (!__any(alive)) return;
    const float* g = grad_out + (alive ? q : 0) * (C + 1);
    float vd[3];
    load_vdir(rays, alive ? q : 0, vd);
    const bool rgba = opt.format == FMT_RGBA;
    float basis[25];       // basis of the current sample (re-evaluated per sample with view rotations)
    float basis2[25];      // basis pass 2 of the reference sees: the one pass 1 ended with
    float accum = 0.f, light_ray = 1.f;
    if (alive) {
        precalc_basis<0>(opt.format, opt.basis_dim, tr, vd[0], vd[1], vd[2], basis);
        float light = 1.f, t = r.tmin;
        while (t < r.tmax) {                               // march 1 (rt_kernel.cu:365-437 minus the atomics)
            Sample s;
            march_step<N2>(tr, r, opt.step_size, t, s);
            if (s.valid) {
                const float* row = tr.features + (int64_t)s.idx * K;
                const float sigma = row[K - 1];
                if (sigma > 0.f) {
                    if (tr.xform != nullptr) rotated_basis(tr, opt.format, opt.basis_dim, s.idx, vd, basis);
                    const float att = pexpf(-s.delta_t * sigma * r.delta_scale);
                    const float weight = light * (1.f - att);
                    float total_color = 0.f;
                    for (int c = 0; c < C; ++c) {
                        float x;
                        if (rgba) {
                            x = row[c];
                        } else {
                            x = 0.f;
                            for (int i = opt.min_comp; i <= opt.max_comp; ++i) x += basis[i] * row[c * opt.basis_dim + i];
                        }
                        total_color += (float)(1.0 / (1.0 + (double)pexpf(-x))) * g[c];
                    }
                    light *= att;
                    accum += weight * total_color;
                }
            }
            t = march_advance(t, s.delta_t);
        }
        float total_grad = 0.f;
        for (int j = 0; j < C; ++j) total_grad += g[j];
        accum += light * opt.background_brightness * total_grad;
        light_ray = light;
        for (int i = 0; i < 25; ++i) basis2[i] = basis[i];
    }
    // march 2, wave-synchronous (rt_kernel.cu:439-494 plus the colour terms of :408-425)
    float light = 1.f;
    float t = alive ? r.tmin : 0.f;
    const float tmax = alive ? r.tmax : -1.f;
    const int ROWS = (K <= 32) ? 2 : 1;
    const int LPR = 64 / ROWS;
    while (__any(t < tmax)) {
        bool active = false;
        int32_t idx = -1;
        Sample s;
        const float* row = nullptr;
        if (t < tmax) {
            march_step<N2>(tr, r, opt.step_size, t, s);
            t = march_advance(t, s.delta_t);
            if (s.valid) {
                row = tr.features + (int64_t)s.idx * K;
                if (row[K - 1] > 0.f) { active = true; idx = s.idx; }
            }
        }
        const unsigned long long amask = __ballot(active);
        if (amask == 0ull) continue;
        const int n = __popcll(amask);
        if (active) {
            const int slot = __popcll(amask & lane_lt);
            sidx[slot] = idx;
            float* st = stage + slot * KS;
            for (int j = 0; j < K - 1; ++j) st[j] = 0.f;       // columns outside the component range stay 0
            const float sigma = row[K - 1];
            if (tr.xform != nullptr) rotated_basis(tr, opt.format, opt.basis_dim, idx, vd, basis);
            const float att = pexpf(-s.delta_t * sigma * r.delta_scale);
            const float weight = light * (1.f - att);
            float total_color = 0.f;
            for (int c = 0; c < C; ++c) {
                if (rgba) {
                    const double sd = 1.0 / (1.0 + (double)pexpf(-row[c]));
                    const float sig = (float)sd;
                    st[c] = weight * sig * (1.f - sig) * g[c];
                    total_color = (float)((double)total_color + sd * (double)g[c]);
                } else {
                    const int off = c * opt.basis_dim;
                    float x = 0.f, x2 = 0.f;
                    for (int i = opt.min_comp; i <= opt.max_comp; ++i) {
                        x += basis[i] * row[off + i];
                        x2 += basis2[i] * row[off + i];
                    }
                    const float sig = (float)(1.0 / (1.0 + (double)pexpf(-x)));
                    const float gsig = (float)((double)sig * (1.0 - (double)sig));
                    for (int i = opt.min_comp; i <= opt.max_comp; ++i) st[off + i] = weight * basis[i] * gsig * g[c];
                    total_color = (float)((double)total_color + 1.0 / (1.0 + (double)pexpf(-x2)) * (double)g[c]);
                }
            }
            light *= att;
            accum -= weight * total_color;
            st[K - 1] = s.delta_t * r.delta_scale * (total_color * light - accum)
                      + s.delta_t * r.delta_scale * g[C] * light_ray;
        }
        // cooperative flush: contiguous K-float segments per atomic instruction
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int half = (ROWS == 2) ? (lane >> 5) : 0;
        const int j = (ROWS == 2) ? (lane & 31) : lane;
        for (int base = 0; base < n; base += ROWS) {
            const int rw = base + half;
            if (rw < n) {
                const int32_t ridx = sidx[rw];
                for (int col = j; col < K; col += LPR)
                    atomicAdd(grad + (int64_t)ridx * gstride + col, stage[rw * KS + col]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace svoxt
