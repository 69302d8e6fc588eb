// svoxt_kernels.hip -- hand-written CDNA4 (gfx950) kernels for svox_t's
// volume-render hot path and the C ABI declared in include/svoxt.h.
//
// Mapping: one ray per wavefront lane, one wavefront per workgroup (kBlock); a
// wavefront takes 64 consecutive rays, or an 8x8 pixel tile when the caller says
// the batch is an image.  The per-ray output accumulators live in registers for
// the specialised payloads (RGBA C=3 / C=31, SH with 1/4/9/16/25 basis functions
// x 3 channels; the SH ones also with per-leaf view rotations, XF) and in global
// memory only for the generic fallback (any K / SG / ASG / component sub-ranges /
// view rotations on other payloads), which mirrors the reference's
// read-modify-write of `out` (svox_t/csrc/rt_kernel.cu:300,304).  Rays come from
// tensors or, in camera mode, are generated per pixel (svoxt_device.h setup_ray).
//
// Kernels, in file order:
//   render_fwd_kernel          trace_ray; optionally records each ray's composited samples
//   render_fwd_generic_kernel  fallback forward
//   render_bwd_kernel          trace_ray_backward: replays recorded samples (or marches),
//                              stages gradient rows in LDS, flushes them as shaped atomics;
//                              <..., GATHER>: list walk of the two-kernel backward / tail-only launch
//   grad_merge_kernel          second kernel of the two-kernel backward (per-tile merge in LDS)
//   grad_fused_kernel          list walk + per-tile merge as one kernel: the backward of an image
//   render_bwd_generic_kernel  fallback backward, per-lane atomics (opacity backward, K > 64)
//   render_bwd_generic_staged_kernel  fallback backward with LDS-staged, shaped atomics
//   opacity_fwd_kernel (+ opacity_walk_kernel, opacity_merge_kernel: backward from lists),
//   depth_kernel, count_fwd_kernel
//   query_fwd_kernel, query_bwd_kernel, leaves_count / scan / scatter kernels
//   compact_rows_kernel, accel_build_kernel, accel_nodes_kernel
// Other translation units of the library: svoxt_build.hip (octree from a point
// cloud, construct_tree), svoxt_motion.hip (motion variants, point skinning), svoxt_order.hip
// (coherent order for ray batches that are not images).
// The design rationale and the measurements behind each choice are in DESIGN.md 5.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see build.py).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/svoxt.h"
#include "svoxt_device.h"
#include "svoxt_host.h"

#pragma clang fp contract(off)

namespace svoxt {

// One wavefront per workgroup: the finest scheduling granularity for kernels whose
// wavefronts differ 10x in cost (measured: 64 -> 492, 128 -> 485, 256 -> 477,
// 512 -> 466 Mrays/s on the headline workload).
constexpr int kBlock = 64;

// ---------------------------------------------------------------------------
// Forward: trace_ray (rt_kernel.cu:222-328) + render_ray_kernel (:655-671)
// ---------------------------------------------------------------------------

// FMT: FMT_RGBA or FMT_SH (specialised);  C: colour channels;  BD: basis dim.
// REC: also record every composited sample as (feature row, delta_t) in
// rec[k][q] (k < S) and, per ray, aux[q] = {count | overflow << 31, t at which
// the first unrecorded sample starts}, for svoxt_volume_render_bwd_replay.
// REC requires sigma_thresh == stop_thresh == 0 (the backward ignores both,
// rt_kernel.cu:382,456); the march is then not cut short when the
// transmittance underflows to exactly 0 -- the remaining samples have weight
// 0 and leave the output bits unchanged, but they belong in the list.
constexpr uint32_t kRecOverflow = 0x80000000u;

// Records are written once and read once or twice, much later: non-temporal
// accesses keep them from displacing the tree and the feature table in L2 /
// Infinity Cache (measured: forward 0.42 -> 0.38 ms).
__device__ __forceinline__ void rec_put(uint2* p, uint32_t idx, float delta_t) {
    const unsigned long long v = (unsigned long long)idx | ((unsigned long long)__float_as_uint(delta_t) << 32);
    __builtin_nontemporal_store(v, reinterpret_cast<unsigned long long*>(p));
}
__device__ __forceinline__ uint2 rec_get(const uint2* p) {
    const unsigned long long v = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(p));
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

// Where record k of the ray handled by launch thread `tid` (tile tid >> 6, lane tid & 63) lives:
// rec[tile][k / 8][lane][k % 8] -- the 8 records of a block are the lane's own 64-byte line.
// (Round 1 kept rec[k][q]: every record a lone 8-byte store to a different line.  A wavefront's
// vector-memory operations complete in order -- stores count in vmcnt -- so each of those stores
// sat in front of the next tree word of the march: the record stores, not the loads, were what
// made a crossing cost 2.5 us under load (r02: halving the records took the march kernel from
// 0.227 to 0.178 ms, removing the sigma gather changed nothing), and they reached memory as
// partial lines, 2.5x write amplification.  Now a lane stages 8 records in LDS and writes one
// whole line per 8 records.)
constexpr int kRecBlock = 8;
constexpr int kMaxRecBlocks = 64;        // blocks per ray at most: max_samples <= 512 with a block table

// The lists as the kernels see them.  A BLOCK is the 8 consecutive records of the 64 rays of a tile
// (4 KB: 64 lines of 64 bytes).  Dense (tab == NULL): block b of tile T is block T * (S / 8) + b of
// `rec` -- every ray owns S slots.  Pooled (r02; tab != NULL): tab[T * (S / 8) + b] names the
// block, handed out from `rec`'s pool_blocks blocks by a counter the first time a ray of the tile
// starts it (-1: never): memory follows the samples that exist (mean 9 per ray on the headline
// workload against a cap of 96), S only caps a ray.
struct RecLists {
    uint2* __restrict__ rec;
    int32_t* __restrict__ tab;
    int32_t* __restrict__ pool_next;
    int64_t pool_blocks;
    int S;
    // optional: 16 bytes per record slot, (att, e_0, e_1, e_2) of a 3-channel sample as the BACKWARD
    // needs them -- att = exp(-delta_t * sigma * delta_scale) in the backward's association
    // (rt_kernel.cu:397), e_c = exp(-x_c).  Layout [block][k / 4 % 2][lane][k % 4]: the 4 consecutive
    // entries of a ray are one 64-byte line (written whole by the recording forward, terms_index).
    float4* __restrict__ terms;
};
__device__ __forceinline__ int64_t terms_index(int64_t block, int lane, int k) {
    return ((((block << 1) + ((k >> 2) & 1)) << 6) + lane) * 4 + (k & 3);
}
// ... or position-major, [block][k % 8][lane]: what kernels that hold one list position of the 64 rays
// of a tile per wavefront write and read as 1 KB at a time (shade_tile_kernel, grad_fused_kernel)
__device__ __forceinline__ int64_t terms_index_pm(int64_t block, int lane, int k) {
    return ((block << 3) + (k & 7)) * 64 + lane;
}

__device__ __forceinline__ int64_t rec_block(const RecLists& L, int64_t tile, int b) {
    const int64_t e = tile * (int64_t)(L.S >> 3) + b;
    return L.tab != nullptr ? (int64_t)L.tab[e] : e;
}
// Kernels whose wavefronts each work on ONE tile and one block at a time keep the tile's table in a
// register -- lane b holds block b -- and read it with readlane: no table load in front of every
// record load (r02: the per-tile backward lost 0.02 ms to exactly that).
__device__ __forceinline__ int32_t rec_tab_reg(const RecLists& L, int64_t tile, int lane) {
    const int nb = L.S >> 3;
    return (L.tab != nullptr && lane < nb) ? L.tab[tile * (int64_t)nb + lane] : -1;
}
__device__ __forceinline__ int64_t rec_block_u(const RecLists& L, int32_t tabreg, int64_t tile, int b /* wavefront-uniform */) {
    if (L.tab == nullptr) return tile * (int64_t)(L.S >> 3) + b;
    return (int64_t)__builtin_amdgcn_readlane(tabreg, __builtin_amdgcn_readfirstlane(b));
}
// The pool is cut into kSubPools equal parts with a counter each (64 bytes apart), chosen by the tile:
// one counter for every hand-out was a single hot address -- 40 000 returning atomics per forward
// of the headline workload, 0.25 -> 0.31 ms (r02).
constexpr int kSubPools = 32;
constexpr int kSubPoolStride = 16;       // int32 between two counters
// where record k of the ray handled by launch thread `tid` (tile tid >> 6, lane tid & 63) lives
__device__ __forceinline__ int64_t rec_index(const RecLists& L, int64_t tid, int k) {
    return (((rec_block(L, tid >> 6, k >> 3) << 6) + (tid & 63)) << 3) + (k & 7);
}
__device__ __forceinline__ int64_t rec_index_in(int64_t block, int64_t tid, int k) {
    return (((block << 6) + (tid & 63)) << 3) + (k & 7);
}

// Writers (one wavefront per workgroup = one tile).  ltab: the tile's block table in LDS
// ([kMaxRecBlocks], -1 = not handed out yet; rec_tab_init).  rec_block_begin is called by the lanes
// that are about to write the FIRST record of block b (a divergent subset of the wavefront, possibly
// with different b): the block is taken from the table, or a leader among them takes one from the
// pool for all.  Returns -2 when the pool is used up (the ray's list then counts as full).
__device__ __forceinline__ void rec_tab_init(int32_t* ltab) {
    if (threadIdx.x < kMaxRecBlocks) ltab[threadIdx.x] = -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int64_t rec_block_begin(const RecLists& L, int32_t* ltab, int64_t tile, int b) {
    if (L.tab == nullptr) return tile * (int64_t)(L.S >> 3) + b;
    int id = ltab[b];
    while (true) {
        const unsigned long long m = __ballot(id == -1);          // lanes here whose block is not handed out yet
        if (m == 0ull) break;
        const int leader = __ffsll((long long)m) - 1;
        const int bl = __shfl(b, leader, 64);
        int nid = 0;
        if ((int)(threadIdx.x & 63) == leader) {
            const int sp = (int)(tile & (kSubPools - 1));
            const int64_t per = L.pool_blocks / kSubPools;          // blocks of one part
            nid = atomicAdd(L.pool_next + sp * kSubPoolStride, 1) + 1;     // the counters start at -1, like the table
            nid = (int64_t)nid < per ? (int)(sp * per + nid) : -2;
            ltab[bl] = nid;
            if (nid >= 0) L.tab[tile * (int64_t)(L.S >> 3) + bl] = nid;
        }
        nid = __shfl(nid, leader, 64);
        if (b == bl) id = nid;
    }
    return (int64_t)id;
}

// the staging buffer of one wavefront: [8][64] records, lane-contiguous (conflict-free ds_write_b64)
__device__ __forceinline__ void rec_stage_flush(const uint2* __restrict__ lds, int lane, uint2* __restrict__ rec,
                                                int64_t block) {
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    v4u* dst = reinterpret_cast<v4u*>(rec + (((block << 6) + lane) << 3));      // the lane's 64-byte line of the block
#pragma unroll
    for (int j = 0; j < kRecBlock / 2; ++j) {
        const uint2 a = lds[(2 * j) * 64 + lane], b = lds[(2 * j + 1) * 64 + lane];
        __builtin_nontemporal_store(v4u{a.x, a.y, b.x, b.y}, dst + j);
    }
}
// record number k (the k-th of this ray) <- (feature row, delta_t), into the block rec_block_begin
// gave for k's block; a full block goes out as one line
__device__ __forceinline__ void rec_stage_put(uint2* __restrict__ lds, int lane, uint2* __restrict__ rec, int64_t block,
                                              int k, uint32_t idx, float delta_t) {
    lds[(k & 7) * 64 + lane] = make_uint2(idx, __float_as_uint(delta_t));
    if ((k & 7) == 7) rec_stage_flush(lds, lane, rec, block);
}
// at the end of a ray with nrec records: the partly filled last block (its unused slots carry stale values)
__device__ __forceinline__ void rec_stage_finish(const uint2* __restrict__ lds, int lane, uint2* __restrict__ rec,
                                                 int64_t block, int nrec) {
    if (nrec & 7) rec_stage_flush(lds, lane, rec, block);
}

// the lane's line of four staged (att, e_0, e_1, e_2) entries, the one that holds list position k
__device__ __forceinline__ void terms_flush(const float4* __restrict__ lds, int lane, float4* __restrict__ terms,
                                            int64_t block, int k) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f* dst = reinterpret_cast<v4f*>(terms + terms_index(block, lane, k & ~3));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 v = lds[j * 64 + lane];
        __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, dst + j);
    }
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

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
template <int FMT, int C, int BD, bool N2, bool REC, bool XF = false, bool RESUME = false>
__global__ void __launch_bounds__(kBlock)
render_fwd_kernel(TreeDev tr, RaysDev rays, Opts opt, float* __restrict__ out,
                  RecLists L, uint4* __restrict__ aux) {
    static_assert(!XF || FMT == FMT_SH, "view rotations only matter for view-dependent formats");
    static_assert(!(RESUME && REC), "the tail launch does not record");
    constexpr int K = (FMT == FMT_RGBA) ? (C + 1) : (C * BD + 1);
    __shared__ uint2 rstage[REC ? kRecBlock * kBlock : 1];
    __shared__ int32_t ltab[REC ? kMaxRecBlocks : 1];
    // (att, e_0, e_1, e_2) of the last <= 4 recorded samples of each ray, for the backward (C == 3)
    __shared__ float4 tstage[(REC && C == 3 && !XF) ? 4 * kBlock : 1];
    if constexpr (REC) rec_tab_init(ltab);
    const int S = L.S;
    int64_t cur_block = 0;
    const int64_t tid = ((int64_t)blockIdx.x + rays.tile0) * kBlock + threadIdx.x;
    const int64_t q = ray_of_thread(rays, tid);
    if (q >= rays.Q) return;
    float* o = out + q * (C + 1);
    float t_start = 0.f;
    if constexpr (RESUME) {
        const uint4 a = aux[q];
        if ((a.x & kRecOverflow) == 0u) return;
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
        if constexpr (!XF) precalc_basis<BD>(FMT_SH, BD, tr, vd[0], vd[1], vd[2], basis);
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
    auto shade = [&](const float (&row)[K], int32_t idx, float delta_t, float t_cur, uint32_t slot) -> bool {
        const float sigma = row[K - 1];
        if (!(sigma > opt.sigma_thresh)) return false;
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
                float tmp = 0.f;
#pragma unroll
                for (int i = 0; i < BD; ++i) tmp += basis[i] * row[c * BD + i];
                ex[c] = pexpf(-tmp);
                acc[c] = (float)((double)acc[c] + (double)weight / (1.0 + (double)ex[c]));
            }
        } else {
#pragma unroll
            for (int j = 0; j < C; ++j) {
                ex[j] = pexpf(-row[j]);
                acc[j] = (float)((double)acc[j] + (double)weight / (1.0 + (double)ex[j]));
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
        light *= att;
        if (tr.weight_accum != nullptr) atomicAdd(tr.weight_accum + leaf_slot<N2>(tr, r, t_cur, slot), weight);
        if constexpr (!REC) {
            if (light <= opt.stop_thresh) return true;
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
    }
    if constexpr (RESUME) aux[q].z = __float_as_uint(light);
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
// One bit per feature row: sigma > thresh (svoxt_sigma_mask_build).  A wavefront's 64 rows are one 8-byte word.
__global__ void __launch_bounds__(256)
sigma_mask_kernel(const float* __restrict__ features, int64_t M, int K, float thresh, unsigned long long* __restrict__ mask) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool on = row < M && features[row * K + (K - 1)] > thresh;
    const unsigned long long b = __ballot(on);
    if ((threadIdx.x & 63) == 0 && (row >> 6) < (M + 63) / 64) mask[row >> 6] = b;
}

// MASK (no stop rule): whether a row's sigma exceeds sigma_thresh comes from one bit per feature row
// (svoxt_sigma_mask_build: M / 8 bytes, resident in L2) instead of a 4-byte gather that pulls a
// 64-byte line of the feature table -- half of this kernel's traffic, and HBM traffic once the table
// has left the Infinity Cache (r02, depth 9 / 32-float rows: forward 1.29 -> 1.15 ms with no gather at all).
template <bool N2, bool STOP, int ACC, bool MASK = false>
__global__ void __launch_bounds__(kBlock)
march_rec_kernel(TreeDev tr, RaysDev rays, Opts opt, RecLists L, uint4* __restrict__ aux,
                 const uint32_t* __restrict__ sigma_mask = nullptr) {
    static_assert(!(MASK && STOP), "the stop rule needs sigma itself");
    __shared__ uint2 rstage[kRecBlock * kBlock];
    __shared__ int32_t ltab[kMaxRecBlocks];
    rec_tab_init(ltab);
    const int S = L.S;
    int64_t cur_block = 0;
    const int64_t tid = ((int64_t)blockIdx.x + rays.tile0) * kBlock + threadIdx.x;
    const int64_t q = ray_of_thread(rays, tid);
    if (q >= rays.Q) return;
    Ray r;
    if (!setup_ray(tr, rays, opt, q, r)) {
        aux[q] = make_uint4(0u, 0u, __float_as_uint(1.f), 0u);
        return;
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
    const float kNone = -__builtin_inff();
    float p_sigma = kNone, p_dt = 0.f, p_t = 0.f;
    int32_t p_idx = 0;
    while (t < r.tmax) {
        Sample s;
        march_step<N2, ACC>(tr, r, opt.step_size, t, s);
        const float t_cur = t;
        t = march_advance(t, s.delta_t);
        bool keep = true;
        if (p_sigma > opt.sigma_thresh) {
            bool room = nrec < S;
            if (room && (nrec & 7) == 0) {
                cur_block = rec_block_begin(L, ltab, tid >> 6, nrec >> 3);
                room = cur_block >= 0;
            }
            if (room) {
                rec_stage_put(rstage, (int)threadIdx.x, L.rec, cur_block, nrec, (uint32_t)p_idx, p_dt);
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
        if (keep && s.valid) {
            if constexpr (MASK) p_sigma = ((sigma_mask[s.idx >> 5] >> (s.idx & 31)) & 1u) ? __builtin_inff() : kNone;
            else p_sigma = sig_col[(int64_t)s.idx * K];
            p_idx = s.idx;
            p_dt = s.delta_t;
            p_t = t_cur;
        }
    }
    if (p_sigma > opt.sigma_thresh) {    // the last crossing's sample
        bool room = nrec < S;
        if (room && (nrec & 7) == 0) {
            cur_block = rec_block_begin(L, ltab, tid >> 6, nrec >> 3);
            room = cur_block >= 0;
        }
        if (room) {
            rec_stage_put(rstage, (int)threadIdx.x, L.rec, cur_block, nrec, (uint32_t)p_idx, p_dt);
            ++nrec;
        } else {
            over = kRecOverflow;
            t_resume = p_t;
        }
    }
    rec_stage_finish(rstage, (int)threadIdx.x, L.rec, cur_block, nrec);
    aux[q] = make_uint4((uint32_t)nrec | over, __float_as_uint(t_resume), __float_as_uint(1.f), 0u);
}

// WTERMS (recording forwards, no view rotations): the wavefronts that form a sample's exponentials also
// leave them, with the attenuation in the backward's association (rt_kernel.cu:397), in L.terms
// (position-major: 1 KB per wavefront and list position) for the exact backward.
template <int FMT, int BD, bool XF, bool STOP, bool WTERMS = false>
__global__ void __launch_bounds__(512)
shade_tile_kernel(TreeDev tr, RaysDev rays, Opts opt, RecLists L,
                  uint4* __restrict__ aux, float* __restrict__ out) {
    constexpr int C = 3, W = 8, P = W - 1;
    constexpr int K = (FMT == FMT_RGBA) ? (C + 1) : (C * BD + 1);
    constexpr int NB = (FMT == FMT_SH) ? BD : 1;
    typedef float v4f __attribute__((ext_vector_type(4)));
    __shared__ v4f terms[2][P][64];              // (att, e_0, e_1, e_2) of a list position, per ray
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x + rays.tile0;
    const int64_t q = ray_of_thread(rays, tile * 64 + lane);
    const bool inb = q < rays.Q;
    uint4 a = make_uint4(0u, 0u, 0u, 0u);
    if (inb) a = aux[q];
    const int nrec = (int)(a.x & ~kRecOverflow);
    int maxn = nrec;
    for (int off = 32; off > 0; off >>= 1) maxn = max(maxn, __shfl_xor(maxn, off, 64));
    maxn = __builtin_amdgcn_readfirstlane(maxn);     // what decides the barrier count is scalar
    const int nround = (maxn + P - 1) / P;           // the same in every wavefront of the workgroup
    const int32_t tabreg = rec_tab_reg(L, tile, lane);

    float delta_scale = 0.f;
    float basis[NB];
    float vd[3] = {0.f, 0.f, 0.f};
    if (wave > 0 && nrec > 0) {
        Ray r;
        setup_ray(tr, rays, opt, q, r);               // a ray with samples hits the cube
        delta_scale = r.delta_scale;
        if constexpr (FMT == FMT_SH) {
            load_vdir(rays, q, vd);
            if constexpr (!XF) precalc_basis<BD>(FMT_SH, BD, tr, vd[0], vd[1], vd[2], basis);
        }
    }
    float light = 1.f, acc[C] = {0.f, 0.f, 0.f};
    bool stopped = false;

    for (int rd = 0; rd <= nround; ++rd) {
        if (wave > 0) {
            const int k = rd * P + (wave - 1);
            if (rd < nround && k < nrec) {
                const int64_t blk = rec_block_u(L, tabreg, tile, k >> 3);
                const uint2 e = rec_get(L.rec + rec_index_in(blk, lane, k));
                const int32_t idx = (int32_t)e.x;
                float row[K];
                load_row<K>(tr.features + (int64_t)idx * K, row);
                v4f tv;
                tv.x = pexpf(-__uint_as_float(e.y) * delta_scale * row[K - 1]);
                if constexpr (FMT == FMT_SH) {
                    if constexpr (XF) rotated_sh_basis<BD>(tr, idx, vd, basis);
                    float ex[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        float tmp = 0.f;
#pragma unroll
                        for (int i = 0; i < BD; ++i) tmp += basis[i] * row[c * BD + i];
                        ex[c] = pexpf(-tmp);
                    }
                    tv.y = ex[0]; tv.z = ex[1]; tv.w = ex[2];
                } else {
                    tv.y = pexpf(-row[0]); tv.z = pexpf(-row[1]); tv.w = pexpf(-row[2]);
                }
                terms[rd & 1][wave - 1][lane] = tv;
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
                    if constexpr (STOP) {
                        if (light <= opt.stop_thresh) stopped = true;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (wave == 0 && inb) {
        float* o = out + q * (C + 1);
        if (stopped) {
            const float scale = (float)(1.0 / (1.0 - (double)light));
#pragma unroll
            for (int j = 0; j < C; ++j) o[j] = acc[j] * scale;
            o[C] = 1.f - light;
            a.x &= ~kRecOverflow;                    // nothing left for the tail launch
        } else if (a.x & kRecOverflow) {             // state for render_fwd_kernel<..., RESUME>
#pragma unroll
            for (int j = 0; j < C; ++j) o[j] = acc[j];
            o[C] = light;
        } else {
            const float bg = light * opt.background_brightness;
#pragma unroll
            for (int j = 0; j < C; ++j) o[j] = acc[j] + bg;
            o[C] = 1.f - light;
        }
        a.z = __float_as_uint(light);                // the final transmittance, for the single-march backward
        aux[q] = a;
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
// FAST (opt-in tolerance mode): the quotient in float with the hardware reciprocal,
// acc += w * rcp(1 + e): each term within 2e-7 of the reference's double-precision quotient.
template <int K, bool STOP, bool FAST>
__global__ void __launch_bounds__(256)
shade_chan_kernel(TreeDev tr, RaysDev rays, Opts opt, RecLists L,
                  uint4* __restrict__ aux, float* __restrict__ out) {
    static_assert(K == 8 || K == 16 || K == 32, "row widths with a channel-lane instance");
    constexpr int RPW = 64 / K;                                  // rays per wavefront
    const int lane = threadIdx.x & 63;
    const int c = lane & (K - 1);
    const int sig_lane = lane | (K - 1);                         // the sigma lane of this lane's ray
    // t: the launch thread of march_rec_kernel that holds this ray (tile t >> 6, lane t & 63)
    const int64_t t = (int64_t)rays.tile0 * 64 + ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RPW + (lane / K);
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
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    for (int kb = 0; kb < maxn; kb += kRecBlock) {
        uint32_t idx[kRecBlock];
        float dt[kRecBlock], ex[kRecBlock];
        const int n_here = min(nrec - kb, kRecBlock);            // records of this ray in the block (<= 0: none)
        const int64_t blk = rec_block_u(L, tabreg, t >> 6, kb >> 3);
        if (n_here > 0) {
            const v4u* line = reinterpret_cast<const v4u*>(L.rec + rec_index_in(blk, t, kb));   // the ray's line of this block
#pragma unroll
            for (int j = 0; j < kRecBlock / 2; ++j) {
                const v4u w = __builtin_nontemporal_load(line + j);
                idx[2 * j] = w.x; dt[2 * j] = __uint_as_float(w.y);
                idx[2 * j + 1] = w.z; dt[2 * j + 1] = __uint_as_float(w.w);
            }
        }
        float x[kRecBlock];
#pragma unroll
        for (int j = 0; j < kRecBlock; ++j) {
            x[j] = 0.f;
            if (j < n_here) x[j] = tr.features[(int64_t)(int32_t)idx[j] * K + c];
        }
#pragma unroll
        for (int j = 0; j < kRecBlock; ++j) {
            ex[j] = 1.f;
            if (j < n_here) ex[j] = pexpf(c == K - 1 ? -dt[j] * ds * x[j] : -x[j]);
        }
#pragma unroll
        for (int j = 0; j < kRecBlock; ++j) {
            const float att = __shfl(ex[j], sig_lane, 64);       // every lane takes part
            if (j < n_here && !stopped) {
                const float weight = light * (1.f - att);
                if constexpr (FAST) acc += weight * __builtin_amdgcn_rcpf(1.f + ex[j]);
                else acc = (float)((double)acc + (double)weight / (1.0 + (double)ex[j]));
                light *= att;
                if constexpr (STOP) {
                    if (light <= opt.stop_thresh) stopped = true;
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
        if (stopped) a.x &= ~kRecOverflow;
        a.z = __float_as_uint(light);
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
tail_chan_kernel(TreeDev tr, RaysDev rays, Opts opt, uint4* __restrict__ aux, float* __restrict__ out) {
    // One workgroup (four wavefronts) per 64-ray tile; with lists that hold every sample -- the
    // usual case since they are pooled -- the launch is one look at the tile's 64 aux entries.
    constexpr int RPW = 64 / K;
    __shared__ int any_over;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x + rays.tile0;
    if (wave == 0) {
        const int64_t q0 = ray_of_thread(rays, tile * 64 + lane);
        const bool ov = q0 < rays.Q && (aux[q0].x & kRecOverflow) != 0u;
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
        bool alive = (a.x & kRecOverflow) != 0u;
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
            if (active) ex = pexpf(c == K - 1 ? -dt * r.delta_scale * x : -x);
            const float att = __shfl(ex, sig_lane, 64);
            if (active) {
                const float weight = light * (1.f - att);
                if constexpr (FAST) acc += weight * __builtin_amdgcn_rcpf(1.f + ex);
                else acc = (float)((double)acc + (double)weight / (1.0 + (double)ex));
                light *= att;
                if (light <= opt.stop_thresh) { stopped = true; alive = false; }
            }
        }
        if ((a.x & kRecOverflow) == 0u) continue;
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
template <int FMT, int C, int BD, bool N2, bool REPLAY, bool XF = false, bool GATHER = false, bool ONEPASS = false>
__global__ void __launch_bounds__(kBlock, (GATHER && !XF && C == 3) ? 4 : 1)
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

    const int lane = threadIdx.x & 63;
    float* stage = stage_all + (threadIdx.x >> 6) * (64 * KS);
    int32_t* sidx = sidx_all + (threadIdx.x >> 6) * 64;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t q = ray_of_thread(rays, tid);
    Ray r;
    bool alive = q < rays.Q;
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
            precalc_basis<BD>(FMT_SH, BD, tr, vd[0], vd[1], vd[2], basis);
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
// TERMS (EXACT only): 1 = sweep 1 hands (att, e_0, e_1, e_2) of every sample to sweep 2 through
// L.terms (position-major); 2 / 3 = the recording forward left them there (render_fwd_kernel:
// lane-major lines; shade_tile_kernel: position-major): neither sweep gathers a feature row or forms
// an exponential.
// Two workgroups per CU (77 KB of LDS each) need at most 128 registers: said to the compiler,
// because one branch too many costs exactly that (r02: 116 -> 130 registers, 0.38 -> 0.55 ms).
template <int FMT, int BD, bool EXACT, bool COUNT = false, int TERMS = 0>
__global__ void __launch_bounds__(512, 4)
grad_fused_kernel(TreeDev tr, RaysDev rays, Opts opt, const float* __restrict__ grad_out,
                  RecLists L, const uint4* __restrict__ aux, const float* __restrict__ fwd_out,
                  float* __restrict__ grad, int gstride, unsigned long long* __restrict__ counters = nullptr) {
    float4* __restrict__ terms = L.terms;
    constexpr int C = 3, W = 8, NT = 64 * W, T = 1024, R = 1024, RPP = R / (64 * W);
    constexpr int K = (FMT == FMT_RGBA) ? (C + 1) : (C * BD + 1);
    constexpr int NB = (FMT == FMT_SH) ? BD : 0;
    constexpr int BDS = (FMT == FMT_SH) ? (BD | 1) : 1;
    constexpr int HALF = K < 16 ? K : 16;                    // columns 0-15 / 16-K: see grad_merge_kernel
    constexpr int KS = HALF | 1;                             // staging row: one round of columns
    static_assert(K <= 32 && RPP == 2, "sizes");
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* stage = stage_all + wave * 64 * KS;
    int32_t* seg = seg_all + wave * 64;
    const int32_t tabreg = rec_tab_reg(L, blockIdx.x, lane);
    const int64_t q = ray_of_thread(rays, (int64_t)blockIdx.x * 64 + lane);
    uint4 a = make_uint4(0u, 0u, 0u, 0u);
    if (q < rays.Q) a = aux[q];
    const int nrec = (int)(a.x & ~kRecOverflow);
    int maxn = nrec;
    for (int off = 32; off > 0; off >>= 1) maxn = max(maxn, __shfl_xor(maxn, off, 64));
    maxn = __builtin_amdgcn_readfirstlane(maxn);             // everything that decides barriers is scalar
    if (maxn == 0) return;                                   // the same in every wavefront of the workgroup

    Ray r;
    float basis[NB > 0 ? NB : 1], g[C + 1];
    float accum = 0.f, light = 1.f;
    float light_ray = __uint_as_float(a.z);
    r.delta_scale = 0.f;
#pragma unroll
    for (int j = 0; j <= C; ++j) g[j] = 0.f;
    if (nrec > 0) {
        setup_ray(tr, rays, opt, q, r);                      // for delta_scale (a ray with samples hits the cube)
        if constexpr (FMT == FMT_SH) {
            float vd[3];
            load_vdir(rays, q, vd);
            precalc_basis<BD>(FMT_SH, BD, tr, vd[0], vd[1], vd[2], basis);
        }
#pragma unroll
        for (int j = 0; j <= C; ++j) g[j] = grad_out[q * (C + 1) + j];
        if (wave == 0) {
            if constexpr (!EXACT) {
                const float* o = fwd_out + q * (C + 1);      // see render_bwd_kernel: single march
#pragma unroll
                for (int c = 0; c < C; ++c) accum += g[c] * o[c];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) bases[lane * BDS + i] = basis[i];
#pragma unroll
            for (int c = 0; c < C; ++c) gl[lane * 3 + c] = g[c];
        }
    }
    for (int i = threadIdx.x; i < T; i += NT) { keys[i] = -1; cnt[i] = 0; }
    for (int i = threadIdx.x; i < R; i += NT) r_sl[i] = 0xffffffffu;
    if constexpr (EXACT) {
        // ---- sweep 1: pass 1 of the reference without its atomics (accum_sample), W positions per
        // round; r_w / r_sg carry (att, total_color) from the wavefront that formed them to wavefront 0,
        // double-buffered by the parity of the round (one barrier per round)
        float light1 = 1.f;
        const int nr1 = (maxn + W - 1) / W;                  // the same in every wavefront
        for (int rd = 0; rd <= nr1; ++rd) {
            const int k = rd * W + wave;
            if (rd < nr1 && k < nrec) {
                const int64_t blk = rec_block_u(L, tabreg, blockIdx.x, k >> 3);
                float att, ex[C];                             // exp(-x_c): sigmoid_d(x) = 1.0 / (1.0 + double(exp(-x)))
                if constexpr (TERMS >= 2) {
                    const float4 tv = terms[TERMS == 2 ? terms_index(blk, lane, k) : terms_index_pm(blk, lane, k)];
                    att = tv.x; ex[0] = tv.y; ex[1] = tv.z; ex[2] = tv.w;
                } else {
                    const uint2 e = rec_get(L.rec + rec_index_in(blk, lane, k));
                    float row[K];
                    load_row<K>(tr.features + (int64_t)(int32_t)e.x * K, row);
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
                    // sweep 2 needs the same attenuation and the same three exponentials: 16 bytes per
                    // sample instead of gathering the row and forming them again
                    if constexpr (TERMS == 1) terms[terms_index_pm(blk, lane, k)] = make_float4(att, ex[0], ex[1], ex[2]);
                }
                float total_color = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) total_color += (float)(1.0 / (1.0 + (double)ex[c])) * g[c];
                const int sl = ((rd & 1) * W + wave) * 64 + lane;
                r_w[sl] = att; r_sg[sl] = total_color;
            }
            if (wave == 0 && rd > 0) {
                // (the operands of the round's eight positions first, then what depends on the step before)
                float av[W], tv[W];
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const int sl = (((rd - 1) & 1) * W + j) * 64 + lane;
                    av[j] = r_w[sl]; tv[j] = r_sg[sl];
                }
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    if ((rd - 1) * W + j < nrec) {
                        const float weight = light1 * (1.f - av[j]);
                        light1 *= av[j];
                        accum += weight * tv[j];
                    }
                }
            }
            __syncthreads();
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

    for (int k0 = 0; k0 < maxn; k0 += RPP * W) {
        // ---- terms + advance, RPP rounds of W list positions
#pragma unroll 1
        for (int rd = 0; rd < RPP; ++rd) {
            const int kb = k0 + rd * W;                      // the same in every wavefront
            if (kb >= maxn) break;
            const int k = kb + wave;
            const int slot = (rd * W + wave) * 64 + lane;
            if (k < nrec) {
                const int64_t blk = rec_block_u(L, tabreg, blockIdx.x, k >> 3);
                const uint2 e = rec_get(L.rec + rec_index_in(blk, lane, k));
                float att, tc, cf[C];
                if constexpr (EXACT && TERMS != 0) {
                    const float4 tv = terms[TERMS == 2 ? terms_index(blk, lane, k) : terms_index_pm(blk, lane, k)];
                    const float ex[C] = {tv.y, tv.z, tv.w};
                    att = tv.x;
                    tc = 0.f;
#pragma unroll
                    for (int c = 0; c < C; ++c) {             // sample_terms from here on, operation for operation
                        const double sd = 1.0 / (1.0 + (double)ex[c]);
                        if constexpr (FMT == FMT_SH) {
                            const float sig = (float)sd;
                            cf[c] = (float)((double)sig * (1.0 - (double)sig));
                        } else {
                            cf[c] = (float)sd;
                        }
                        tc = (float)((double)tc + sd * (double)g[c]);
                    }
                } else {
                    float row[K];
                    load_row<K>(tr.features + (int64_t)(int32_t)e.x * K, row);
                    sample_terms<FMT, C, BD, K>(row, basis, g, __uint_as_float(e.y), r.delta_scale, att, tc, cf);
                }
                const int32_t idx = (int32_t)e.x;
                uint32_t h = ((uint32_t)idx * 0x9E3779B1u) >> (32 - __builtin_ctz(T));
                while (true) {
                    const int32_t old = atomicCAS(keys + h, -1, idx);
                    if (old == -1 || old == idx) break;
                    h = (h + 1u) & (uint32_t)(T - 1);
                }
                atomicAdd(cnt + h, 1);
                r_sl[slot] = (h << 6) | (uint32_t)lane;
                r_w[slot] = att; r_sg[slot] = tc; r_dt[slot] = __uint_as_float(e.y);
                r_c[slot] = cf[0]; r_c[R + slot] = cf[1]; r_c[2 * R + slot] = cf[2];
            }
            __syncthreads();
            if (wave == 0) {
                float av[W], tv[W], dv[W];
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const int s2 = (rd * W + j) * 64 + lane;
                    av[j] = r_w[s2]; tv[j] = r_sg[s2]; dv[j] = r_dt[s2];
                }
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    if (kb + j < nrec) {
                        const int s2 = (rd * W + j) * 64 + lane;
                        float cf[C] = {0.f, 0.f, 0.f};
                        if constexpr (FMT == FMT_RGBA) { cf[0] = r_c[s2]; cf[1] = r_c[R + s2]; cf[2] = r_c[2 * R + s2]; }
                        float wgt, sg;
                        sample_advance<FMT, C>(av[j], tv[j], cf, g, dv[j], r.delta_scale, light_ray, light, accum, wgt, sg);
                        r_w[s2] = wgt; r_sg[s2] = sg;
                        if constexpr (FMT == FMT_RGBA) { r_c[s2] = cf[0]; r_c[R + s2] = cf[1]; r_c[2 * R + s2] = cf[2]; }
                    }
                }
            }
        }
        // ---- sort: exclusive scan of the counters -- by wavefront 1, while wavefront 0 still advances
        // the last round (the counters are complete since the round's barrier) -- then the
        // counting sort of the record numbers
        if (wave == 1) {
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
            if (lane == 63) s_nb = incl;
        }
        __syncthreads();
        const int nb = __builtin_amdgcn_readfirstlane(s_nb);
        if constexpr (COUNT) {          // distinct feature rows of this pass = occupied table slots
            unsigned long long rows = 0;
            for (int i = threadIdx.x; i < T; i += NT) rows += keys[i] >= 0 ? 1u : 0u;
            rows = wave_sum(rows);
            if (lane == 0 && rows) atomicAdd(counters + 1, rows);
        }
        for (int rr = threadIdx.x; rr < R; rr += NT) {
            const uint32_t v = r_sl[rr];
            if (v != 0xffffffffu) order[atomicAdd(cnt + (v >> 6), 1)] = (uint16_t)rr;
        }
        __syncthreads();
        // ---- reduce: 64 sorted records at a time per wavefront, the K columns in two rounds
        // (0-15, 16-K: each atomic instruction of a row touches one 64-byte line of a row that
        // starts on a 128-byte boundary)
        for (int base = wave * 64; base < nb; base += NT) {
            const int p = base + lane;
            int my_sl = -1;
            float w = 0.f, sgv = 0.f, cc[3] = {0.f, 0.f, 0.f}, gc[3] = {0.f, 0.f, 0.f};
            float bx[NB > 0 ? NB : 1];
            if (p < nb) {
                const int rr = (int)order[p];
                const int v = (int)r_sl[rr];
                my_sl = v >> 6;
                const int rl = v & 63;
                sgv = r_sg[rr];
#pragma unroll
                for (int c3 = 0; c3 < C; ++c3) cc[c3] = r_c[c3 * R + rr];
                if constexpr (FMT == FMT_SH) {
                    w = r_w[rr];
#pragma unroll
                    for (int i = 0; i < NB; ++i) bx[i] = bases[rl * BDS + i];
#pragma unroll
                    for (int c3 = 0; c3 < C; ++c3) gc[c3] = gl[rl * 3 + c3];
                }
            }
            seg[lane] = my_sl >= 0 ? keys[my_sl] : -1;          // the feature row staged row `lane` belongs to
            const int grp = lane >> 4, sub = lane & 15;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int c_lo = h2 * HALF;
                const int ncol = (h2 == 0) ? HALF : K - HALF;
                if (ncol > 0) {
                    if (p < nb) {
                        float* st = stage + lane * KS;
#pragma unroll
                        for (int j = 0; j < HALF; ++j) {
                            const int col = c_lo + j;           // compile-time after unrolling
                            if (col < K) {
                                float val;
                                if (col == K - 1) val = sgv;
                                else if constexpr (FMT == FMT_SH) val = w * bx[col % (NB > 0 ? NB : 1)] * cc[col / (NB > 0 ? NB : 1)] * gc[col / (NB > 0 ? NB : 1)];
                                else val = cc[col];
                                st[j] = val;
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const bool has = sub < ncol;
                    int cur = -1;
                    float acc = 0.f;
                    unsigned long long nreq = 0;
                    int keyv[16];
                    float xv[16];
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int row = grp * 16 + t;
                        keyv[t] = seg[row];
                        xv[t] = stage[row * KS + (has ? sub : 0)];
                    }
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int key = keyv[t];
                        if (key != cur) {
                            if (cur >= 0 && has) atomicAdd(grad + (int64_t)cur * gstride + c_lo + sub, acc);
                            if constexpr (COUNT) nreq += (cur >= 0 && sub == 0) ? 1u : 0u;
                            acc = 0.f;
                            cur = key;
                        }
                        acc += xv[t];
                    }
                    if (cur >= 0 && has) atomicAdd(grad + (int64_t)cur * gstride + c_lo + sub, acc);
                    if constexpr (COUNT) {
                        nreq += (cur >= 0 && sub == 0) ? 1u : 0u;
                        nreq = wave_sum(nreq);
                        if (lane == 0 && nreq) atomicAdd(counters, nreq);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (k0 + RPP * W >= maxn) break;                 // last pass (scalar condition)
        __syncthreads();
        for (int i = threadIdx.x; i < T; i += NT) { keys[i] = -1; cnt[i] = 0; }
        for (int i = threadIdx.x; i < R; i += NT) r_sl[i] = 0xffffffffu;
        __syncthreads();
    }
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
//            wavefront 0 then runs along the rays (weight, transmittance, accum).  The second
//            pass's total_color goes to L.terms, one float per record (position-major).
//   sweep 2  (:439-494 + the colour entries :419-425) per window: lane = ray forms the attenuation
//            again (sigma gather) and enters the record's feature row in a hash table; wavefront
//            0 runs along the rays (accum -= weight * total_color; sigma entries); counting sort
//            by feature row; then lane = COLUMN: a group of K lanes takes one distinct row, forms
//            the sigmoid of its column once per (tile, window, row), adds up the row's records
//            -- ((weight * sig) * (1 - sig)) * g_c of the record's ray, the reference's order --
//            and sends ONE atomic row: sigmoid work and requests divided by the reuse (2.0-2.6x
//            at 1024 x 1024 / depth 9, exp/reuse_probe.py).
// Rays whose list overflowed: a tail-only launch of render_bwd_kernel<..., GATHER> in front
// (aux.z / .w carry its pass-1 results), as for grad_fused_kernel<EXACT>.
template <int K>
__global__ void __launch_bounds__(512, 6)
grad_wide_kernel(TreeDev tr, RaysDev rays, Opts opt, const float* __restrict__ grad_out,
                 RecLists L, const uint4* __restrict__ aux, float* __restrict__ grad, int gstride) {
    static_assert(K == 8 || K == 16 || K == 32, "row widths with an instance");
    constexpr int C = K - 1, W = 8, NT = 64 * W, T = 1024, R = 1024, RPP = R / (64 * W);
    constexpr int KG = K | 1;                                // odd stride: conflict-free gradient rows
    constexpr int SPW = 64 / K;                              // distinct rows a wavefront reduces at a time
    static_assert(RPP == 2, "sizes");
    __shared__ int32_t keys[T];
    __shared__ int32_t cnt[T];
    __shared__ uint16_t order[R];                // sweep 1: the window's records, compacted; sweep 2: sorted by row
    __shared__ uint16_t slots[T];                // sweep 2: the occupied table entries
    __shared__ uint32_t r_sl[R];                 // sweep 1: feature row; sweep 2: table entry << 6 | lane, ~0: no record
    __shared__ float r_w[R], r_sg[R], r_t1[R], r_dt[R];   // (sweep 2: r_t1 / r_dt hold weight / sigma entry in sorted order)
    __shared__ uint8_t s_ray[R];                 // sweep 2: the ray of the sorted record
    __shared__ float gl[64 * KG];
    __shared__ float dsl[64];
    __shared__ int32_t s_nb, s_ns;
    float* __restrict__ tot2 = reinterpret_cast<float*>(L.terms);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t tabreg = rec_tab_reg(L, blockIdx.x, lane);
    const int64_t q = ray_of_thread(rays, (int64_t)blockIdx.x * 64 + lane);
    uint4 a = make_uint4(0u, 0u, 0u, 0u);
    if (q < rays.Q) a = aux[q];
    const int nrec = (int)(a.x & ~kRecOverflow);
    int maxn = nrec;
    for (int off = 32; off > 0; off >>= 1) maxn = max(maxn, __shfl_xor(maxn, off, 64));
    maxn = __builtin_amdgcn_readfirstlane(maxn);             // everything that decides barriers is scalar
    if (maxn == 0) return;                                   // the same in every wavefront of the workgroup

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
    __syncthreads();

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
                if (have[rd]) e[rd] = rec_get(L.rec + rec_index_in(rec_block_u(L, tabreg, blockIdx.x, k >> 3), lane, k));
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
                        order[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)slot;
                    }
                }
            }
        }
        __syncthreads();
        const int nb1 = __builtin_amdgcn_readfirstlane(s_nb);
        // G = K / 8 neighbouring lanes per record, 8 row columns each: four times the busy lanes and a
        // quarter of the dependent work per lane (a window holds a few hundred records for 512 lanes);
        // the sums over the channels run in the reference's order, handed from lane to lane
        constexpr int G = K / 8;
        const int gq = threadIdx.x & (G - 1);
        for (int p0 = wave * (64 / G); p0 < nb1; p0 += NT / G) {      // (scalar bounds: every lane takes part in the shuffles)
            const int p = p0 + (lane / G);
            const bool on = p < nb1;
            const int slot = on ? (int)order[p] : 0;
            const int ray = slot & 63;
            float row[8];
            float a1[8];
            double a2[8];
            if (on) {
                load_row<8>(tr.features + (int64_t)(int32_t)r_sl[slot] * K + 8 * gq, row);
                const float* __restrict__ gr = gl + ray * KG + 8 * gq;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (8 * gq + j < C) {                      // (the last lane's eighth column is sigma)
                        const double sd = sigmoid_d<true>(row[j]);
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
        __syncthreads();
        if (threadIdx.x == 0) s_nb = 0;
        // along the rays (wavefront 0); everybody: the hand-over, 256 contiguous bytes per position
#pragma unroll 1
        for (int rd = 0; rd < RPP; ++rd) {
            const int k = k0 + rd * W + wave;
            if (k < nrec) {
                const int64_t blk = rec_block_u(L, tabreg, blockIdx.x, k >> 3);
                tot2[terms_index_pm(blk, lane, k)] = r_sg[(rd * W + wave) * 64 + lane];
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
        __syncthreads();
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
    __syncthreads();
    for (int k0 = 0; k0 < maxn; k0 += RPP * W) {
        {   // the window's two rounds together: records and hand-over first, then the sigma gathers, then the table
            uint2 e[RPP];
            float t2[RPP], sigma[RPP];
            bool have[RPP];
#pragma unroll
            for (int rd = 0; rd < RPP; ++rd) {
                const int k = k0 + rd * W + wave;
                have[rd] = k < nrec;
                e[rd] = make_uint2(0u, 0u);
                t2[rd] = 0.f;
                if (have[rd]) {
                    const int64_t blk = rec_block_u(L, tabreg, blockIdx.x, k >> 3);
                    e[rd] = rec_get(L.rec + rec_index_in(blk, lane, k));
                    t2[rd] = tot2[terms_index_pm(blk, lane, k)];
                }
            }
#pragma unroll
            for (int rd = 0; rd < RPP; ++rd) {
                sigma[rd] = 0.f;
                if (have[rd]) sigma[rd] = tr.features[(int64_t)(int32_t)e[rd].x * K + (K - 1)];
            }
#pragma unroll
            for (int rd = 0; rd < RPP; ++rd) {
                if (have[rd]) {
                    const int32_t idx = (int32_t)e[rd].x;
                    uint32_t h = ((uint32_t)idx * 0x9E3779B1u) >> (32 - __builtin_ctz(T));
                    while (true) {
                        const int32_t old = atomicCAS(keys + h, -1, idx);
                        if (old == -1 || old == idx) break;
                        h = (h + 1u) & (uint32_t)(T - 1);
                    }
                    atomicAdd(cnt + h, 1);
                    const int slot = (rd * W + wave) * 64 + lane;
                    r_sl[slot] = (h << 6) | (uint32_t)lane;
                    r_w[slot] = pexpf<true>(-__uint_as_float(e[rd].y) * sigma[rd] * ds);
                    r_sg[slot] = t2[rd];
                    r_dt[slot] = __uint_as_float(e[rd].y);
                }
            }
        }
        __syncthreads();
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
                if (mine[j2] > 0) slots[urun++] = (uint16_t)(lane * PER + j2);
            }
            if (lane == 63) { s_nb = incl; s_ns = uincl; }
        }
        __syncthreads();
        for (int rr = threadIdx.x; rr < R; rr += NT) {
            const uint32_t v = r_sl[rr];
            if (v != 0xffffffffu) {
                const int pos = atomicAdd(cnt + (v >> 6), 1);
                r_t1[pos] = r_w[rr];                         // (r_dt was consumed by the chain above)
                r_dt[pos] = r_sg[rr];
                s_ray[pos] = (uint8_t)(v & 63u);
            }
        }
        __syncthreads();
        // ---- reduce: lane = column; after the scatter cnt[h] is where the records of entry h END
        const int ns = __builtin_amdgcn_readfirstlane(s_ns);
        const int col = lane & (K - 1), sub = lane / K;
        constexpr int D = 4;                                 // rows in flight per lane: a row gather is ~1 us, its use ~0.3 us
        for (int i0 = wave * SPW; i0 < ns; i0 += D * W * SPW) {
            int32_t idxs[D];
            int ps[D], pes[D];
            float xs[D];
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int i = i0 + u * W * SPW + sub;
                idxs[u] = -1; ps[u] = 0; pes[u] = 0; xs[u] = 0.f;
                if (i < ns) {
                    const int h = (int)slots[i];
                    idxs[u] = keys[h];
                    pes[u] = cnt[h];
                    // the entry's records start where the previous occupied entry's end (table order = sorted order)
                    if (i > 0) ps[u] = cnt[(int)slots[i - 1]];
                    xs[u] = tr.features[(int64_t)idxs[u] * K + col];
                }
            }
#pragma unroll
            for (int u = 0; u < D; ++u) {
                if (i0 + u * W * SPW >= ns) break;           // (scalar)
                const int32_t idx = idxs[u];
                const int p = ps[u];
                float sig = 0.f, om = 0.f;
                if (idx >= 0 && col < C) {
                    sig = (float)sigmoid_d<true>(xs[u]);
                    om = 1.f - sig;
                }
                float acc = 0.f;
                int n_here = pes[u] - p, n_max = n_here;
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
        }
        if (k0 + RPP * W >= maxn) break;                     // last window (scalar condition)
        __syncthreads();
        for (int i = threadIdx.x; i < T; i += NT) { keys[i] = -1; cnt[i] = 0; }
        for (int i = threadIdx.x; i < R; i += NT) r_sl[i] = 0xffffffffu;
        __syncthreads();
    }
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
    bool over = false;
    float t_resume = 0.f;
    while (t < r.tmax) {
        Sample s;
        march_step<N2>(tr, r, opt.step_size, t, s);
        if (s.valid) {
            const float sigma = tr.features[(int64_t)s.idx * K + (K - 1)];
            if (sigma > opt.sigma_thresh) {
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
                light *= pexpf(-s.delta_t * r.delta_scale * sigma);
                if constexpr (!REC) {
                    if (light <= opt.stop_thresh) break;
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
    for (int k = 0; k < nrec; ++k) {
        uint2* slot = rec + rec_index(L, tid, k);
        const uint2 e = rec_get(slot);
        const float delta_t = __uint_as_float(e.y);
        const float sigma = tr.features[(int64_t)(int32_t)e.x * K + (K - 1)];
        light *= pexpf(-delta_t * sigma * r.delta_scale);
        rec_put(slot, e.x, delta_t * r.delta_scale * g);
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

template <bool N2>
__global__ void __launch_bounds__(kBlock)
query_fwd_kernel(TreeDev tr, const float* __restrict__ points, int64_t Q,
                 float* __restrict__ values, int64_t* __restrict__ node_ids,
                 int64_t* __restrict__ data_ids, uint8_t* __restrict__ hit_mask) {
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (q >= Q) return;
    const float* p = points + 3 * q;
    const float px = tr.offset[0] + tr.scaling[0] * p[0];
    const float py = tr.offset[1] + tr.scaling[1] * p[1];
    const float pz = tr.offset[2] + tr.scaling[2] * p[2];
    Leaf lf;
    locate<N2>(tr, px, py, pz, lf);
    node_ids[q] = (int64_t)lf.slot;
    if (hit_mask != nullptr) hit_mask[lf.slot] = 1;
    const int32_t idx = tr.data[lf.slot];
    const int K = tr.K;
    float* v = values + q * K;
    if (idx >= 0 && (int64_t)idx < tr.M) {
        data_ids[q] = idx;
        const float* row = tr.features + (int64_t)idx * K;
        for (int i = 0; i < K; ++i) v[i] = row[i];
    } else {
        data_ids[q] = -1;
        for (int i = 0; i < K; ++i) v[i] = 0.f;
    }
}

template <bool N2>
__global__ void __launch_bounds__(kBlock)
query_bwd_kernel(TreeDev tr, const float* __restrict__ points, int64_t Q,
                 const float* __restrict__ grad_out, float* __restrict__ grad) {
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (q >= Q) return;
    const float* p = points + 3 * q;
    const float px = tr.offset[0] + tr.scaling[0] * p[0];
    const float py = tr.offset[1] + tr.scaling[1] * p[1];
    const float pz = tr.offset[2] + tr.scaling[2] * p[2];
    Leaf lf;
    locate<N2>(tr, px, py, pz, lf);
    const int32_t idx = tr.data[lf.slot];
    if (idx < 0 || (int64_t)idx >= tr.M) return;
    const int K = tr.K;
    for (int i = 0; i < K; ++i) atomicAdd(grad + (int64_t)idx * K + i, grad_out[q * K + i]);
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
template <typename V>
__global__ void __launch_bounds__(kBlock)
compact_rows_kernel(const V* __restrict__ src, int64_t n, int Kv, int stride_v, V* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / Kv;
        const int c = (int)(i - r * Kv);
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + r * stride_v + c), dst + i);
    }
}

// ---------------------------------------------------------------------------
// Acceleration grid build (N == 2): one thread per cell, see locate_accel()
// ---------------------------------------------------------------------------

__global__ void __launch_bounds__(kBlock)
accel_build_kernel(TreeDev tr, int G, uint2* __restrict__ cells) {
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= (1u << (3 * G))) return;
    const uint32_t mask = (1u << G) - 1u;
    const uint32_t cz = c & mask, cy = (c >> G) & mask, cx = c >> (2 * G);
    int32_t node = 0;
    for (int k = 1; k <= G; ++k) {
        const int sh = G - k;
        const uint32_t c3 = (((cx >> sh) & 1u) << 2) | (((cy >> sh) & 1u) << 1) | ((cz >> sh) & 1u);
        const uint32_t slot = ((uint32_t)node << 3) + c3;
        const int32_t skip = tr.child[slot];
        if (skip == 0) {
            cells[c] = make_uint2((uint32_t)tr.data[slot], kAccelLeaf | (uint32_t)k);
            return;
        }
        node += skip;
    }
    cells[c] = make_uint2((uint32_t)node, 0u);
}

// ... and the (child, data) pairs the descent below the grid reads
__global__ void __launch_bounds__(kBlock)
accel_nodes_kernel(const int32_t* __restrict__ child, const int32_t* __restrict__ data, int64_t n,
                   uint2* __restrict__ nodes) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) nodes[i] = make_uint2((uint32_t)child[i], (uint32_t)data[i]);
}

}  // namespace svoxt

// ===========================================================================
// C ABI
// ===========================================================================

using namespace svoxt;

namespace {
thread_local char g_err[512] = "";
int64_t* g_bwd_counters = nullptr;      // svoxt_set_bwd_counters (instrumentation)
}

namespace svoxt {

int set_error(int code, const char* fmt, const char* a, const char* b) {
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(SVOXT_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return SVOXT_OK;
}

static int fail(int code, const char* fmt, const char* a = "", const char* b = "") {
    return set_error(code, fmt, a, b);
}

int check_tree(const svoxt_tree* t, const char* fn) {
    if (t == nullptr) return fail(SVOXT_ERR_INVALID, "%s: tree is NULL", fn);
    if (t->features == nullptr && t->M > 0) return fail(SVOXT_ERR_INVALID, "%s: tree.features is NULL", fn);
    if (t->data == nullptr || t->child == nullptr) return fail(SVOXT_ERR_INVALID, "%s: tree.data / tree.child is NULL", fn);
    if (t->offset == nullptr || t->scaling == nullptr) return fail(SVOXT_ERR_INVALID, "%s: tree.offset / tree.scaling is NULL", fn);
    if (t->K < 1 || t->M < 0) return fail(SVOXT_ERR_INVALID, "%s: bad feature table extents", fn);
    if (t->N < 2) return fail(SVOXT_ERR_INVALID, "%s: branching factor N must be >= 2", fn);
    if (t->n_internal < 1) return fail(SVOXT_ERR_INVALID, "%s: n_internal must be >= 1", fn);
    if ((double)t->n_internal * t->N * t->N * t->N >= 2147483648.0)
        return fail(SVOXT_ERR_INVALID, "%s: tree too large for 32-bit slot indices", fn);
    if (t->accel != nullptr && (t->accel_log2 < 1 || t->accel_log2 > 8))
        return fail(SVOXT_ERR_INVALID, "%s: accel_log2 must be in [1, 8]", fn);
    if (t->xform != nullptr && t->xform_dim != 0 && t->xform_dim != 3 && t->xform_dim != 4)
        return fail(SVOXT_ERR_INVALID, "%s: xform_dim must be 3 or 4", fn);
    return SVOXT_OK;
}

int check_rays(const svoxt_rays* r, const char* fn) {
    if (r == nullptr) return fail(SVOXT_ERR_INVALID, "%s: rays is NULL", fn);
    if (r->Q < 0) return fail(SVOXT_ERR_INVALID, "%s: negative ray count", fn);
    if (r->c2w != nullptr) {
        if (r->image_width < 1 || r->image_height < 1 || (int64_t)r->image_width * r->image_height != r->Q)
            return fail(SVOXT_ERR_INVALID, "%s: camera mode needs Q == image_width * image_height", fn);
        if (!(r->fx != 0.f) || !(r->fy != 0.f)) return fail(SVOXT_ERR_INVALID, "%s: camera focal lengths must be non-zero", fn);
    } else if (r->Q > 0 && (r->origins == nullptr || r->dirs == nullptr || r->vdirs == nullptr)) {
        return fail(SVOXT_ERR_INVALID, "%s: rays.origins / dirs / vdirs is NULL", fn);
    }
    if (r->Q >= (int64_t)kBlock * 2147483647LL) return fail(SVOXT_ERR_INVALID, "%s: too many rays", fn);
    if (r->image_width < 0 || r->image_height < 0) return fail(SVOXT_ERR_INVALID, "%s: negative image extent", fn);
    return SVOXT_OK;
}

int check_opts(const svoxt_options* o, const svoxt_tree* t, const char* fn, bool needs_basis) {
    if (o == nullptr) return fail(SVOXT_ERR_INVALID, "%s: options is NULL", fn);
    if (o->format < SVOXT_FORMAT_RGBA || o->format > SVOXT_FORMAT_ASG)
        return fail(SVOXT_ERR_INVALID, "%s: unknown data format", fn);
    if (!needs_basis || o->format == SVOXT_FORMAT_RGBA) return SVOXT_OK;
    if (o->basis_dim < 1 || o->basis_dim > 25)
        return fail(SVOXT_ERR_INVALID, "%s: basis_dim must be in [1, 25]", fn);
    if (o->format == SVOXT_FORMAT_SH && o->basis_dim != 1 && o->basis_dim != 4 && o->basis_dim != 9 &&
        o->basis_dim != 16 && o->basis_dim != 25)
        return fail(SVOXT_ERR_INVALID, "%s: SH basis_dim must be 1, 4, 9, 16 or 25", fn);
    if (o->min_comp < 0 || o->max_comp >= o->basis_dim)
        return fail(SVOXT_ERR_INVALID, "%s: min_comp / max_comp outside [0, basis_dim)", fn);
    if (o->format == SVOXT_FORMAT_SG || o->format == SVOXT_FORMAT_ASG) {
        const int need = o->format == SVOXT_FORMAT_SG ? 4 : 11;
        if (t->extra_data == nullptr || t->extra_rows < o->basis_dim || t->extra_cols < need)
            return fail(SVOXT_ERR_INVALID, "%s: SG/ASG formats need extra_data [basis_dim, >=4/11]", fn);
    }
    return SVOXT_OK;
}

TreeDev to_dev(const svoxt_tree* t) {
    TreeDev d;
    d.features = t->features; d.M = t->M; d.K = t->K; d.N = t->N;
    d.data = t->data; d.child = t->child; d.offset = t->offset; d.scaling = t->scaling;
    d.extra = t->extra_data; d.extra_rows = t->extra_rows; d.extra_cols = t->extra_cols;
    d.weight_accum = t->weight_accum;
    d.xform = t->xform;   // consulted by the generic render kernels only
    d.xform_dim = t->xform_dim == 4 ? 4 : 3;
    const bool use_accel = t->accel != nullptr && t->N == 2;
    d.accel = use_accel ? reinterpret_cast<const uint2*>(t->accel) : nullptr;
    d.accel_g = use_accel ? t->accel_log2 : 0;
    return d;
}

RaysDev to_dev(const svoxt_rays* r) {
    RaysDev d;
    d.origins = r->origins; d.dirs = r->dirs; d.vdirs = r->vdirs; d.Q = r->Q;
    // image hint: usable only if the batch is exactly a W x H image of 8x8 tiles
    const bool tiled = r->image_width > 0 && r->image_height > 0 && r->image_width % 8 == 0 &&
                       r->image_height % 8 == 0 && (int64_t)r->image_width * r->image_height == r->Q;
    d.tiles_per_row = tiled ? r->image_width / 8 : 0;
    d.tile0 = 0;
    d.c2w = r->c2w; d.fx = r->fx; d.fy = r->fy;
    d.width = r->image_width; d.height = r->image_height;
    return d;
}

Opts to_dev(const svoxt_options* o) {
    Opts d;
    static_assert(sizeof(Opts) == sizeof(svoxt_options), "options layout");
    memcpy(&d, o, sizeof(d));
    return d;
}

}  // namespace svoxt

namespace {

inline unsigned nblocks(int64_t Q) { return (unsigned)((Q + kBlock - 1) / kBlock); }

// sample lists: rec[tile][block of 8][lane][8], 8 bytes per record (rec_index)
inline int64_t rec_rays(int64_t Q) { return (Q + 63) / 64 * 64; }
// the kernels' view of caller-owned lists / of a dense workspace region
inline RecLists lists_dev(const svoxt_sample_lists* l, int64_t Q, int term_bytes = 16) {
    RecLists L;
    L.rec = reinterpret_cast<uint2*>(l->rec);
    L.tab = reinterpret_cast<int32_t*>(l->blocktab);
    L.pool_next = reinterpret_cast<int32_t*>(l->pool_next);
    L.pool_blocks = l->blocktab != nullptr ? l->pool_blocks : rec_rays(Q) / 64 * (l->max_samples / kRecBlock);
    L.S = l->max_samples;
    // term_bytes (16; 4 for the one-sigmoid-pass backward of wide rows) per record slot, 16-byte aligned, or not at all
    const bool have_terms = l->terms != nullptr && ((uintptr_t)l->terms & 15u) == 0 &&
                            l->terms_bytes >= L.pool_blocks * (int64_t)(64 * kRecBlock) * term_bytes;
    L.terms = have_terms ? reinterpret_cast<float4*>(l->terms) : nullptr;
    return L;
}
inline RecLists dense_lists(void* rec, int64_t S, int64_t Q) {
    RecLists L;
    L.rec = reinterpret_cast<uint2*>(rec);
    L.tab = nullptr;
    L.pool_next = nullptr;
    L.pool_blocks = rec_rays(Q) / 64 * (S / kRecBlock);
    L.S = (int)S;
    L.terms = nullptr;
    return L;
}
inline int64_t rec_capacity(int64_t bytes, int64_t Q) {       // records per ray that fit: a multiple of 8, at most 4096
    if (bytes <= 0 || Q <= 0) return 0;
    int64_t S = bytes / (8 * rec_rays(Q)) / kRecBlock * kRecBlock;
    return S > 4096 ? 4096 : S;
}

// Specialised payloads: (format, C, BD) with all components selected.
struct Payload { int fmt, C, BD; };

// transformation_matrices only matter for view-dependent formats (for RGBA the
// reference's per-sample basis re-evaluation is a no-op, rt_kernel.cu:181-183)
bool uses_xform(const svoxt_tree* t, const svoxt_options* o) {
    return t->xform != nullptr && o->format != SVOXT_FORMAT_RGBA;
}

bool full_comp(const svoxt_options* o) {
    return o->format == SVOXT_FORMAT_RGBA || (o->min_comp == 0 && o->max_comp == o->basis_dim - 1);
}

// specialised kernels with per-leaf view rotations: SH payloads on N = 2 trees
template <bool REC>
bool launch_fwd_xform(const TreeDev& tr, const RaysDev& rays, const Opts& opt, int C, float* out,
                      RecLists L, uint4* aux, hipStream_t st) {
    if (opt.format != FMT_SH || C != 3) return false;
    const unsigned nb = nblocks(rays.Q);
#define SVOXT_FWD_XF(BB)                                                                                    \
    hipLaunchKernelGGL((render_fwd_kernel<FMT_SH, 3, BB, true, REC, true>), dim3(nb), dim3(kBlock), 0, st,  \
                       tr, rays, opt, out, L, aux);                                                    \
    return true;
    switch (opt.basis_dim) {
        case 1: SVOXT_FWD_XF(1)
        case 4: SVOXT_FWD_XF(4)
        case 9: SVOXT_FWD_XF(9)
        case 16: SVOXT_FWD_XF(16)
        case 25: SVOXT_FWD_XF(25)
    }
#undef SVOXT_FWD_XF
    return false;
}

template <bool REPLAY>
bool launch_bwd_xform(const TreeDev& tr, const RaysDev& rays, const Opts& opt, int C,
                      const float* grad_out, float* grad, int gstride, RecLists L, const uint4* aux,
                      const float* fwd_out, hipStream_t st) {
    if (opt.format != FMT_SH || C != 3) return false;
    const unsigned nb = nblocks(rays.Q);
#define SVOXT_BWD_XF(BB)                                                                                      \
    hipLaunchKernelGGL((render_bwd_kernel<FMT_SH, 3, BB, true, REPLAY, true>), dim3(nb), dim3(kBlock), 0, st, \
                       tr, rays, opt, grad_out, grad, gstride, L, aux, fwd_out);                         \
    return true;
    switch (opt.basis_dim) {
        case 1: SVOXT_BWD_XF(1)
        case 4: SVOXT_BWD_XF(4)
        case 9: SVOXT_BWD_XF(9)
        case 16: SVOXT_BWD_XF(16)
        case 25: SVOXT_BWD_XF(25)
    }
#undef SVOXT_BWD_XF
    return false;
}

// can the specialised kernels serve this tree / options pair with its view rotations?
bool xform_special(const svoxt_tree* t, const svoxt_options* o) {
    return t->N == 2 && o->format == SVOXT_FORMAT_SH && t->K == 3 * o->basis_dim + 1 &&
           (o->basis_dim == 1 || o->basis_dim == 4 || o->basis_dim == 9 || o->basis_dim == 16 || o->basis_dim == 25);
}

template <bool N2, bool REC>
bool launch_fwd_special(const TreeDev& tr, const RaysDev& rays, const Opts& opt, int C, float* out,
                        RecLists L, uint4* aux, hipStream_t st) {
    const unsigned nb = nblocks(rays.Q);
#define SVOXT_FWD(F, CC, BB)                                                                   \
    hipLaunchKernelGGL((render_fwd_kernel<F, CC, BB, N2, REC>), dim3(nb), dim3(kBlock), 0, st, \
                       tr, rays, opt, out, L, aux);                                       \
    return true;
    if (opt.format == FMT_RGBA) {
        if (C == 3) { SVOXT_FWD(FMT_RGBA, 3, 0) }
        if (C == 7) { SVOXT_FWD(FMT_RGBA, 7, 0) }
        if (C == 15) { SVOXT_FWD(FMT_RGBA, 15, 0) }
        if (C == 31) { SVOXT_FWD(FMT_RGBA, 31, 0) }
    } else if (opt.format == FMT_SH && C == 3) {
        switch (opt.basis_dim) {
            case 1: SVOXT_FWD(FMT_SH, 3, 1)
            case 4: SVOXT_FWD(FMT_SH, 3, 4)
            case 9: SVOXT_FWD(FMT_SH, 3, 9)
            case 16: SVOXT_FWD(FMT_SH, 3, 16)
            case 25: SVOXT_FWD(FMT_SH, 3, 25)
        }
    }
#undef SVOXT_FWD
    return false;
}

// two-kernel forward (march_rec_kernel + a shade kernel + tail launch of render_fwd_kernel), all
// components.  Default: on for RGBA-style rows of 8 / 16 / 32 floats (shade_chan_kernel: r02,
// 1024 x 1024 depth-9 K = 32), off for the 3-channel payloads (shade_tile_kernel:
// 800x800 depth-8 SH9 0.156 + 0.123 ms against 0.247 ms for the one-kernel forward).
// SVOXT_FWD_SPLIT=0 / 1 forces it off / on where a shade kernel exists.
// with_terms: a recording forward that is to leave the backward's (att, e_c) -- the tile shade kernel
// writes them 1 KB at a time for nothing, the one-kernel forward pays 0.06 ms for its scattered lines
// (r02: 0.145 + 0.13 ms against 0.316 ms) -- so then the two kernels are the default for 3 channels too.
bool fwd_split_enabled(const svoxt_tree* t, const svoxt_options* o, bool with_terms = false) {   // read per call
    const char* e = getenv("SVOXT_FWD_SPLIT");
    if (e != nullptr && *e != 0) return atoi(e) != 0;
    if (o->format == SVOXT_FORMAT_RGBA && (t->K == 8 || t->K == 16 || t->K == 32)) return true;
    return with_terms && t->xform == nullptr && t->weight_accum == nullptr;
}

bool fwd_split_payload(const svoxt_tree* t, const svoxt_options* o, int C) {
    if (t->weight_accum != nullptr) return false;
    if (o->format == SVOXT_FORMAT_RGBA && (t->K == 8 || t->K == 16 || t->K == 32)) return true;   // channel lanes
    if (C != 3) return false;
    if (o->format == SVOXT_FORMAT_RGBA) return t->K == 4;
    if (o->format != SVOXT_FORMAT_SH || t->K != 3 * o->basis_dim + 1) return false;
    return o->basis_dim == 1 || o->basis_dim == 4 || o->basis_dim == 9 || o->basis_dim == 16 || o->basis_dim == 25;
}

// The shade (+ tail) launches of one range of tiles.
template <bool N2, bool STOP>
bool launch_shade(const TreeDev& tr, const RaysDev& rays, const Opts& opt, float* out, RecLists L, uint4* aux,
                  bool xf, bool fast, unsigned nb, hipStream_t st) {
#define SVOXT_SPLIT(F, BB, X)                                                                                 \
    {                                                                                                         \
        if (!STOP && !X && L.terms != nullptr)                                                                \
            hipLaunchKernelGGL((shade_tile_kernel<F, BB, false, STOP, !STOP>), dim3(nb), dim3(512), 0, st,    \
                               tr, rays, opt, L, aux, out);                                                   \
        else                                                                                                  \
        hipLaunchKernelGGL((shade_tile_kernel<F, BB, X, STOP>), dim3(nb), dim3(512), 0, st, tr, rays, opt,    \
                           L, aux, out);                                                                 \
        hipLaunchKernelGGL((render_fwd_kernel<F, 3, BB, N2, false, X, true>), dim3(nb), dim3(kBlock), 0, st,  \
                           tr, rays, opt, out, L, aux);                                      \
        return true;                                                                                          \
    }
    if (opt.format == FMT_RGBA && tr.K != 4) {
        // rows of 8 / 16 / 32 floats: channels on lanes, 64 / K rays per wavefront, 4 wavefronts per workgroup
#define SVOXT_CHAN(KK)                                                                                        \
        {                                                                                                     \
            const unsigned nbc = (unsigned)(((int64_t)nb * 64 / (64 / KK) + 3) / 4);                          \
            if (fast) hipLaunchKernelGGL((shade_chan_kernel<KK, STOP, true>), dim3(nbc), dim3(256), 0, st,    \
                                         tr, rays, opt, L, aux, out);                                    \
            else hipLaunchKernelGGL((shade_chan_kernel<KK, STOP, false>), dim3(nbc), dim3(256), 0, st,        \
                                    tr, rays, opt, L, aux, out);                                         \
            if (fast) hipLaunchKernelGGL((tail_chan_kernel<KK, N2, true>), dim3(nb), dim3(256), 0, st,        \
                                         tr, rays, opt, aux, out);                                            \
            else hipLaunchKernelGGL((tail_chan_kernel<KK, N2, false>), dim3(nb), dim3(256), 0, st,            \
                                    tr, rays, opt, aux, out);                                                 \
            return true;                                                                                      \
        }
        switch (tr.K) {
            case 8: SVOXT_CHAN(8)
            case 16: SVOXT_CHAN(16)
            case 32: SVOXT_CHAN(32)
        }
#undef SVOXT_CHAN
        return false;
    }
    if (opt.format == FMT_RGBA) SVOXT_SPLIT(FMT_RGBA, 0, false)
    if constexpr (N2) {
        if (xf) {
            switch (opt.basis_dim) {
                case 1: SVOXT_SPLIT(FMT_SH, 1, true)
                case 4: SVOXT_SPLIT(FMT_SH, 4, true)
                case 9: SVOXT_SPLIT(FMT_SH, 9, true)
                case 16: SVOXT_SPLIT(FMT_SH, 16, true)
                case 25: SVOXT_SPLIT(FMT_SH, 25, true)
            }
            return false;
        }
    }
    switch (opt.basis_dim) {
        case 1: SVOXT_SPLIT(FMT_SH, 1, false)
        case 4: SVOXT_SPLIT(FMT_SH, 4, false)
        case 9: SVOXT_SPLIT(FMT_SH, 9, false)
        case 16: SVOXT_SPLIT(FMT_SH, 16, false)
        case 25: SVOXT_SPLIT(FMT_SH, 25, false)
    }
#undef SVOXT_SPLIT
    return false;
}

// A side stream and a few events per device, made on first use: the march of one range of tiles
// runs on the caller's stream while the side stream shades the range before it.  (The idea: the
// march is bound by its dependent loads and leaves the vector ALUs half idle, the shade kernels
// are bound by those ALUs.  See fwd_split_chunks for what it measured.)
struct SideStream {
    hipStream_t st = nullptr;
    hipEvent_t ev[8] = {};
    bool ok = false;
};
SideStream* side_stream() {
    static thread_local SideStream tab[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    SideStream& s = tab[dev];
    if (!s.ok) {
        if (hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess) return nullptr;
        for (auto& e : s.ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        s.ok = true;
    }
    return &s;
}

// Measured r02 and NOT the default: with 4 ranges the forward got slower, not faster (800x800
// depth-8 SH9 0.29 -> 0.59 ms, 1024x1024 depth-9 K = 32 1.47 -> 2.0 ms): each cross-stream event
// dependency costs more than the overlap returns.  SVOXT_FWD_CHUNKS=n (2..7) keeps the experiment
// reachable; the default is one range on the caller's stream.
int fwd_split_chunks(int64_t ntiles) {
    const char* e = getenv("SVOXT_FWD_CHUNKS");
    int n = (e != nullptr && *e != 0) ? atoi(e) : 1;
    if (n < 1) n = 1;
    if (n > 7) n = 7;
    return (int64_t)n > ntiles ? (int)ntiles : n;
}

template <bool N2, bool STOP>
bool launch_fwd_split(const TreeDev& tr, const RaysDev& rays, const Opts& opt, float* out,
                      RecLists L, uint4* aux, bool xf, bool fast, hipStream_t st,
                      const uint32_t* sigma_mask = nullptr) {
    const unsigned nb = nblocks(rays.Q);
    if (xf && !N2) return false;
    const bool acc = N2 && tr.accel != nullptr;
    int nchunk = fwd_split_chunks(nb);
    SideStream* ss = nchunk > 1 ? side_stream() : nullptr;
    if (ss == nullptr) nchunk = 1;
    bool ok = true;
    for (int c = 0; c < nchunk && ok; ++c) {
        const unsigned lo = (unsigned)((uint64_t)nb * c / nchunk), hi = (unsigned)((uint64_t)nb * (c + 1) / nchunk);
        RaysDev rc = rays;
        rc.tile0 = lo;
        if constexpr (!STOP) {
            if (sigma_mask != nullptr) {
                if (acc) hipLaunchKernelGGL((march_rec_kernel<N2, false, 1, true>), dim3(hi - lo), dim3(kBlock), 0, st, tr, rc, opt, L, aux, sigma_mask);
                else hipLaunchKernelGGL((march_rec_kernel<N2, false, 0, true>), dim3(hi - lo), dim3(kBlock), 0, st, tr, rc, opt, L, aux, sigma_mask);
            }
        }
        if (STOP || sigma_mask == nullptr) {
            if (acc) hipLaunchKernelGGL((march_rec_kernel<N2, STOP, 1>), dim3(hi - lo), dim3(kBlock), 0, st, tr, rc, opt, L, aux, (const uint32_t*)nullptr);
            else hipLaunchKernelGGL((march_rec_kernel<N2, STOP, 0>), dim3(hi - lo), dim3(kBlock), 0, st, tr, rc, opt, L, aux, (const uint32_t*)nullptr);
        }
        if (nchunk == 1) {
            ok = launch_shade<N2, STOP>(tr, rc, opt, out, L, aux, xf, fast, hi - lo, st);
        } else {
            // the side stream takes over this range once its march is done
            if (hipEventRecord(ss->ev[c], st) != hipSuccess || hipStreamWaitEvent(ss->st, ss->ev[c], 0) != hipSuccess) return false;
            ok = launch_shade<N2, STOP>(tr, rc, opt, out, L, aux, xf, fast, hi - lo, ss->st);
        }
    }
    if (nchunk > 1) {   // join: what follows on the caller's stream sees every pixel
        if (hipEventRecord(ss->ev[7], ss->st) != hipSuccess || hipStreamWaitEvent(st, ss->ev[7], 0) != hipSuccess) return false;
    }
    return ok;
}

// two-kernel backward: SH 1/4/9 (also with view rotations) and RGBA with 3 channels
// (K <= 32) on N = 2 trees
bool launch_bwd_gather(const TreeDev& tr, const RaysDev& rays, const Opts& opt, int C,
                       const float* grad_out, float* grad, int gstride, RecLists L, const uint4* aux,
                       const float* fwd_out, float4* coef, bool xf, hipStream_t st, int terms_state = 0) {
    const unsigned nb = nblocks(rays.Q);
    if (C > 3) {
        // RGBA-style rows of 8 / 16 / 32 floats: the exact per-tile form only (one kernel, a float per
        // list slot in L.terms); tails of overflowed rays first, as for the 3-channel fused kernel
        if (opt.format != FMT_RGBA || coef != nullptr || xf || fwd_out != nullptr || L.terms == nullptr) return false;
#define SVOXT_WIDE(KK)                                                                                        \
    {                                                                                                         \
        hipLaunchKernelGGL((render_bwd_kernel<FMT_RGBA, KK - 1, 0, true, true, false, true>), dim3(nb), dim3(kBlock), 0, st, \
                           tr, rays, opt, grad_out, grad, gstride, L, aux, (const float*)nullptr, (float4*)nullptr); \
        hipLaunchKernelGGL((grad_wide_kernel<KK>), dim3(nb), dim3(512), 0, st, tr, rays, opt, grad_out, L, aux, \
                           grad, gstride);                                                                    \
        return true;                                                                                          \
    }
        if (C == 7 && tr.K == 8) SVOXT_WIDE(8)
        if (C == 15 && tr.K == 16) SVOXT_WIDE(16)
        if (C == 31 && tr.K == 32) SVOXT_WIDE(32)
#undef SVOXT_WIDE
        return false;
    }
    if (C != 3) return false;
    // a caller that hands over a coef buffer asks for the two-kernel form; without one (coef_bytes < 0)
    // the per-tile route runs if it can run as ONE kernel.  (The choice is the caller's alone: the
    // Python layer reads SVOXT_BWD_FUSED, the library reads no environment for this.)
    const bool fused = coef == nullptr && !xf;
    if (!fused && coef == nullptr) return false;            // the two-kernel form needs its buffer
    // four wavefronts per tile and tables of 1024 (measured: one wavefront per tile 0.41 ms,
    // two 0.33, four 0.30 before step 15; tables of 512 / 256 cost more passes than they buy)
#define SVOXT_GATHER(F, BB)                                                                                   \
    if (fused) {   /* tails of overflowed rays (a tail-only launch), then list walk and merge in one kernel */ \
        hipLaunchKernelGGL((render_bwd_kernel<F, 3, BB, true, true, false, true>), dim3(nb), dim3(kBlock), 0, st, \
                           tr, rays, opt, grad_out, grad, gstride, L, aux, fwd_out, (float4*)nullptr);   \
        unsigned long long* ctr = reinterpret_cast<unsigned long long*>(g_bwd_counters);                      \
        if (fwd_out != nullptr && ctr == nullptr)                                                             \
            hipLaunchKernelGGL((grad_fused_kernel<F, BB, false>), dim3(nb), dim3(512), 0, st,                 \
                               tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride);                 \
        else if (ctr == nullptr && L.terms != nullptr && terms_state == 2)                                    \
            hipLaunchKernelGGL((grad_fused_kernel<F, BB, true, false, 2>), dim3(nb), dim3(512), 0, st,        \
                               tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride);                      \
        else if (ctr == nullptr && L.terms != nullptr && terms_state == 3)                                    \
            hipLaunchKernelGGL((grad_fused_kernel<F, BB, true, false, 3>), dim3(nb), dim3(512), 0, st,        \
                               tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride);                      \
        else if (ctr == nullptr && L.terms != nullptr)                                                        \
            hipLaunchKernelGGL((grad_fused_kernel<F, BB, true, false, 1>), dim3(nb), dim3(512), 0, st,        \
                               tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride);                      \
        else if (ctr == nullptr)                                                                              \
            hipLaunchKernelGGL((grad_fused_kernel<F, BB, true>), dim3(nb), dim3(512), 0, st,                  \
                               tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride);                 \
        else if (fwd_out != nullptr)                                                                          \
            hipLaunchKernelGGL((grad_fused_kernel<F, BB, false, true>), dim3(nb), dim3(512), 0, st,           \
                               tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride, ctr);            \
        else                                                                                                  \
            hipLaunchKernelGGL((grad_fused_kernel<F, BB, true, true>), dim3(nb), dim3(512), 0, st,            \
                               tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride, ctr);            \
        return true;                                                                                          \
    }                                                                                                         \
    hipLaunchKernelGGL((render_bwd_kernel<F, 3, BB, true, true, false, true>), dim3(nb), dim3(kBlock), 0, st, \
                       tr, rays, opt, grad_out, grad, gstride, L, aux, fwd_out, coef);                   \
    hipLaunchKernelGGL((grad_merge_kernel<F, BB, 1024, 1024, 4>), dim3(nb), dim3(256), 0, st, tr, rays,      \
                       grad_out, L, coef, aux, grad, gstride);                                                \
    return true;
#define SVOXT_GATHER_XF(BB)                                                                                       \
    hipLaunchKernelGGL((render_bwd_kernel<FMT_SH, 3, BB, true, true, true, true>), dim3(nb), dim3(kBlock), 0, st, \
                       tr, rays, opt, grad_out, grad, gstride, L, aux, fwd_out, coef);                       \
    hipLaunchKernelGGL((grad_merge_kernel<FMT_SH, BB, 1024, 512, 4, true>), dim3(nb), dim3(256), 0, st, tr, rays, \
                       grad_out, L, coef, aux, grad, gstride);                                                    \
    return true;
    if (xf) {
        if (opt.format != FMT_SH) return false;
        switch (opt.basis_dim) {
            case 1: SVOXT_GATHER_XF(1)
            case 4: SVOXT_GATHER_XF(4)
            case 9: SVOXT_GATHER_XF(9)
        }
        return false;
    }
    if (opt.format == FMT_RGBA) { SVOXT_GATHER(FMT_RGBA, 0) }
    if (opt.format == FMT_SH) {
        switch (opt.basis_dim) {
            case 1: SVOXT_GATHER(FMT_SH, 1)
            case 4: SVOXT_GATHER(FMT_SH, 4)
            case 9: SVOXT_GATHER(FMT_SH, 9)
        }
    }
#undef SVOXT_GATHER
#undef SVOXT_GATHER_XF
    return false;
}

template <bool N2, bool REPLAY>
bool launch_bwd_special(const TreeDev& tr, const RaysDev& rays, const Opts& opt, int C,
                        const float* grad_out, float* grad, int gstride, RecLists L, const uint4* aux,
                        const float* fwd_out, hipStream_t st) {
    const unsigned nb = nblocks(rays.Q);
#define SVOXT_BWD(F, CC, BB)                                                                      \
    hipLaunchKernelGGL((render_bwd_kernel<F, CC, BB, N2, REPLAY>), dim3(nb), dim3(kBlock), 0, st, \
                       tr, rays, opt, grad_out, grad, gstride, L, aux, fwd_out);             \
    return true;
#define SVOXT_BWD1(CC)                                                                                          \
    hipLaunchKernelGGL((render_bwd_kernel<FMT_RGBA, CC, 0, N2, true, false, false, true>), dim3(nb), dim3(kBlock), \
                       0, st, tr, rays, opt, grad_out, grad, gstride, L, aux, fwd_out);                         \
    return true;
    if (opt.format == FMT_RGBA) {
        if (C == 3) { SVOXT_BWD(FMT_RGBA, 3, 0) }
        if constexpr (REPLAY) {
            // lists + a float per slot + the exact form asked for: one sigmoid pass instead of two
            if (L.terms != nullptr && fwd_out == nullptr) {
                if (C == 7) { SVOXT_BWD1(7) }
                if (C == 15) { SVOXT_BWD1(15) }
                if (C == 31) { SVOXT_BWD1(31) }
            }
        }
        if (C == 7) { SVOXT_BWD(FMT_RGBA, 7, 0) }
        if (C == 15) { SVOXT_BWD(FMT_RGBA, 15, 0) }
        if (C == 31) { SVOXT_BWD(FMT_RGBA, 31, 0) }
    } else if (opt.format == FMT_SH && C == 3) {
        switch (opt.basis_dim) {
            case 1: SVOXT_BWD(FMT_SH, 3, 1)
            case 4: SVOXT_BWD(FMT_SH, 3, 4)
            case 9: SVOXT_BWD(FMT_SH, 3, 9)
            case 16: SVOXT_BWD(FMT_SH, 3, 16)
            case 25: SVOXT_BWD(FMT_SH, 3, 25)
        }
    }
#undef SVOXT_BWD
#undef SVOXT_BWD1
    return false;
}

int bwd_common(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt,
               const float* grad_out, int32_t grad_cols, float* grad_features, int32_t grad_stride,
               void* workspace, int64_t workspace_bytes, const svoxt_sample_lists* lists,
               const float* fwd_out, void* stream, const char* fn) {
    int rc;
    if ((rc = check_tree(tree, fn)) || (rc = check_rays(rays, fn)) ||
        (rc = check_opts(opt, tree, fn, grad_cols > 1)))
        return rc;
    if (grad_features == nullptr && tree->M > 0) return fail(SVOXT_ERR_INVALID, "%s: grad_features is NULL", fn);
    if (rays->Q > 0 && grad_out == nullptr) return fail(SVOXT_ERR_INVALID, "%s: grad_out is NULL", fn);
    const int C = grad_cols - 1;
    if (C < 0) return fail(SVOXT_ERR_INVALID, "%s: grad_cols must be >= 1", fn);
    if (C > 0) {
        const int want = svoxt_out_data_dim(opt, tree->K);
        if (want != grad_cols) return fail(SVOXT_ERR_INVALID, "%s: grad_out columns do not match get_out_data_dim", fn);
    }
    const int gs = grad_stride > 0 ? grad_stride : tree->K;
    if (gs < tree->K) return fail(SVOXT_ERR_INVALID, "%s: grad_stride smaller than data_dim", fn);
    hipStream_t st = (hipStream_t)stream;
    if (tree->M > 0) {
        const hipError_t e = hipMemsetAsync(grad_features, 0, sizeof(float) * (size_t)tree->M * gs, st);
        if (e != hipSuccess) return fail(SVOXT_ERR_HIP, "%s: hipMemsetAsync: %s", fn, hipGetErrorString(e));
    }
    if (rays->Q == 0 || tree->M == 0) return SVOXT_OK;
    const TreeDev tr = to_dev(tree);
    const RaysDev rd = to_dev(rays);
    const Opts od = to_dev(opt);
    const bool n2 = tree->N == 2;
    bool done = false;
    const bool xf = uses_xform(tree, opt);
    if (xf && lists != nullptr && !(full_comp(opt) && xform_special(tree, opt)))
        return fail(SVOXT_ERR_UNSUPPORTED, "%s: sample lists with transformation_matrices need an SH payload on an N = 2 tree", fn);
    if (C > 0 && full_comp(opt) && xf && xform_special(tree, opt)) {
        const int64_t S = workspace != nullptr ? rec_capacity(workspace_bytes, rays->Q) : 0;
        if (lists != nullptr) {
            const int64_t need = (int64_t)lists->max_samples * rays->Q * 32;
            if (lists->coef != nullptr && tree->K <= 32 && lists->coef_bytes >= need)
                done = launch_bwd_gather(tr, rd, od, C, grad_out, grad_features, gs, lists_dev(lists, rays->Q),
                                         reinterpret_cast<const uint4*>(lists->aux), fwd_out,
                                         reinterpret_cast<float4*>(lists->coef), true, st);
            if (!done)
                done = launch_bwd_xform<true>(tr, rd, od, C, grad_out, grad_features, gs, lists_dev(lists, rays->Q),
                                              reinterpret_cast<const uint4*>(lists->aux), fwd_out, st);
        }
        else
            done = launch_bwd_xform<false>(tr, rd, od, C, grad_out, grad_features, gs,
                                           dense_lists(S > 0 ? workspace : nullptr, S, rays->Q), nullptr, nullptr, st);
    } else if (C > 0 && full_comp(opt) && !xf)
    {
        // per-ray sample lists: S entries of 8 bytes per ray, laid out rec[k][q]
        const int64_t S = workspace != nullptr ? rec_capacity(workspace_bytes, rays->Q) : 0;
        const RecLists wl = dense_lists(S > 0 ? workspace : nullptr, S, rays->Q);
        if (lists != nullptr) {
            const bool wide = opt->format == SVOXT_FORMAT_RGBA && C > 3;     // one float per slot (render_bwd_kernel<ONEPASS>)
            const RecLists ll = lists_dev(lists, rays->Q, wide ? 4 : 16);
            const uint4* laux = reinterpret_cast<const uint4*>(lists->aux);
            // coef_bytes < 0 (and no coef): the per-tile route if it can run fused, which needs no buffer
            const bool have_coef = lists->coef != nullptr &&
                                   lists->coef_bytes >= (int64_t)lists->max_samples * rays->Q * 16;
            if (n2 && tree->K <= 32 && (have_coef || lists->coef_bytes < 0))
                // (ll.terms: the exact one-kernel form's hand-over buffer; terms_state 2 = the forward filled it)
                done = launch_bwd_gather(tr, rd, od, C, grad_out, grad_features, gs, ll, laux,
                                         fwd_out, have_coef ? reinterpret_cast<float4*>(lists->coef) : nullptr, false, st,
                                         (lists->terms_state == 2 || lists->terms_state == 3) ? lists->terms_state : 1);
            if (!done) done = n2 ? launch_bwd_special<true, true>(tr, rd, od, C, grad_out, grad_features, gs, ll, laux, fwd_out, st)
                      : launch_bwd_special<false, true>(tr, rd, od, C, grad_out, grad_features, gs, ll, laux, fwd_out, st);
            if (!done) return fail(SVOXT_ERR_UNSUPPORTED, "%s: no specialised kernel for this payload", fn);
        } else {
            done = n2 ? launch_bwd_special<true, false>(tr, rd, od, C, grad_out, grad_features, gs, wl, nullptr, nullptr, st)
                      : launch_bwd_special<false, false>(tr, rd, od, C, grad_out, grad_features, gs, wl, nullptr, nullptr, st);
        }
    } else if (lists != nullptr) {
        return fail(SVOXT_ERR_UNSUPPORTED, "%s: sample lists need a specialised payload", fn);
    }
    if (!done) {
        const unsigned nb = nblocks(rays->Q);
        const size_t lds = (size_t)(kBlock / 64) * 64 * (tree->K | 1) * sizeof(float) + kBlock * sizeof(int32_t);
        if (C > 0 && lds <= 65536) {      // shaped atomics through LDS staging (default dynamic-LDS limit: 64 KiB)
            if (n2) hipLaunchKernelGGL((render_bwd_generic_staged_kernel<true>), dim3(nb), dim3(kBlock), lds, st, tr, rd, od, C, grad_out, grad_features, gs);
            else hipLaunchKernelGGL((render_bwd_generic_staged_kernel<false>), dim3(nb), dim3(kBlock), lds, st, tr, rd, od, C, grad_out, grad_features, gs);
        } else {
            // C == 0 (opacity: one value per sample, nothing to shape) or rows too wide to stage
            if (n2) hipLaunchKernelGGL((render_bwd_generic_kernel<true>), dim3(nb), dim3(kBlock), 0, st, tr, rd, od, C, grad_out, grad_features, gs);
            else hipLaunchKernelGGL((render_bwd_generic_kernel<false>), dim3(nb), dim3(kBlock), 0, st, tr, rd, od, C, grad_out, grad_features, gs);
        }
    }
    return check_launch(fn);
}

}  // namespace

extern "C" {

int svoxt_abi_version(void) { return SVOXT_ABI_VERSION; }

const char* svoxt_last_error(void) { return g_err; }

int svoxt_out_data_dim(const svoxt_options* opt, int32_t K) {
    if (opt == nullptr || K < 1) return -1;
    if (opt->format != SVOXT_FORMAT_RGBA) {
        if (opt->basis_dim < 1) return -1;
        return (K - 1) / opt->basis_dim + 1;
    }
    return K;
}

static int check_lists(const svoxt_sample_lists* l, const svoxt_options* opt, const char* fn) {
    if (l == nullptr) return fail(SVOXT_ERR_INVALID, "%s: lists is NULL", fn);
    if (l->rec == nullptr || l->aux == nullptr || l->max_samples < 8 || l->max_samples > 4096 || l->max_samples % 8 != 0)
        return fail(SVOXT_ERR_INVALID, "%s: lists need rec, aux and max_samples a multiple of 8 in [8, 4096]", fn);
    if (l->blocktab != nullptr && (l->pool_next == nullptr || l->pool_blocks < kSubPools || l->pool_blocks % kSubPools != 0 ||
                                   l->max_samples > kMaxRecBlocks * kRecBlock))
        return fail(SVOXT_ERR_INVALID, "%s: pooled lists need pool_next, pool_blocks a positive multiple of 32 and max_samples <= 512", fn);
    if (((uintptr_t)l->rec & 63u) != 0) return fail(SVOXT_ERR_INVALID, "%s: lists.rec must be 64-byte aligned", fn);
    if (opt->sigma_thresh != 0.f || opt->stop_thresh != 0.f)
        return fail(SVOXT_ERR_UNSUPPORTED, "%s: sample lists require sigma_thresh == stop_thresh == 0", fn);
    return SVOXT_OK;
}

// a recording forward starts with an empty block table and pool (pooled lists only)
static int lists_begin(const svoxt_sample_lists* l, int64_t Q, hipStream_t st, const char* fn) {
    if (l->blocktab == nullptr) return SVOXT_OK;
    const size_t n = (size_t)(rec_rays(Q) / 64) * (l->max_samples / kRecBlock) * sizeof(int32_t);
    const size_t nc = sizeof(int32_t) * kSubPools * kSubPoolStride;
    hipError_t e;
    if (reinterpret_cast<char*>(l->blocktab) + n == reinterpret_cast<char*>(l->pool_next)) {
        e = hipMemsetAsync(l->blocktab, 0xff, n + nc, st);       // counters right behind the table: one fill
    } else {
        e = hipMemsetAsync(l->blocktab, 0xff, n, st);
        if (e == hipSuccess) e = hipMemsetAsync(l->pool_next, 0xff, nc, st);
    }
    if (e != hipSuccess) return fail(SVOXT_ERR_HIP, "%s: hipMemsetAsync: %s", fn, hipGetErrorString(e));
    return SVOXT_OK;
}

static int fwd_common(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt, float* out,
                      const svoxt_sample_lists* lists, void* stream, const char* fn,
                      void* workspace = nullptr, int64_t workspace_bytes = 0, int32_t flags = 0,
                      const svoxt_sample_lists* scratch = nullptr) {
    int rc;
    if ((rc = check_tree(tree, fn)) || (rc = check_rays(rays, fn)) || (rc = check_opts(opt, tree, fn, true)))
        return rc;
    if (lists != nullptr && (rc = check_lists(lists, opt, fn))) return rc;
    if (lists != nullptr && uses_xform(tree, opt) && !(full_comp(opt) && xform_special(tree, opt)))
        return fail(SVOXT_ERR_UNSUPPORTED, "%s: sample lists with transformation_matrices need an SH payload on an N = 2 tree", fn);
    if (rays->Q == 0) return SVOXT_OK;
    if (out == nullptr) return fail(SVOXT_ERR_INVALID, "%s: out is NULL", fn);
    const int C = svoxt_out_data_dim(opt, tree->K) - 1;
    if (C < 0) return fail(SVOXT_ERR_INVALID, "%s: bad output width", fn);
    if (opt->format != SVOXT_FORMAT_RGBA && (int64_t)C * opt->basis_dim > tree->K - 1)
        return fail(SVOXT_ERR_INVALID, "%s: data_dim is not channels * basis_dim + 1", fn);
    hipStream_t st = (hipStream_t)stream;
    if (lists != nullptr && (rc = lists_begin(lists, rays->Q, st, fn))) return rc;
    const TreeDev tr = to_dev(tree);
    const RaysDev rd = to_dev(rays);
    const Opts od = to_dev(opt);
    const bool n2 = tree->N == 2;
    bool done = false;
    // two kernels (march, shade per tile) where there is room for the lists: the caller's,
    // or scratch (then the stop rule applies while marching -- those lists serve no backward)
    const bool fast = (flags & SVOXT_FWD_FAST_SIGMOID) != 0;
    const bool want_terms = lists != nullptr && lists->terms != nullptr && C == 3;
    if (fwd_split_enabled(tree, opt, want_terms) && full_comp(opt) && fwd_split_payload(tree, opt, C) &&
        (!uses_xform(tree, opt) || xform_special(tree, opt))) {
        const bool xf = uses_xform(tree, opt);
        // one bit per feature row, built for this feature table and this sigma_thresh (else ignored)
        const uint32_t* smask = (tree->sigma_mask != nullptr && tree->sigma_mask_thresh == opt->sigma_thresh)
                                    ? reinterpret_cast<const uint32_t*>(tree->sigma_mask) : nullptr;
        if (lists != nullptr) {
            uint4* aux = reinterpret_cast<uint4*>(lists->aux);
            done = n2 ? launch_fwd_split<true, false>(tr, rd, od, out, lists_dev(lists, rays->Q), aux, xf, false, st, smask)
                      : launch_fwd_split<false, false>(tr, rd, od, out, lists_dev(lists, rays->Q), aux, xf, false, st, smask);
        } else if (scratch != nullptr && smask != nullptr && opt->stop_thresh == 0.f) {
            // With stop_thresh = 0 the stop rule ends a ray only once its transmittance is exactly 0; every
            // later sample then has weight 0 * (1 - att) = 0 and the final rescale is by 1 / (1 - 0): the
            // outputs are the same without it, and the march needs no sigma value -- the bitmask will do.
            if ((rc = lists_begin(scratch, rays->Q, st, fn))) return rc;
            uint4* aux = reinterpret_cast<uint4*>(scratch->aux);
            done = n2 ? launch_fwd_split<true, false>(tr, rd, od, out, lists_dev(scratch, rays->Q), aux, xf, fast, st, smask)
                      : launch_fwd_split<false, false>(tr, rd, od, out, lists_dev(scratch, rays->Q), aux, xf, fast, st, smask);
        } else if (scratch != nullptr) {      // caller-owned lists as scratch (dense or pooled): the stop rule applies
            if ((rc = lists_begin(scratch, rays->Q, st, fn))) return rc;
            uint4* aux = reinterpret_cast<uint4*>(scratch->aux);
            done = n2 ? launch_fwd_split<true, true>(tr, rd, od, out, lists_dev(scratch, rays->Q), aux, xf, fast, st)
                      : launch_fwd_split<false, true>(tr, rd, od, out, lists_dev(scratch, rays->Q), aux, xf, fast, st);
        } else if (workspace != nullptr && rec_capacity(workspace_bytes - rec_rays(rays->Q) * 16, rays->Q) >= kRecBlock) {
            const int64_t S = rec_capacity(workspace_bytes - rec_rays(rays->Q) * 16, rays->Q);
            uint4* aux = reinterpret_cast<uint4*>(workspace);                       // aux first: rec stays 64-byte aligned
            const RecLists L = dense_lists(reinterpret_cast<char*>(workspace) + rec_rays(rays->Q) * 16, S, rays->Q);
            done = n2 ? launch_fwd_split<true, true>(tr, rd, od, out, L, aux, xf, fast, st)
                      : launch_fwd_split<false, true>(tr, rd, od, out, L, aux, xf, fast, st);
        }
        if (done) return check_launch(fn);
    }
    if (uses_xform(tree, opt)) {
        // per-leaf view rotations re-evaluate the basis per sample: specialised for SH
        // payloads on N = 2 trees, the generic kernel otherwise
        if (full_comp(opt) && xform_special(tree, opt)) {
            if (lists != nullptr)
                done = launch_fwd_xform<true>(tr, rd, od, C, out, lists_dev(lists, rays->Q),
                                              reinterpret_cast<uint4*>(lists->aux), st);
            else
                done = launch_fwd_xform<false>(tr, rd, od, C, out, RecLists{}, nullptr, st);
        }
    } else if (full_comp(opt)) {
        if (lists != nullptr) {
            uint4* aux = reinterpret_cast<uint4*>(lists->aux);
            done = n2 ? launch_fwd_special<true, true>(tr, rd, od, C, out, lists_dev(lists, rays->Q), aux, st)
                      : launch_fwd_special<false, true>(tr, rd, od, C, out, lists_dev(lists, rays->Q), aux, st);
        } else {
            done = n2 ? launch_fwd_special<true, false>(tr, rd, od, C, out, RecLists{}, nullptr, st)
                      : launch_fwd_special<false, false>(tr, rd, od, C, out, RecLists{}, nullptr, st);
        }
    }
    if (!done) {
        if (lists != nullptr) return fail(SVOXT_ERR_UNSUPPORTED, "%s: sample lists need a specialised payload", fn);
        const unsigned nb = nblocks(rays->Q);
        if (n2) hipLaunchKernelGGL((render_fwd_generic_kernel<true>), dim3(nb), dim3(kBlock), 0, st, tr, rd, od, C, out);
        else hipLaunchKernelGGL((render_fwd_generic_kernel<false>), dim3(nb), dim3(kBlock), 0, st, tr, rd, od, C, out);
    }
    return check_launch(fn);
}

int svoxt_volume_render_fwd(const svoxt_tree* tree, const svoxt_rays* rays,
                            const svoxt_options* opt, float* out, void* stream) {
    return fwd_common(tree, rays, opt, out, nullptr, stream, "svoxt_volume_render_fwd");
}

int64_t svoxt_fwd_workspace_bytes(int64_t Q, int32_t max_samples) {
    if (Q < 0 || max_samples < 1 || max_samples > 4096) return -1;
    return rec_rays(Q) * (16 + (int64_t)((max_samples + 7) / 8 * 8) * 8);
}

int svoxt_volume_render_fwd_ws(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt,
                               float* out, void* workspace, int64_t workspace_bytes, int32_t flags, void* stream) {
    if (workspace_bytes < 0) return fail(SVOXT_ERR_INVALID, "%s: negative workspace size", "svoxt_volume_render_fwd_ws");
    if (workspace != nullptr && ((uintptr_t)workspace & 63u) != 0)
        return fail(SVOXT_ERR_INVALID, "%s: workspace must be 64-byte aligned", "svoxt_volume_render_fwd_ws");
    return fwd_common(tree, rays, opt, out, nullptr, stream, "svoxt_volume_render_fwd_ws", workspace, workspace_bytes, flags);
}

int svoxt_volume_render_fwd_scratch(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt,
                                    float* out, const svoxt_sample_lists* scratch, int32_t flags, void* stream) {
    const char* fn = "svoxt_volume_render_fwd_scratch";
    if (scratch == nullptr) return fail(SVOXT_ERR_INVALID, "%s: scratch is NULL", fn);
    if (scratch->rec == nullptr || scratch->aux == nullptr || scratch->max_samples < 8 || scratch->max_samples % 8 != 0 ||
        scratch->max_samples > 4096 || ((uintptr_t)scratch->rec & 63u) != 0 ||
        (scratch->blocktab != nullptr && (scratch->pool_next == nullptr || scratch->pool_blocks < kSubPools ||
                                          scratch->pool_blocks % kSubPools != 0 ||
                                          scratch->max_samples > kMaxRecBlocks * kRecBlock)))
        return fail(SVOXT_ERR_INVALID, "%s: scratch lists are malformed (see svoxt_sample_lists)", fn);
    return fwd_common(tree, rays, opt, out, nullptr, stream, fn, nullptr, 0, flags, scratch);
}

int64_t svoxt_sigma_mask_bytes(int64_t M) { return M < 0 ? -1 : (M + 63) / 64 * 8; }

int svoxt_sigma_mask_build(const svoxt_tree* tree, float sigma_thresh, void* mask, void* stream) {
    const char* fn = "svoxt_sigma_mask_build";
    int rc;
    if ((rc = check_tree(tree, fn))) return rc;
    if (tree->M == 0) return SVOXT_OK;
    if (mask == nullptr || ((uintptr_t)mask & 7u) != 0) return fail(SVOXT_ERR_INVALID, "%s: mask is NULL or not 8-byte aligned", fn);
    const int64_t words = (tree->M + 63) / 64;
    hipLaunchKernelGGL(svoxt::sigma_mask_kernel, dim3((unsigned)((words + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       tree->features, tree->M, tree->K, sigma_thresh, reinterpret_cast<unsigned long long*>(mask));
    return check_launch(fn);
}

int svoxt_fwd_fills_terms(const svoxt_tree* tree, const svoxt_options* opt) {
    if (tree == nullptr || opt == nullptr || !svoxt_can_record(tree, opt)) return 0;
    if (uses_xform(tree, opt)) return 0;
    const int C = svoxt_out_data_dim(opt, tree->K) - 1;
    if (C != 3) return 0;
    // 3: the two-kernel forward (tile shade kernel, position-major); 2: the one-kernel forward (lane-major lines)
    return (fwd_split_enabled(tree, opt, true) && full_comp(opt) && fwd_split_payload(tree, opt, C)) ? 3 : 2;
}

int svoxt_can_record(const svoxt_tree* tree, const svoxt_options* opt) {
    if (tree == nullptr || opt == nullptr) return 0;
    if (opt->sigma_thresh != 0.f || opt->stop_thresh != 0.f || !full_comp(opt)) return 0;
    if (uses_xform(tree, opt)) return xform_special(tree, opt) ? 1 : 0;
    const int C = svoxt_out_data_dim(opt, tree->K) - 1;
    if (opt->format == SVOXT_FORMAT_RGBA) return (C == 3 || C == 7 || C == 15 || C == 31) ? 1 : 0;
    if (opt->format == SVOXT_FORMAT_SH && C == 3 && tree->K == 3 * opt->basis_dim + 1)
        return (opt->basis_dim == 1 || opt->basis_dim == 4 || opt->basis_dim == 9 || opt->basis_dim == 16 ||
                opt->basis_dim == 25) ? 1 : 0;
    return 0;
}

int svoxt_volume_render_fwd_record(const svoxt_tree* tree, const svoxt_rays* rays,
                                   const svoxt_options* opt, float* out,
                                   const svoxt_sample_lists* lists, void* stream) {
    const char* fn = "svoxt_volume_render_fwd_record";
    if (lists == nullptr) return fail(SVOXT_ERR_INVALID, "%s: lists is NULL", fn);
    return fwd_common(tree, rays, opt, out, lists, stream, fn);
}

int svoxt_volume_render_bwd_replay(const svoxt_tree* tree, const svoxt_rays* rays,
                                   const svoxt_options* opt, const float* grad_out,
                                   int32_t grad_cols, float* grad_features, int32_t grad_stride,
                                   const svoxt_sample_lists* lists, const float* fwd_out, void* stream) {
    const char* fn = "svoxt_volume_render_bwd_replay";
    int rc;
    if (grad_cols < 2) return fail(SVOXT_ERR_INVALID, "%s: grad_cols must be C+1 >= 2", fn);
    if (opt == nullptr) return fail(SVOXT_ERR_INVALID, "%s: options is NULL", fn);
    if ((rc = check_lists(lists, opt, fn))) return rc;
    return bwd_common(tree, rays, opt, grad_out, grad_cols, grad_features, grad_stride, nullptr, 0, lists, fwd_out,
                      stream, fn);
}

int svoxt_volume_render_bwd(const svoxt_tree* tree, const svoxt_rays* rays,
                            const svoxt_options* opt, const float* grad_out,
                            int32_t grad_cols, float* grad_features, int32_t grad_stride,
                            void* workspace, int64_t workspace_bytes, void* stream) {
    if (grad_cols < 2)
        return fail(SVOXT_ERR_INVALID, "%s: grad_cols must be C+1 >= 2 (use svoxt_opacity_render_bwd for C = 0)",
                    "svoxt_volume_render_bwd");
    if (workspace_bytes < 0) return fail(SVOXT_ERR_INVALID, "%s: negative workspace size", "svoxt_volume_render_bwd");
    return bwd_common(tree, rays, opt, grad_out, grad_cols, grad_features, grad_stride, workspace, workspace_bytes,
                      nullptr, nullptr, stream, "svoxt_volume_render_bwd");
}

int svoxt_opacity_render_fwd(const svoxt_tree* tree, const svoxt_rays* rays,
                             const svoxt_options* opt, float* out, void* stream) {
    const char* fn = "svoxt_opacity_render_fwd";
    int rc;
    if ((rc = check_tree(tree, fn)) || (rc = check_rays(rays, fn)) || (rc = check_opts(opt, tree, fn, false)))
        return rc;
    if (rays->Q == 0) return SVOXT_OK;
    if (out == nullptr) return fail(SVOXT_ERR_INVALID, "%s: out is NULL", fn);
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = nblocks(rays->Q);
    if (tree->N == 2) hipLaunchKernelGGL((opacity_fwd_kernel<true>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays), to_dev(opt), out);
    else hipLaunchKernelGGL((opacity_fwd_kernel<false>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays), to_dev(opt), out);
    return check_launch(fn);
}

int svoxt_opacity_render_bwd(const svoxt_tree* tree, const svoxt_rays* rays,
                             const svoxt_options* opt, const float* grad_out,
                             float* grad_features, void* stream) {
    return bwd_common(tree, rays, opt, grad_out, 1, grad_features, 0, nullptr, 0, nullptr, nullptr, stream,
                      "svoxt_opacity_render_bwd");
}

int svoxt_opacity_render_fwd_record(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt,
                                    float* out, const svoxt_sample_lists* lists, void* stream) {
    const char* fn = "svoxt_opacity_render_fwd_record";
    int rc;
    if ((rc = check_tree(tree, fn)) || (rc = check_rays(rays, fn)) || (rc = check_opts(opt, tree, fn, false)) ||
        (rc = check_lists(lists, opt, fn))) return rc;
    if (rays->Q == 0) return SVOXT_OK;
    if (out == nullptr) return fail(SVOXT_ERR_INVALID, "%s: out is NULL", fn);
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = nblocks(rays->Q);
    if ((rc = lists_begin(lists, rays->Q, st, fn))) return rc;
    const RecLists L = lists_dev(lists, rays->Q);
    uint4* aux = reinterpret_cast<uint4*>(lists->aux);
    if (tree->N == 2) hipLaunchKernelGGL((opacity_fwd_kernel<true, true>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays), to_dev(opt), out, L, aux);
    else hipLaunchKernelGGL((opacity_fwd_kernel<false, true>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays), to_dev(opt), out, L, aux);
    return check_launch(fn);
}

int svoxt_opacity_render_bwd_replay(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt,
                                    const float* grad_out, float* grad_features, int32_t grad_stride,
                                    const svoxt_sample_lists* lists, void* stream) {
    const char* fn = "svoxt_opacity_render_bwd_replay";
    int rc;
    if ((rc = check_tree(tree, fn)) || (rc = check_rays(rays, fn)) || (rc = check_opts(opt, tree, fn, false)) ||
        (rc = check_lists(lists, opt, fn))) return rc;
    if (grad_features == nullptr && tree->M > 0) return fail(SVOXT_ERR_INVALID, "%s: grad_features is NULL", fn);
    if (rays->Q > 0 && grad_out == nullptr) return fail(SVOXT_ERR_INVALID, "%s: grad_out is NULL", fn);
    const int gs = grad_stride > 0 ? grad_stride : tree->K;
    if (gs < tree->K) return fail(SVOXT_ERR_INVALID, "%s: grad_stride smaller than data_dim", fn);
    hipStream_t st = (hipStream_t)stream;
    if (tree->M > 0) {
        const hipError_t e = hipMemsetAsync(grad_features, 0, sizeof(float) * (size_t)tree->M * gs, st);
        if (e != hipSuccess) return fail(SVOXT_ERR_HIP, "%s: hipMemsetAsync: %s", fn, hipGetErrorString(e));
    }
    if (rays->Q == 0 || tree->M == 0) return SVOXT_OK;
    const unsigned nb = nblocks(rays->Q);
    const RaysDev rd = to_dev(rays);
    const RecLists L = lists_dev(lists, rays->Q);
    uint4* aux = reinterpret_cast<uint4*>(lists->aux);
    if (tree->N == 2) hipLaunchKernelGGL((opacity_walk_kernel<true>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), rd, to_dev(opt), grad_out, grad_features, gs, L, aux);
    else hipLaunchKernelGGL((opacity_walk_kernel<false>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), rd, to_dev(opt), grad_out, grad_features, gs, L, aux);
    hipLaunchKernelGGL((opacity_merge_kernel<1024, 4>), dim3(nb), dim3(256), 0, st, rd, L, aux, grad_features, gs, (int)tree->K - 1);
    return check_launch(fn);
}

int svoxt_render_depth(const svoxt_tree* tree, const svoxt_rays* rays,
                       const svoxt_options* opt, float* depth, void* stream) {
    const char* fn = "svoxt_render_depth";
    int rc;
    if ((rc = check_tree(tree, fn)) || (rc = check_rays(rays, fn)) || (rc = check_opts(opt, tree, fn, false)))
        return rc;
    if (rays->Q == 0) return SVOXT_OK;
    if (depth == nullptr) return fail(SVOXT_ERR_INVALID, "%s: depth is NULL", fn);
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = nblocks(rays->Q);
    if (tree->N == 2) hipLaunchKernelGGL((depth_kernel<true>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays), to_dev(opt), depth);
    else hipLaunchKernelGGL((depth_kernel<false>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays), to_dev(opt), depth);
    return check_launch(fn);
}

int svoxt_query_fwd(const svoxt_tree* tree, const float* points, int64_t Q,
                    float* values, int64_t* node_ids, int64_t* data_ids,
                    uint8_t* hit_mask, void* stream) {
    const char* fn = "svoxt_query_fwd";
    int rc;
    if ((rc = check_tree(tree, fn))) return rc;
    if (Q < 0) return fail(SVOXT_ERR_INVALID, "%s: negative point count", fn);
    if (Q == 0) return SVOXT_OK;
    if (points == nullptr || values == nullptr || node_ids == nullptr || data_ids == nullptr)
        return fail(SVOXT_ERR_INVALID, "%s: points / values / node_ids / data_ids is NULL", fn);
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = nblocks(Q);
    if (tree->N == 2) hipLaunchKernelGGL((query_fwd_kernel<true>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), points, Q, values, node_ids, data_ids, hit_mask);
    else hipLaunchKernelGGL((query_fwd_kernel<false>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), points, Q, values, node_ids, data_ids, hit_mask);
    return check_launch(fn);
}

int svoxt_query_bwd(const svoxt_tree* tree, const float* points, int64_t Q,
                    const float* grad_out, float* grad_features, void* stream) {
    const char* fn = "svoxt_query_bwd";
    int rc;
    if ((rc = check_tree(tree, fn))) return rc;
    if (Q < 0) return fail(SVOXT_ERR_INVALID, "%s: negative point count", fn);
    if (grad_features == nullptr && tree->M > 0) return fail(SVOXT_ERR_INVALID, "%s: grad_features is NULL", fn);
    hipStream_t st = (hipStream_t)stream;
    if (tree->M > 0) {
        const hipError_t e = hipMemsetAsync(grad_features, 0, sizeof(float) * (size_t)tree->M * tree->K, st);
        if (e != hipSuccess) return fail(SVOXT_ERR_HIP, "%s: hipMemsetAsync: %s", fn, hipGetErrorString(e));
    }
    if (Q == 0 || tree->M == 0) return SVOXT_OK;
    if (points == nullptr || grad_out == nullptr) return fail(SVOXT_ERR_INVALID, "%s: points / grad_out is NULL", fn);
    const unsigned nb = nblocks(Q);
    if (tree->N == 2) hipLaunchKernelGGL((query_bwd_kernel<true>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), points, Q, grad_out, grad_features);
    else hipLaunchKernelGGL((query_bwd_kernel<false>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), points, Q, grad_out, grad_features);
    return check_launch(fn);
}

int svoxt_count_fwd(const svoxt_tree* tree, const svoxt_rays* rays,
                    const svoxt_options* opt, int64_t* counters, void* stream) {
    const char* fn = "svoxt_count_fwd";
    int rc;
    if ((rc = check_tree(tree, fn)) || (rc = check_rays(rays, fn)) || (rc = check_opts(opt, tree, fn, false)))
        return rc;
    if (counters == nullptr) return fail(SVOXT_ERR_INVALID, "%s: counters is NULL", fn);
    if (rays->Q == 0) return SVOXT_OK;
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = nblocks(rays->Q);
    unsigned long long* c = reinterpret_cast<unsigned long long*>(counters);
    TreeDev tr = to_dev(tree);
    tr.accel = nullptr;   // the counters are the reference's: levels of the plain root descent
    if (tree->N == 2) hipLaunchKernelGGL((count_fwd_kernel<true>), dim3(nb), dim3(kBlock), 0, st, tr, to_dev(rays), to_dev(opt), c);
    else hipLaunchKernelGGL((count_fwd_kernel<false>), dim3(nb), dim3(kBlock), 0, st, tr, to_dev(rays), to_dev(opt), c);
    return check_launch(fn);
}

int svoxt_set_bwd_counters(int64_t* counters) {
    g_bwd_counters = counters;
    return SVOXT_OK;
}

int svoxt_count_touched(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt,
                        uint8_t* row_mask, uint8_t* tree_mask, int64_t* longest, void* stream) {
    const char* fn = "svoxt_count_touched";
    int rc;
    if ((rc = check_tree(tree, fn)) || (rc = check_rays(rays, fn)) || (rc = check_opts(opt, tree, fn, false)))
        return rc;
    if (row_mask == nullptr || tree_mask == nullptr || longest == nullptr)
        return fail(SVOXT_ERR_INVALID, "%s: row_mask / tree_mask / longest is NULL", fn);
    if (rays->Q == 0) return SVOXT_OK;
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = nblocks(rays->Q);
    const uint32_t n_slots = (uint32_t)(tree->n_internal * tree->N * tree->N * tree->N);
    unsigned long long* lg = reinterpret_cast<unsigned long long*>(longest);
    if (tree->N == 2) hipLaunchKernelGGL((count_touched_kernel<true>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays), to_dev(opt), row_mask, tree_mask, n_slots, lg);
    else hipLaunchKernelGGL((count_touched_kernel<false>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays), to_dev(opt), row_mask, tree_mask, n_slots, lg);
    return check_launch(fn);
}

int64_t svoxt_bwd_workspace_bytes(int64_t Q, int32_t max_samples) {
    if (Q < 0 || max_samples < 0) return -1;
    return rec_rays(Q) * (int64_t)((max_samples + 7) / 8 * 8) * 8;
}

int64_t svoxt_query_leaves_workspace_bytes(int64_t n_slots) {
    if (n_slots < 0) return -1;
    return (int64_t)sizeof(int32_t) * ((n_slots + kSeg - 1) / kSeg + 1);
}

int svoxt_query_leaves(const uint8_t* hit_mask, int64_t n_slots, int32_t N, int64_t* leaf_node,
                       int64_t* count, void* workspace, void* stream) {
    const char* fn = "svoxt_query_leaves";
    if (n_slots < 0 || N < 2) return fail(SVOXT_ERR_INVALID, "%s: bad extents", fn);
    if (count == nullptr) return fail(SVOXT_ERR_INVALID, "%s: count is NULL", fn);
    hipStream_t st = (hipStream_t)stream;
    if (n_slots == 0) {
        const hipError_t e = hipMemsetAsync(count, 0, sizeof(int64_t), st);
        return e == hipSuccess ? SVOXT_OK : fail(SVOXT_ERR_HIP, "%s: %s", fn, hipGetErrorString(e));
    }
    if (hit_mask == nullptr || leaf_node == nullptr || workspace == nullptr)
        return fail(SVOXT_ERR_INVALID, "%s: hit_mask / leaf_node / workspace is NULL", fn);
    const int64_t nseg = (n_slots + kSeg - 1) / kSeg;
    if (nseg > 2147483647LL) return fail(SVOXT_ERR_INVALID, "%s: mask too large", fn);
    int32_t* seg = reinterpret_cast<int32_t*>(workspace);
    hipLaunchKernelGGL(leaves_count_kernel, dim3((unsigned)nseg), dim3(kBlock), 0, st, hit_mask, n_slots, seg);
    hipLaunchKernelGGL(leaves_scan_kernel, dim3(1), dim3(kBlock), 0, st, seg, (int)nseg, count);
    hipLaunchKernelGGL(leaves_scatter_kernel, dim3((unsigned)nseg), dim3(kBlock), 0, st, hit_mask, n_slots, (int)N, seg, leaf_node);
    return check_launch(fn);
}

int svoxt_compact_rows(const float* src, int64_t M, int32_t K, int32_t stride, float* dst, void* stream) {
    const char* fn = "svoxt_compact_rows";
    if (M < 0 || K < 1 || stride < K) return fail(SVOXT_ERR_INVALID, "%s: bad extents", fn);
    if (M == 0) return SVOXT_OK;
    if (src == nullptr || dst == nullptr) return fail(SVOXT_ERR_INVALID, "%s: src / dst is NULL", fn);
    typedef float v4f __attribute__((ext_vector_type(4)));
    const bool vec = K % 4 == 0 && stride % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
    const int64_t n = vec ? M * (K / 4) : M * K;
    const int64_t want = (n + kBlock - 1) / kBlock;
    const unsigned nb = (unsigned)(want < 16384 ? want : 16384);
    if (vec)
        hipLaunchKernelGGL((compact_rows_kernel<v4f>), dim3(nb), dim3(kBlock), 0, (hipStream_t)stream,
                           reinterpret_cast<const v4f*>(src), n, (int)(K / 4), (int)(stride / 4), reinterpret_cast<v4f*>(dst));
    else
        hipLaunchKernelGGL((compact_rows_kernel<float>), dim3(nb), dim3(kBlock), 0, (hipStream_t)stream,
                           src, n, (int)K, (int)stride, dst);
    return check_launch(fn);
}

int64_t svoxt_accel_bytes(int32_t log2_res, int64_t n_internal) {
    if (log2_res < 1 || log2_res > 8 || n_internal < 0) return -1;
    return ((int64_t)sizeof(uint2) << (3 * log2_res)) + (int64_t)sizeof(uint2) * 8 * n_internal;
}

int svoxt_accel_build(const svoxt_tree* tree, int32_t log2_res, void* cells, void* stream) {
    const char* fn = "svoxt_accel_build";
    int rc;
    if ((rc = check_tree(tree, fn))) return rc;
    if (tree->N != 2) return fail(SVOXT_ERR_UNSUPPORTED, "%s: the acceleration grid exists for N == 2 only", fn);
    if (log2_res < 1 || log2_res > 8) return fail(SVOXT_ERR_INVALID, "%s: log2_res must be in [1, 8]", fn);
    if (cells == nullptr) return fail(SVOXT_ERR_INVALID, "%s: cells is NULL", fn);
    TreeDev tr = to_dev(tree);
    tr.accel = nullptr;
    const unsigned n = 1u << (3 * log2_res);
    hipLaunchKernelGGL(accel_build_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       (hipStream_t)stream, tr, (int)log2_res, reinterpret_cast<uint2*>(cells));
    const int64_t slots = tree->n_internal * 8;
    if (slots > 0)
        hipLaunchKernelGGL(accel_nodes_kernel, dim3((unsigned)((slots + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           (hipStream_t)stream, tr.child, tr.data, slots, reinterpret_cast<uint2*>(cells) + n);
    return check_launch(fn);
}

}  // extern "C"
