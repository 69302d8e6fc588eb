// svoxt_lists.h -- the launch shape shared by the render kernels and the per-ray sample lists a
// forward records for its backward (or as scratch for its own shade kernel): 8-byte records in
// blocks of 8 list positions x 64 rays, dense or handed out from a pool through a per-tile table,
// their LDS staging, and the backward's per-sample hand-over.  Included by svoxt_kernels.hip only
// (one translation unit: the kernels are templates launched from its C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "svoxt_device.h"

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
// REC records every sample with sigma > 0 whatever the thresholds (the backward ignores both,
// rt_kernel.cu:382,456; r05: the forward's thresholds then only decide what is COMPOSITED); the march is then not cut short when the
// transmittance underflows to exactly 0 -- the remaining samples have weight
// 0 and leave the output bits unchanged, but they belong in the list.
constexpr uint32_t kRecOverflow = 0x80000000u;
// svoxt_sample_lists.tile_state (fwd_roles_kernel, svoxt_fwd_kernels.h): per-tile states (rounded up to even), 16 counter
// slots of 32 words, 8 queues (one per XCD) of one 64-bit entry per tile
constexpr int kRoleXcds = 8;
__host__ __device__ inline int64_t roles_even(int64_t tiles) { return (tiles + 1) & ~(int64_t)1; }   // (the 64-bit queue entries behind it stay aligned)
__host__ __device__ inline int64_t roles_state_words(int64_t tiles) { return roles_even(tiles) + 16 * 32 + (int64_t)kRoleXcds * 2 * tiles; }
// (r05) aux[q].w between a recording forward's shade and its tail launch: the shade ended the ray's COMPOSITING by the
// stop rule (T <= stop_thresh, rt_kernel.cu:313-319) and has written the pixel; the forward's tail leaves the ray alone,
// the overflow flag stays for the backward's tail (which marches every sample with sigma > 0 the list could not hold).
// 0 otherwise; the backward's tail-only launch later overwrites .z / .w of overflowed rays with its pass-1 results.
constexpr uint32_t kAuxStopped = 1u;

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

// The same reads for a consumer that runs INSIDE the kernel that wrote the lists (fwd_roles_kernel: the
// shading workgroup of a tile starts as soon as the tile's march has published it): relaxed agent-scope
// atomic loads -- global_load ... sc1: served by the XCD's L2, never by a stale line of this CU's vector
// cache (the block table's lines are shared between neighbouring tiles, which other workgroups of the CU
// may have read before this tile's entries were written).
__device__ __forceinline__ uint2 rec_get_coherent(const uint2* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ uint4 aux_get_coherent(const uint4* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
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
template <bool COHERENT = false>
__device__ __forceinline__ int32_t rec_tab_reg(const RecLists& L, int64_t tile, int lane) {
    const int nb = L.S >> 3;
    if (L.tab == nullptr || lane >= nb) return -1;
    if constexpr (COHERENT) return __hip_atomic_load(L.tab + tile * (int64_t)nb + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return L.tab[tile * (int64_t)nb + lane];
}
__device__ __forceinline__ int64_t rec_block_u(const RecLists& L, int32_t tabreg, int64_t tile, int b /* wavefront-uniform */) {
    if (L.tab == nullptr) return tile * (int64_t)(L.S >> 3) + b;
    return (int64_t)__builtin_amdgcn_readlane(tabreg, __builtin_amdgcn_readfirstlane(b));
}
// (r04) Pooled lists: word 1 of the pool's counter block (the counters are 16 words apart; lists_begin's fill leaves
// -1) becomes 1 when a recording kernel fills a list to its capacity.  The tail launches (rays marched on past their
// lists: render_fwd_kernel<RESUME>, render_bwd_kernel's tail-only form, fwd_finish_kernel) read it with one scalar load
// and end at once when no ray of the batch overflowed -- the common case -- instead of fetching every ray's list length
// to find that out.  Conservative: a stop rule that ends a ray at its last record leaves the word set.
constexpr int kPoolOverflowWord = 1;
__device__ __forceinline__ void note_overflow(const RecLists& L) {
    if (L.pool_next != nullptr) L.pool_next[kPoolOverflowWord] = 1;
}
__device__ __forceinline__ bool no_ray_overflowed(const RecLists& L) {       // (uniform; false where nobody keeps the word)
    if (L.pool_next == nullptr) return false;
    const int32_t* p = L.pool_next + kPoolOverflowWord;
    int32_t w;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(p) : "memory");
    return w != 1;
}
// Pooled lists: a tile none of whose rays recorded a sample never took its first block -- entry 0 of its table is
// still the -1 lists_begin left.  ONE scalar load (the address is the workgroup's) answers that before a kernel
// that works per tile has requested anything else: on an 800 x 800 view of the depth-8 shell two tiles in three
// are empty, and what their workgroups cost is what a per-tile kernel gets back (see fwd_roles_kernel).
__device__ __forceinline__ bool tile_never_recorded(const RecLists& L, int64_t tile) {
    if (L.tab == nullptr) return false;
    const int32_t* p = L.tab + tile * (int64_t)(L.S >> 3);
    int32_t b0;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b0) : "s"(p) : "memory");
    return b0 == -1;
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
__device__ __forceinline__ void rec_tab_init(int32_t* ltab) {       // (ltab: this WAVEFRONT's table)
    static_assert(kMaxRecBlocks <= 64, "one lane per table entry");
    if ((int)(threadIdx.x & 63) < kMaxRecBlocks) ltab[threadIdx.x & 63] = -1;
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

// A workgroup barrier that orders LDS traffic only: global loads a wavefront has already requested (the
// next round's record and hand-over) stay in flight across it, where __syncthreads() waits for them
// (its release fence covers global memory: s_waitcnt vmcnt(0)).  For barriers between phases that hand
// data over through LDS alone.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// A checksum of a tile's lists as its march wrote them (r04): handed to the shading workgroup of fwd_roles_kernel with the
// tile id, recomputed there from what that workgroup LOADED -- a stale line (the hand-over rests on measured cache
// behaviour, not on the memory model: see fwd_roles_kernel) then shows as a mismatch and the tile is left to the
// fallback launch instead of being shaded from wrong records.  Per ray: the XOR over its records of rec_hash(k, row,
// delta_t) (XOR, so that the wavefronts of the consumer, which each load some list positions of all 64 rays, can fold
// what they loaded by themselves and combine per ray afterwards); per tile: every ray's word, with its list length |
// overflow flag mixed in, put through a bijection of its lane with full avalanche -- a difference in ONE ray always
// shows, differences in several cancel with probability 2^-32 -- and XORed over the 64 lanes.
__device__ __forceinline__ uint32_t rec_hash(int k, uint32_t idx, uint32_t dt_bits) {
    const uint32_t kk = (uint32_t)k;
    return __builtin_rotateleft32(idx + kk * 0x9E3779u, kk & 31u) ^ __builtin_rotateleft32(dt_bits ^ kk, (5u * kk + 11u) & 31u);
}
__device__ __forceinline__ uint32_t wave_xor(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v ^= (uint32_t)__shfl_xor((int)v, off, 64);
    return v;
}
// (rays_xor: XOR of rec_hash over ALL records of this lane's ray; ax: the ray's aux.x; 0 / 0 for a lane without a ray.)
// Per lane a bijection with full avalanche (murmur3's finalizer on the word offset by the lane): a first version that
// only multiplied by an odd number of the lane let the SAME small difference in every lane -- which is what the test's
// "stale" records are -- cancel in the XOR over the lanes once in ~2 000 tiles (measured: one tile in 690 shaded from
// perturbed records and marked good); with the finalizer 0 of 2 000 000 simulated tiles per flipped bit.
__device__ __forceinline__ uint32_t tile_checksum(uint32_t rays_xor, uint32_t ax, int lane) {
    uint32_t h = (rays_xor ^ (ax * 0x9E3779B1u)) + (uint32_t)lane * 0x9E3779B1u;
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return wave_xor(h);
}

// CHECK instances of the per-tile backwards (svoxt_set_bwd_check; r04): every index into an LDS array, a private
// array that is not unrolled away, the lists' block pool and the feature / gradient tables is compared with its
// extent; a violation adds one to bad[site] and the index becomes 0, so that the checked run itself cannot leave its
// arrays.  The production instances (CHECK = false) compile to the bare index: same code as without the call.
constexpr int kChkBase = 2;              // words [0, 2) of the counter block are svoxt_set_bwd_counters'
constexpr int kChkSites = 30;
template <bool CHECK, typename I>
__device__ __forceinline__ I chk(I i, int64_t n, unsigned long long* __restrict__ bad, int site) {
    if constexpr (CHECK) {
        if ((uint64_t)(int64_t)i >= (uint64_t)n) {
            atomicAdd(bad + kChkBase + site, 1ull);
            return (I)0;
        }
    }
    return i;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

}  // namespace svoxt
