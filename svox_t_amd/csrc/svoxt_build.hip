// svoxt_build.hip -- octree construction from a point cloud on gfx950 (N = 2),
// and construct_tree.  C ABI: svoxt_build_* / svoxt_construct_tree (include/svoxt.h).
//
// What it replaces: depth-1 rounds of `tree[points].refine()` (helpers.py:101-109
// -> query_vertical, svox_kernel.cu:45-94, 240-324 -> N3Tree.refine, svox.py:488-560)
// followed by construct_tree (svox_kernel.cu:110-121).  Every round of that loop
// descends all points through the tree built so far, marks the leaves they land
// in, numbers the marked leaves and appends one node per leaf.  All points start
// at the root and every round splits exactly the leaves that hold a point, so
// after r rounds the internal nodes are the cells of side 2^-l (l <= r) that
// contain a point -- the tree is a function of the points' l-bit cell codes alone:
//
//   cell code of a point at level l: the top l bits of its three tree-space
//   coordinates, interleaved (x, y, z) most significant first -- the same bits
//   the reference's `p *= N; floor; p -= floor` descent extracts (common.cuh:
//   63-100; exact in fp32), taken here from a 22-bit fixed-point conversion.
//
//   node numbering: refine() appends the new nodes of a round in the order of the
//   leaf list it is given; with the sorted unique-leaf list (packed id node*8+slot)
//   the nodes of level l are numbered in increasing cell-code order, after all
//   nodes of levels < l.
//
// Pipeline (all on one stream, no host involvement until the node count is read):
//   mark     occupied cells of level D-1: one plain byte store per point into a byte
//            map (D <= 9; a whole resident grid hammering the few words of a bitmap
//            with atomicOr took 112 us for 500 k points, the byte stores take 6),
//            packed to bits afterwards; at D = 10 (128 MiB of bytes) atomicOr on bits
//   reduce   level l bitmap = "byte of level l+1 bitmap != 0"   (8 children = 1 byte);
//            levels of <= 1024 words in one single-workgroup launch
//   scan     exclusive prefix popcounts per level -> rank of every occupied cell,
//            level offsets, n_internal                          (3 launches for all levels)
//   emit     one thread per bitmap byte (8 sibling cells): child / data /
//            parent_depth rows of its nodes
//   assign   data[leaf of point i] = min i                      (atomicMin per point)
// HBM-bound integer work: 12 B per point read twice, bitmaps of 8^(D-1)/8 bytes,
// 72 B written per node.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svoxt.h"
#include "svoxt_device.h"
#include "svoxt_host.h"

#pragma clang fp contract(off)

namespace svoxt {

constexpr int kBuildBlock = 256;
constexpr int kMaxLevels = 10;          // levels 0..9 can hold nodes (depth <= 10)
constexpr int kScanSeg = 1024;          // bitmap words per scan segment (4 per thread)
constexpr int kByteMapMaxDepth = 9;     // level D-1 <= 8: a byte map of <= 16 MiB
constexpr int kSmallLevel = 5;          // levels <= 5 (<= 1024 words) are reduced by one workgroup

// Where each level's bitmap / rank words and scan segments sit in the workspace.
struct BuildLevels {
    int32_t depth;                       // D
    int64_t word_off[kMaxLevels];        // levels 1..D-1 (level 0 is the root: no bitmap)
    int64_t words[kMaxLevels];
    int64_t seg_off[kMaxLevels];
    int64_t total_words, total_segs;
    int64_t bytemap_bytes;               // byte map of level D-1 (0: mark bits with atomics instead)
};

// workspace: [level_off int32[16]] [bitmap u32[total_words]] [rank u32[total_words]] [segsum u32[total_segs]]
struct BuildSpace {
    int32_t* level_off;                  // level_off[l] = index of the first node of level l; [D] = n_internal
    uint32_t* bitmap;
    uint32_t* rank;
    uint32_t* segsum;
    uint8_t* bytemap;
};

__host__ inline BuildLevels make_levels(int depth) {
    BuildLevels lv{};
    lv.depth = depth;
    int64_t w = 0, s = 0;
    for (int l = 1; l < depth; ++l) {
        const int64_t cells = 1ll << (3 * l);
        lv.word_off[l] = w;
        lv.words[l] = cells >= 32 ? cells / 32 : 1;
        lv.seg_off[l] = s;
        w += lv.words[l];
        s += (lv.words[l] + kScanSeg - 1) / kScanSeg;
    }
    lv.total_words = w;
    lv.total_segs = s;
    lv.bytemap_bytes = (depth >= 2 && depth <= kByteMapMaxDepth) ? (int64_t)1 << (3 * (depth - 1)) : 0;
    if (lv.bytemap_bytes > 0 && lv.bytemap_bytes < 32) lv.bytemap_bytes = 32;     // one bitmap word's worth
    return lv;
}

__host__ inline int64_t space_bytes(const BuildLevels& lv) {
    return 64 + 4 * (2 * lv.total_words + lv.total_segs) + 64 + lv.bytemap_bytes;
}

__host__ inline BuildSpace carve(const BuildLevels& lv, void* workspace) {
    BuildSpace sp;
    char* base = static_cast<char*>(workspace);
    sp.level_off = reinterpret_cast<int32_t*>(base);
    sp.bitmap = reinterpret_cast<uint32_t*>(base + 64);
    sp.rank = sp.bitmap + lv.total_words;
    sp.segsum = sp.rank + lv.total_words;
    // 16-byte aligned: 64 + a multiple of 4, rounded up
    const int64_t off = (64 + 4 * (2 * lv.total_words + lv.total_segs) + 15) / 16 * 16;
    sp.bytemap = reinterpret_cast<uint8_t*>(base + off);
    return sp;
}

// 3*levels-bit cell code of a world-space point (levels <= 10)
__device__ __forceinline__ uint32_t point_code(const float* __restrict__ offset, const float* __restrict__ scaling,
                                               const float* __restrict__ p, int levels) {
    // transform_coord (common.cuh:45-51) and the clamp of query_single_from_root (:38-42)
    const float px = fmaxf(0.f, fminf(kClampHi, offset[0] + scaling[0] * p[0]));
    const float py = fmaxf(0.f, fminf(kClampHi, offset[1] + scaling[1] * p[1]));
    const float pz = fmaxf(0.f, fminf(kClampHi, offset[2] + scaling[2] * p[2]));
    const float S = (float)(1 << kFixBits);
    const uint32_t ux = (uint32_t)(px * S), uy = (uint32_t)(py * S), uz = (uint32_t)(pz * S);
    uint32_t code = 0;
    for (int k = 1; k <= levels; ++k) {
        const int sh = kFixBits - k;
        code = (code << 3) | (((ux >> sh) & 1u) << 2) | (((uy >> sh) & 1u) << 1) | ((uz >> sh) & 1u);
    }
    return code;
}

// number of occupied cells of level l with a code below c
__device__ __forceinline__ uint32_t cell_rank(const BuildLevels& lv, const BuildSpace& sp, int l, uint32_t c) {
    const int64_t w = lv.word_off[l] + (c >> 5);
    return sp.rank[w] + __popc(sp.bitmap[w] & ((1u << (c & 31u)) - 1u));
}

__global__ void __launch_bounds__(kBuildBlock)
build_mark_bits_kernel(const float* __restrict__ points, int64_t P, const float* __restrict__ offset,
                       const float* __restrict__ scaling, int levels, uint32_t* __restrict__ bitmap) {
    const int64_t i = (int64_t)blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= P) return;
    const uint32_t c = point_code(offset, scaling, points + 3 * i, levels);
    atomicOr(bitmap + (c >> 5), 1u << (c & 31u));
}

// all writers store the same value: no atomic needed, the stores merge in L2
__global__ void __launch_bounds__(kBuildBlock)
build_mark_bytes_kernel(const float* __restrict__ points, int64_t P, const float* __restrict__ offset,
                        const float* __restrict__ scaling, int levels, uint8_t* __restrict__ bytemap) {
    const int64_t i = (int64_t)blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= P) return;
    bytemap[point_code(offset, scaling, points + 3 * i, levels)] = 1;
}

// bitmap[t] bit j = (bytemap[32 t + j] != 0)
__global__ void __launch_bounds__(kBuildBlock)
build_pack_kernel(const uint8_t* __restrict__ bytemap, uint32_t* __restrict__ bitmap, int64_t words) {
    const int64_t t = (int64_t)blockIdx.x * kBuildBlock + threadIdx.x;
    if (t >= words) return;
    const uint4* src = reinterpret_cast<const uint4*>(bytemap + 32 * t);
    uint32_t out = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint4 v = src[h];
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t w = w4[q];
            const uint32_t nib = ((w & 0x000000ffu) ? 1u : 0u) | ((w & 0x0000ff00u) ? 2u : 0u) |
                                 ((w & 0x00ff0000u) ? 4u : 0u) | ((w & 0xff000000u) ? 8u : 0u);
            out |= nib << (h * 16 + q * 4);
        }
    }
    bitmap[t] = out;
}

// word t of the coarser level: bit j = (byte 32 t + j of the finer level's bitmap != 0)
__device__ __forceinline__ uint32_t reduce_word(const uint32_t* __restrict__ fine, int64_t t, int64_t coarse_cells) {
    uint32_t out = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int64_t cell0 = t * 32 + q * 4;
        if (cell0 >= coarse_cells) break;
        const uint32_t w = fine[cell0 >> 2];
        const uint32_t nib = ((w & 0x000000ffu) ? 1u : 0u) | ((w & 0x0000ff00u) ? 2u : 0u) |
                             ((w & 0x00ff0000u) ? 4u : 0u) | ((w & 0xff000000u) ? 8u : 0u);
        out |= nib << (q * 4);
    }
    return out;
}

__global__ void __launch_bounds__(kBuildBlock)
build_reduce_kernel(const uint32_t* __restrict__ fine, uint32_t* __restrict__ coarse, int64_t coarse_words,
                    int64_t coarse_cells) {
    const int64_t t = (int64_t)blockIdx.x * kBuildBlock + threadIdx.x;
    if (t < coarse_words) coarse[t] = reduce_word(fine, t, coarse_cells);
}

// levels `from` .. 1 in one workgroup (each at most 1024 words)
__global__ void __launch_bounds__(kBuildBlock)
build_reduce_small_kernel(BuildLevels lv, BuildSpace sp, int from) {
    for (int l = from; l >= 1; --l) {
        const uint32_t* fine = sp.bitmap + lv.word_off[l + 1];
        uint32_t* coarse = sp.bitmap + lv.word_off[l];
        for (int64_t t = threadIdx.x; t < lv.words[l]; t += kBuildBlock)
            coarse[t] = reduce_word(fine, t, (int64_t)1 << (3 * l));
        __syncthreads();                  // the next level reads what this one wrote
    }
}

__device__ __forceinline__ int level_of_seg(const BuildLevels& lv, int64_t seg) {
    int l = lv.depth - 1;
    while (l > 1 && seg < lv.seg_off[l]) --l;
    return l;
}

// popcount of every scan segment, all levels in one launch
__global__ void __launch_bounds__(kBuildBlock)
build_segsum_kernel(BuildLevels lv, BuildSpace sp) {
    __shared__ uint32_t wsum[kBuildBlock / 64];
    const int64_t seg = blockIdx.x;
    const int l = level_of_seg(lv, seg);
    const int64_t w0 = (seg - lv.seg_off[l]) * kScanSeg;
    uint32_t c = 0;
    for (int i = threadIdx.x; i < kScanSeg; i += kBuildBlock)
        if (w0 + i < lv.words[l]) c += __popc(sp.bitmap[lv.word_off[l] + w0 + i]);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < kBuildBlock / 64; ++w) tot += wsum[w];
        sp.segsum[seg] = tot;
    }
}

// exclusive scan of one value per thread over the workgroup; *total = the sum
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t mine, uint32_t* wsum /*[kBuildBlock/64 + 1]*/,
                                                         uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    __syncthreads();                      // wsum may still be read from a previous call
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int w = 0; w < kBuildBlock / 64; ++w) { if (w < wave) before += wsum[w]; all += wsum[w]; }
    *total = all;
    return before + incl - mine;
}

// one workgroup: per level, exclusive scan of its segment sums in place; level offsets; node count
__global__ void __launch_bounds__(kBuildBlock)
build_levelscan_kernel(BuildLevels lv, BuildSpace sp, int64_t* __restrict__ n_internal) {
    __shared__ uint32_t wsum[kBuildBlock / 64];
    uint32_t first = 1;                   // level 0 holds the root alone
    if (threadIdx.x == 0) { sp.level_off[0] = 0; sp.level_off[1] = 1; }
    for (int l = 1; l < lv.depth; ++l) {
        const int64_t nseg = (lv.words[l] + kScanSeg - 1) / kScanSeg;
        uint32_t* s = sp.segsum + lv.seg_off[l];
        const int64_t per = (nseg + kBuildBlock - 1) / kBuildBlock;
        const int64_t lo = (int64_t)threadIdx.x * per;
        const int64_t hi = lo + per < nseg ? lo + per : nseg;
        uint32_t sum = 0;
        for (int64_t i = lo; i < hi; ++i) sum += s[i];
        uint32_t level_total;
        uint32_t run = block_exclusive_scan(sum, wsum, &level_total);
        for (int64_t i = lo; i < hi; ++i) { const uint32_t v = s[i]; s[i] = run; run += v; }
        first += level_total;
        if (threadIdx.x == 0) sp.level_off[l + 1] = (int32_t)first;
    }
    if (threadIdx.x == 0) *n_internal = (int64_t)first;
}

// rank[w] = occupied cells of the level in words before w (all levels in one launch)
__global__ void __launch_bounds__(kBuildBlock)
build_rank_kernel(BuildLevels lv, BuildSpace sp) {
    __shared__ uint32_t wsum[kBuildBlock / 64];
    const int64_t seg = blockIdx.x;
    const int l = level_of_seg(lv, seg);
    const int64_t w0 = (seg - lv.seg_off[l]) * kScanSeg + 4 * (int64_t)threadIdx.x;   // 4 consecutive words per thread
    const int64_t base = lv.word_off[l];
    uint32_t pc[4], mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        pc[j] = w0 + j < lv.words[l] ? __popc(sp.bitmap[base + w0 + j]) : 0u;
        mine += pc[j];
    }
    uint32_t seg_total;
    uint32_t run = sp.segsum[seg] + block_exclusive_scan(mine, wsum, &seg_total);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (w0 + j < lv.words[l]) sp.rank[base + w0 + j] = run;
        run += pc[j];
    }
}

// One thread per bitmap byte = 8 sibling cells (+ one thread for the root):
// writes the child / data / parent_depth rows of the byte's occupied cells
// (svox.py:535-546).  Siblings are consecutive nodes and share their parent.
__global__ void __launch_bounds__(kBuildBlock)
build_emit_kernel(BuildLevels lv, BuildSpace sp, int32_t* __restrict__ child, int32_t* __restrict__ data,
                  int32_t* __restrict__ parent_depth, int64_t n_rows, int32_t empty_index) {
    const int64_t g = (int64_t)blockIdx.x * kBuildBlock + threadIdx.x;     // byte index over all levels
    const int64_t total_bytes = 4 * lv.total_words;
    if (g > total_bytes) return;
    int l = 0;
    uint32_t bits = 1u;                  // the root: one occupied "cell" of level 0
    uint32_t c0 = 0;                     // first cell of this byte
    int32_t node = 0;
    if (g < total_bytes) {
        l = lv.depth - 1;
        while (l > 1 && g < 4 * lv.word_off[l]) --l;
        const int64_t b = g - 4 * lv.word_off[l];           // byte within the level
        if (b * 8 >= ((int64_t)1 << (3 * l))) return;       // padding of a level smaller than one word
        const int64_t wi = lv.word_off[l] + (b >> 2);
        const uint32_t word = sp.bitmap[wi];
        const int sh = (int)(b & 3) * 8;
        bits = (word >> sh) & 0xffu;
        if (bits == 0u) return;
        c0 = (uint32_t)(b * 8);
        node = sp.level_off[l] + (int32_t)(sp.rank[wi] + __popc(word & ((1u << sh) - 1u)));
    }
    const bool kids_are_nodes = l + 1 < lv.depth;
    const int32_t kids_first = kids_are_nodes ? sp.level_off[l + 1] : 0;
    // the 8 cells of a byte are the children of cell (byte index) of the level above
    const int32_t pnode = l > 1 ? sp.level_off[l - 1] + (int32_t)cell_rank(lv, sp, l - 1, c0 >> 3) : 0;
    typedef int32_t v4i __attribute__((ext_vector_type(4)));
    const v4i empty4 = {empty_index, empty_index, empty_index, empty_index};
    // the children of cells c0 .. c0+7 are the 8 bytes = 2 words starting at byte c0 of the next level
    uint32_t kw[2] = {0u, 0u};
    int32_t kid_next = 0;
    if (kids_are_nodes) {
        const int64_t k0 = lv.word_off[l + 1] + (c0 >> 2);
        kw[0] = sp.bitmap[k0];
        kw[1] = l >= 1 ? sp.bitmap[k0 + 1] : 0u;            // the root's children are one byte
        kid_next = kids_first + (int32_t)sp.rank[k0];
        if (l == 0) kw[0] &= 0xffu;
    }
    for (int j = 0; j < 8; ++j) {
        const uint32_t kb = (kw[j >> 2] >> ((j & 3) * 8)) & 0xffu;     // children of cell c0 + j
        if ((bits >> j) & 1u) {
            if ((int64_t)node < n_rows) {
                int32_t ch[8];
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    ch[s] = ((kb >> s) & 1u) ? kid_next + __popc(kb & ((1u << s) - 1u)) - node : 0;
                v4i* crow = reinterpret_cast<v4i*>(child + (int64_t)node * 8);
                crow[0] = v4i{ch[0], ch[1], ch[2], ch[3]};
                crow[1] = v4i{ch[4], ch[5], ch[6], ch[7]};
                v4i* drow = reinterpret_cast<v4i*>(data + (int64_t)node * 8);
                drow[0] = empty4;
                drow[1] = empty4;
                // _pack_index of the split leaf (svox.py:542); the root row stays (0, 0) as in a fresh N3Tree
                parent_depth[2 * (int64_t)node + 0] = l > 0 ? pnode * 8 + j : 0;
                parent_depth[2 * (int64_t)node + 1] = l;
            }
            ++node;
        }
        kid_next += __popc(kb);          // a cell that is not occupied has no occupied children
    }
}

// construct_tree_kernel (svox_kernel.cu:110-121) on the tree just emitted
__global__ void __launch_bounds__(kBuildBlock)
build_assign_kernel(const float* __restrict__ points, int64_t P, const float* __restrict__ offset,
                    const float* __restrict__ scaling, BuildLevels lv, BuildSpace sp,
                    int32_t* __restrict__ data, int64_t n_rows) {
    const int64_t i = (int64_t)blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= P) return;
    const uint32_t c = point_code(offset, scaling, points + 3 * i, lv.depth);
    const int l = lv.depth - 1;
    const int32_t node = l > 0 ? sp.level_off[l] + (int32_t)cell_rank(lv, sp, l, c >> 3) : 0;
    if ((int64_t)node < n_rows) atomicMin(data + (int64_t)node * 8 + (c & 7u), (int32_t)i);
}

// construct_tree on an arbitrary tree: two passes make "smallest index wins"
// independent of what the leaf held before.
template <bool N2>
__global__ void __launch_bounds__(kBuildBlock)
construct_kernel(TreeDev tr, const float* __restrict__ points, int64_t P, int pass) {
    const int64_t i = (int64_t)blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= P) return;
    const float* p = points + 3 * i;
    const float px = tr.offset[0] + tr.scaling[0] * p[0];
    const float py = tr.offset[1] + tr.scaling[1] * p[1];
    const float pz = tr.offset[2] + tr.scaling[2] * p[2];
    Leaf lf;
    locate<N2>(tr, px, py, pz, lf);
    int32_t* d = const_cast<int32_t*>(tr.data) + lf.slot;
    if (pass == 0) *d = 0x7fffffff;
    else atomicMin(d, (int32_t)i);
}

// N3Tree.refine for an explicit leaf list (svox.py:520-546) on a tree of any N:
// leaf i of the list becomes internal node `filled + i`; its N^3 slots inherit the
// leaf's data word; child / parent_depth bookkeeping as refine() does with tensor ops.
__global__ void __launch_bounds__(kBuildBlock)
refine_kernel(const int64_t* __restrict__ leaf_node, int64_t U, int N, int64_t filled,
              int32_t* __restrict__ child, int32_t* __restrict__ data, int32_t* __restrict__ parent_depth,
              const int32_t* __restrict__ node_id) {
    const int64_t i = (int64_t)blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= U) return;
    const int64_t* ln = leaf_node + 4 * i;
    const int64_t n3 = (int64_t)N * N * N;
    const int64_t slot = ((ln[0] * N + ln[1]) * N + ln[2]) * N + ln[3];
    const int64_t node = filled + i;
    child[slot] = (int32_t)(node - ln[0]);                       // :535-536
    const int32_t word = data[slot];
    for (int64_t k = 0; k < n3; ++k) {                            // :537-538
        data[node * n3 + k] = word;
        child[node * n3 + k] = 0;
    }
    parent_depth[2 * node + 0] = node_id != nullptr ? node_id[i] : (int32_t)slot;   // :539 (_pack_index)
    parent_depth[2 * node + 1] = parent_depth[2 * ln[0] + 1] + 1;                   // :540-541
}

}  // namespace svoxt

using namespace svoxt;

namespace {

int check_build_args(const char* fn, const float* points, int64_t P, const float* offset, const float* scaling,
                     int32_t depth, const void* workspace, int64_t workspace_bytes) {
    if (depth < 1 || depth > kMaxLevels) return set_error(SVOXT_ERR_INVALID, "%s: depth must be in [1, 10]", fn);
    if (P < 0 || P >= 2147483647LL) return set_error(SVOXT_ERR_INVALID, "%s: point count must be in [0, 2^31)", fn);
    if ((P > 0 && points == nullptr) || offset == nullptr || scaling == nullptr)
        return set_error(SVOXT_ERR_INVALID, "%s: points / offset / scaling is NULL", fn);
    if (workspace == nullptr || workspace_bytes < space_bytes(make_levels(depth)))
        return set_error(SVOXT_ERR_INVALID, "%s: workspace is NULL or smaller than svoxt_build_workspace_bytes(depth)", fn);
    return SVOXT_OK;
}

unsigned blocks_for(int64_t n) { return (unsigned)((n + kBuildBlock - 1) / kBuildBlock); }

}  // namespace

extern "C" {

int64_t svoxt_build_workspace_bytes(int32_t depth) {
    if (depth < 1 || depth > kMaxLevels) return -1;
    return space_bytes(make_levels(depth));
}

int svoxt_build_count(const float* points, int64_t P, const float* offset, const float* scaling,
                      int32_t depth, void* workspace, int64_t workspace_bytes,
                      int64_t* n_internal, void* stream) {
    const char* fn = "svoxt_build_count";
    int rc;
    if ((rc = check_build_args(fn, points, P, offset, scaling, depth, workspace, workspace_bytes))) return rc;
    if (n_internal == nullptr) return set_error(SVOXT_ERR_INVALID, "%s: n_internal is NULL", fn);
    hipStream_t st = (hipStream_t)stream;
    const BuildLevels lv = make_levels(depth);
    const BuildSpace sp = carve(lv, workspace);
    const int top = depth - 1;                           // finest level that holds nodes
    if (top >= 1) {
        uint32_t* top_bits = sp.bitmap + lv.word_off[top];
        const bool bytes = lv.bytemap_bytes > 0;
        const hipError_t e = bytes ? hipMemsetAsync(sp.bytemap, 0, lv.bytemap_bytes, st)
                                   : hipMemsetAsync(top_bits, 0, 4 * lv.words[top], st);
        if (e != hipSuccess) return set_error(SVOXT_ERR_HIP, "%s: %s", fn, hipGetErrorString(e));
        if (bytes) {
            if (P > 0)
                hipLaunchKernelGGL(build_mark_bytes_kernel, dim3(blocks_for(P)), dim3(kBuildBlock), 0, st,
                                   points, P, offset, scaling, top, sp.bytemap);
            hipLaunchKernelGGL(build_pack_kernel, dim3(blocks_for(lv.words[top])), dim3(kBuildBlock), 0, st,
                               sp.bytemap, top_bits, lv.words[top]);
        } else if (P > 0) {
            hipLaunchKernelGGL(build_mark_bits_kernel, dim3(blocks_for(P)), dim3(kBuildBlock), 0, st,
                               points, P, offset, scaling, top, top_bits);
        }
        int l = top - 1;
        for (; l > kSmallLevel; --l)
            hipLaunchKernelGGL(build_reduce_kernel, dim3(blocks_for(lv.words[l])), dim3(kBuildBlock), 0, st,
                               sp.bitmap + lv.word_off[l + 1], sp.bitmap + lv.word_off[l], lv.words[l],
                               (int64_t)1 << (3 * l));
        if (l >= 1)
            hipLaunchKernelGGL(build_reduce_small_kernel, dim3(1), dim3(kBuildBlock), 0, st, lv, sp, l);
        hipLaunchKernelGGL(build_segsum_kernel, dim3((unsigned)lv.total_segs), dim3(kBuildBlock), 0, st, lv, sp);
    }
    hipLaunchKernelGGL(build_levelscan_kernel, dim3(1), dim3(kBuildBlock), 0, st, lv, sp, n_internal);
    if (top >= 1)
        hipLaunchKernelGGL(build_rank_kernel, dim3((unsigned)lv.total_segs), dim3(kBuildBlock), 0, st, lv, sp);
    return check_launch(fn);
}

int svoxt_build_emit(const float* points, int64_t P, const float* offset, const float* scaling,
                     int32_t depth, const void* workspace, int64_t workspace_bytes,
                     int32_t* child, int32_t* data, int32_t* parent_depth,
                     int64_t n_internal, int32_t empty_index, void* stream) {
    const char* fn = "svoxt_build_emit";
    int rc;
    if ((rc = check_build_args(fn, points, P, offset, scaling, depth, workspace, workspace_bytes))) return rc;
    if (child == nullptr || data == nullptr || parent_depth == nullptr)
        return set_error(SVOXT_ERR_INVALID, "%s: child / data / parent_depth is NULL", fn);
    if (n_internal < 1 || n_internal * 8 >= 2147483648LL)
        return set_error(SVOXT_ERR_INVALID, "%s: n_internal must be in [1, 2^28)", fn);
    if ((int64_t)empty_index < P) return set_error(SVOXT_ERR_INVALID, "%s: empty_index must be >= the point count", fn);
    hipStream_t st = (hipStream_t)stream;
    const BuildLevels lv = make_levels(depth);
    const BuildSpace sp = carve(lv, const_cast<void*>(workspace));
    hipLaunchKernelGGL(build_emit_kernel, dim3(blocks_for(4 * lv.total_words + 1)), dim3(kBuildBlock), 0, st,
                       lv, sp, child, data, parent_depth, n_internal, empty_index);
    if (P > 0)
        hipLaunchKernelGGL(build_assign_kernel, dim3(blocks_for(P)), dim3(kBuildBlock), 0, st,
                           points, P, offset, scaling, lv, sp, data, n_internal);
    return check_launch(fn);
}

int svoxt_construct_tree(const svoxt_tree* t, const float* points, int64_t P, void* stream) {
    const char* fn = "svoxt_construct_tree";
    if (t == nullptr || t->data == nullptr || t->child == nullptr || t->offset == nullptr || t->scaling == nullptr)
        return set_error(SVOXT_ERR_INVALID, "%s: tree / data / child / offset / scaling is NULL", fn);
    if (t->N < 2 || t->n_internal < 1 || (double)t->n_internal * t->N * t->N * t->N >= 2147483648.0)
        return set_error(SVOXT_ERR_INVALID, "%s: bad tree extents", fn);
    if (P < 0 || P >= 2147483647LL) return set_error(SVOXT_ERR_INVALID, "%s: point count must be in [0, 2^31)", fn);
    if (P == 0) return SVOXT_OK;
    if (points == nullptr) return set_error(SVOXT_ERR_INVALID, "%s: points is NULL", fn);
    TreeDev tr{};
    tr.N = t->N; tr.data = t->data; tr.child = t->child; tr.offset = t->offset; tr.scaling = t->scaling;
    hipStream_t st = (hipStream_t)stream;
    for (int pass = 0; pass < 2; ++pass) {
        if (t->N == 2)
            hipLaunchKernelGGL((construct_kernel<true>), dim3(blocks_for(P)), dim3(kBuildBlock), 0, st, tr, points, P, pass);
        else
            hipLaunchKernelGGL((construct_kernel<false>), dim3(blocks_for(P)), dim3(kBuildBlock), 0, st, tr, points, P, pass);
    }
    return check_launch(fn);
}

int svoxt_refine(const int64_t* leaf_node, int64_t n_leaves, int32_t N, int64_t filled, int64_t capacity,
                 int32_t* child, int32_t* data, int32_t* parent_depth, const int32_t* node_id, void* stream) {
    const char* fn = "svoxt_refine";
    if (n_leaves < 0 || N < 2 || filled < 1 || filled + n_leaves > capacity)
        return set_error(SVOXT_ERR_INVALID, "%s: bad extents (filled + n_leaves must fit the capacity)", fn);
    if ((double)capacity * N * N * N >= 2147483648.0)
        return set_error(SVOXT_ERR_INVALID, "%s: tree too large for 32-bit slot indices", fn);
    if (n_leaves == 0) return SVOXT_OK;
    if (leaf_node == nullptr || child == nullptr || data == nullptr || parent_depth == nullptr)
        return set_error(SVOXT_ERR_INVALID, "%s: a pointer is NULL", fn);
    hipLaunchKernelGGL(refine_kernel, dim3(blocks_for(n_leaves)), dim3(kBuildBlock), 0, (hipStream_t)stream,
                       leaf_node, n_leaves, (int)N, filled, child, data, parent_depth, node_id);
    return check_launch(fn);
}

}  // extern "C"
