// svoxt_device.h -- device-side building blocks shared by the render, depth,
// opacity, query and counting kernels (gfx950 / CDNA4 only).
//
// Numerical contract.  The *stepping* arithmetic of a ray (world->tree
// transform, direction normalisation, slab tests, leaf lookup, step length)
// decides which leaves a ray visits, so it is kept bit-identical to the
// source semantics of the reference (and therefore to oracle/):
//   - the translation unit is compiled with -ffp-contract=off and carries
//     `#pragma clang fp contract(off)`: no multiply-add is ever fused here;
//   - sqrt and divide are the correctly rounded forms;
//   - the reference's mixed float/double expressions are evaluated in double
//     where the reference evaluates them in double.
// For N == 2 the root->leaf descent is done on integer cell coordinates.  This
// is exact: for p in [0,1) every `p*2; floor; subtract` of the reference
// (svox_t/csrc/include/common.cuh:78-86) is an exact floating-point operation,
// so the slot path is the bit string of floor(p * 2^k) and the leaf-local
// coordinate is frac(p * 2^k), both computed here without rounding.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace svoxt {

struct TreeDev {
    const float* __restrict__ features;
    int64_t M;
    int K;
    int N;
    const int32_t* __restrict__ data;
    const int32_t* __restrict__ child;
    const float* __restrict__ offset;
    const float* __restrict__ scaling;
    const float* __restrict__ extra;
    int extra_rows;
    int extra_cols;
    float* weight_accum;
    const float* __restrict__ xform;   // transformation_matrices [M, d, d] or null (generic kernels only)
    int xform_dim;                     // d: 3 or 4
    // optional acceleration grid (N == 2 only): 2^G cells per axis, one uint2 per
    // cell, see locate_accel().  Derived data: a cache of what a root descent
    // of `child` / `data` would find, never a different answer.
    const uint32_t* __restrict__ accel;
    int accel_g;
    bool accel_bricks;       // the cells lie in 4 x 4 x 4 bricks (accel_cell_index)
    // optional (RGBA-style rows of 8 / 16 / 32 floats): exp_table[row][c] = pexpf(-features[row][c]) for the feature
    // columns, sigma unchanged in the last one (svoxt_exp_table_build).  Derived data, like the grid: the same bits
    // the kernels would form themselves, formed once per row instead of once per sample and sweep.
    const float* __restrict__ etab;
};

struct RaysDev {
    const float* __restrict__ origins;
    const float* __restrict__ dirs;
    const float* __restrict__ vdirs;
    int64_t Q;
    int tiles_per_row;   // > 0: rays are a row-major W x H image (W = 8 * tiles_per_row), walk it in 8x8 tiles
    uint32_t tile0;      // launches that cover a range of the batch's 64-ray tiles start at this one (else 0)
    // camera mode (c2w != null): ray q is pixel (q % width, q / width) of a pinhole
    // camera, generated in the kernel; origins / dirs / vdirs are not read
    const float* __restrict__ c2w;   // camera-to-world, rows of 4 floats, 3 rows used
    float fx, fy;
    int width, height;
    const int32_t* __restrict__ order;   // != null: launch thread i works on ray order[i] (svoxt_rays.order)
    int super_tiles;     // != 0 (images only): the 8 x 8 pixel tiles are walked in super-tiles (ray_of_thread)
    int tile_rows;       // super_tiles: rows of tiles (H / 8) ...
    float inv_st;        // ... and 1 / (8 tiles_per_row): the walk's one division by a number that is not a power of two
    float inv_t;         // images of fewer than 2^22 tiles: 1 / tiles_per_row (else 0: ray_of_thread divides)
};

// Which ray a thread works on.  By default thread i takes ray i (a wavefront =
// 64 consecutive rays).  When the caller states that the batch is a row-major
// image, a wavefront takes an 8x8 pixel tile instead (or, given a permutation, the rays it
// puts next to each other: svoxt_rays.order): neighbouring rays then
// traverse the same leaves in both image directions (fewer divergent
// iterations, better cache reuse, more gradient rows merged before the
// atomics).  Only the assignment of rays to lanes changes; every ray's result
// is the same and is written at the ray's own index.
#ifndef SVOXT_SUPER_TILE
#define SVOXT_SUPER_TILE 8
#endif
__device__ __forceinline__ int64_t ray_of_thread(const RaysDev& rays, int64_t tid) {
    if (rays.order != nullptr) return tid < rays.Q ? (int64_t)rays.order[tid] : rays.Q;
    if (rays.tiles_per_row <= 0) return tid;
    const int64_t tile = tid >> 6;
    const int within = (int)(tid & 63);
    int64_t ty, tx;
    if (rays.inv_t != 0.f) {                                     // (a 64-bit division per thread of every image kernel otherwise)
        const int t32 = (int)tile, T = rays.tiles_per_row;
        int y = (int)((float)t32 * rays.inv_t), x = t32 - y * T;
        if (x < 0) { --y; x += T; } else if (x >= T) { ++y; x -= T; }
        ty = y; tx = x;
    } else {
        ty = tile / rays.tiles_per_row; tx = tile - ty * rays.tiles_per_row;
    }
    if (SVOXT_SUPER_TILE > 0 && rays.super_tiles != 0) {
        // (r04) The tiles are walked in SUPER-TILES of S x S tiles (64 x 64 pixels), row-major inside and between them,
        // the last column / row of super-tiles as narrow / low as the image leaves them: workgroups that are resident
        // together then render a few compact patches of the image instead of a band eight pixels high across all of
        // it, and what they gather and add to lies close together.  1024 x 1024 / depth 9 / K = 32 (the tree's 578 MiB
        // of features do not fit the caches): forward+backward 2.64 -> 2.44 ms, the per-tile backward 1.55 -> 1.35;
        // super-tiles of 4 / 16 tiles: 2.46 / 2.48.  800 x 800 / depth 8, whose 71 MiB fit the Infinity Cache, LOSES 1.7 %
        // (its busy tiles -- one in three -- then come in clusters): the host sets super_tiles only for feature tables
        // that do not fit (to_dev(rays, tree): a function of the tree's M and K alone, so that the forward that records
        // lists and the backward that walks them agree).  A tile is the same 8 x 8 pixels either way.
        // (every thread of every image kernel comes through here -- the 32 channel lanes of a ray in shade_chan_kernel
        // each -- so: the division by 8 T as a multiplication by the host's reciprocal with a correction, shifts and masks
        // inside a full super-tile, real divisions only in the last, ragged row and column of super-tiles.  With four
        // plain divisions the shade kernel of config 4 lost 0.03 ms to this mapping alone.)
        constexpr int S = SVOXT_SUPER_TILE;
        static_assert(S == 8 || S == 0, "the shifts below");
        const int T = rays.tiles_per_row, TR = rays.tile_rows, ST = S * T;
        const int t = (int)tile;
        if (t < T * TR) {
            int srow = (int)((float)t * rays.inv_st);            // t < 2^22 (the host's condition): off by one at most
            int u = t - srow * ST;
            if (u < 0) { --srow; u += ST; } else if (u >= ST) { ++srow; u -= ST; }
            const int h = min(S, TR - srow * S);                 // tile rows of this row of super-tiles
            const int full = T >> 3, wl = T & 7;                 // full-width super-tiles per row, width of the last one
            if (h == S && u < full * (S * S)) {                  // inside a full super-tile: the common case
                const int v = u & (S * S - 1);
                ty = srow * S + (v >> 3);
                tx = (u >> 6) * S + (v & 7);
            } else if (u < full * h * S) {
                const int sc = u / (h * S), v = u - sc * h * S;
                ty = srow * S + (v >> 3);
                tx = sc * S + (v & 7);
            } else {
                const int v = u - full * h * S;
                ty = srow * S + v / wl;
                tx = full * S + v % wl;
            }
        }
    }
    return ((ty << 3) + (within >> 3)) * ((int64_t)rays.tiles_per_row << 3) + (tx << 3) + (within & 7);
}

// svox_t/csrc/include/data_spec.hpp:129-145
struct Opts {
    float step_size;
    float background_brightness;
    int format;
    int basis_dim;
    int ndc_width;
    int ndc_height;
    float ndc_focal;
    int min_comp;
    int max_comp;
    float sigma_thresh;
    float stop_thresh;
};

enum { FMT_RGBA = 0, FMT_SH = 1, FMT_SG = 2, FMT_ASG = 3 };

// (float)(1.0 - 1e-6): the reference clamps with `min(scalar_t(1.0) - 1e-6, q)`
// evaluated in double and rounds the result to float
// (svox_t/csrc/include/common.cuh:40); for a float q that equals
// fminf(q, (float)(1.0 - 1e-6)).
constexpr float kClampHi = (float)(1.0 - 1e-6);

struct Ray {
    float ox, oy, oz;      // origin in tree space
    float dx, dy, dz;      // unit direction in tree space
    float ix, iy, iz;      // 1 / (d + 1e-9)
    float delta_scale;
    float tmin, tmax;
};

// One slab test of a ray against the unit cube (rt_kernel.cu:202-218).
__device__ __forceinline__ void dda_unit(float cx, float cy, float cz,
                                         float ix, float iy, float iz,
                                         float& tmin, float& tmax) {
    float t1, t2;
    tmin = 0.0f;
    tmax = 1e9f;
    t1 = -cx * ix; t2 = t1 + ix;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    t1 = -cy * iy; t2 = t1 + iy;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    t1 = -cz * iz; t2 = t1 + iz;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
}

// cam2world_ray (rt_kernel.cu:1153-1166): unit world-space direction of pixel q,
// with the reference's mixed arithmetic (pixel offset and focal division in
// double, `x*x + y*y` in float, `+ 1.0` in double, sqrtf of the float).
__device__ __forceinline__ void camera_dir(const RaysDev& rays, int64_t q, float d[3]) {
    const int iy = (int)(q / rays.width), ix = (int)(q - (int64_t)iy * rays.width);
    float x = (float)(((double)ix - 0.5 * (double)rays.width) / (double)rays.fx);
    float y = (float)(-((double)iy - 0.5 * (double)rays.height) / (double)rays.fy);
    float z = sqrtf((float)((double)(x * x + y * y) + 1.0));
    x /= z; y /= z; z = -1.0f / z;
    const float* c = rays.c2w;
    d[0] = c[0] * x + c[1] * y + c[2] * z;
    d[1] = c[4] * x + c[5] * y + c[6] * z;
    d[2] = c[8] * x + c[9] * y + c[10] * z;
}

// The direction the view-dependent basis is evaluated for: rays.vdirs[q], or
// in camera mode the pixel's direction before the NDC warp (rt_kernel.cu:1203).
__device__ __forceinline__ void load_vdir(const RaysDev& rays, int64_t q, float vd[3]) {
    if (rays.c2w == nullptr) {
        const float* v = rays.vdirs + 3 * q;
        vd[0] = v[0]; vd[1] = v[1]; vd[2] = v[2];
    } else {
        camera_dir(rays, q, vd);
    }
}

// The direction part of the preamble: a world-space direction scaled into tree space and normalised
// (common.cuh:45-51), and _get_delta_scale (rt_kernel.cu:188-199) = 1 / the scaled direction's length.
__device__ __forceinline__ float dir_to_tree(const TreeDev& tr, const float d[3], float& dx, float& dy, float& dz) {
    dx = d[0] * tr.scaling[0]; dy = d[1] * tr.scaling[1]; dz = d[2] * tr.scaling[2];
    // sqrtf and `/` are the correctly rounded forms under hipcc's default
    // -fhip-fp32-correctly-rounded-divide-sqrt (__fsqrt_rn is NOT: it maps to
    // the native approximation unless OCML_BASIC_ROUNDED_OPERATIONS is set).
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    const float ds = 1.f / nrm;
    dx *= ds; dy *= ds; dz *= ds;
    return ds;
}

// Ray preamble: render_ray_kernel's transform (rt_kernel.cu:663-665,
// common.cuh:45-51), _get_delta_scale (:188-199), invdir (:237, double) and
// the cube slab test (:239-241).  Returns false when the ray misses the cube.
// Camera mode: the ray comes from cam2world_ray + maybe_world2ndc (:1170-1190)
// as in render_image_kernel (:1193-1211).
__device__ __forceinline__ bool setup_ray(const TreeDev& tr, const RaysDev& rays, const Opts& opt,
                                          int64_t q, Ray& r) {
    const float s0 = tr.scaling[0], s1 = tr.scaling[1], s2 = tr.scaling[2];
    const float f0 = tr.offset[0], f1 = tr.offset[1], f2 = tr.offset[2];
    float o[3], d[3];
    if (rays.c2w == nullptr) {
        const float* po = rays.origins + 3 * q;
        const float* pd = rays.dirs + 3 * q;
        o[0] = po[0]; o[1] = po[1]; o[2] = po[2];
        d[0] = pd[0]; d[1] = pd[1]; d[2] = pd[2];
    } else {
        camera_dir(rays, q, d);
        o[0] = rays.c2w[3]; o[1] = rays.c2w[7]; o[2] = rays.c2w[11];
        if (opt.ndc_width >= 0) {            // `if (opt.ndc_width < 0) return;` (:1174)
            const float near = 1.f;
            const float t = -(near + o[2]) / d[2];
            o[0] = o[0] + t * d[0]; o[1] = o[1] + t * d[1]; o[2] = o[2] + t * d[2];
            const float kx = (2 * opt.ndc_focal) / opt.ndc_width, ky = (2 * opt.ndc_focal) / opt.ndc_height;
            const float n0 = -kx * (d[0] / d[2] - o[0] / o[2]);
            const float n1 = -ky * (d[1] / d[2] - o[1] / o[2]);
            const float n2 = -2 * near / o[2];
            o[0] = -kx * (o[0] / o[2]);
            o[1] = -ky * (o[1] / o[2]);
            o[2] = 1 + 2 * near / o[2];
            const float norm = sqrtf(n0 * n0 + n1 * n1 + n2 * n2);
            d[0] = n0 / norm; d[1] = n1 / norm; d[2] = n2 / norm;
        }
    }
    r.ox = f0 + s0 * o[0];
    r.oy = f1 + s1 * o[1];
    r.oz = f2 + s2 * o[2];
    float dx, dy, dz;
    const float ds = dir_to_tree(tr, d, dx, dy, dz);
    r.dx = dx; r.dy = dy; r.dz = dz;
    r.delta_scale = ds;
    r.ix = (float)(1.0 / ((double)dx + 1e-9));
    r.iy = (float)(1.0 / ((double)dy + 1e-9));
    r.iz = (float)(1.0 / ((double)dz + 1e-9));
    dda_unit(r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, r.tmin, r.tmax);
    return !(r.tmax < 0 || r.tmin > r.tmax);
}

struct Leaf {
    uint32_t slot;      // flat index into child / data: ((node*N+u)*N+v)*N+w
    float lx, ly, lz;   // leaf-local coordinates in [0,1)
    float cube_sz;      // N^levels
    int levels;         // child words read
};

// Generic-N continuation of the descent (common.cuh:74-99), starting at
// `node` with node-local coordinates p in [0,1).
__device__ __forceinline__ void descend_generic(const TreeDev& tr, int32_t node,
                                                float px, float py, float pz,
                                                float cube_sz, int levels, Leaf& lf) {
    const int Ni = tr.N;
    const float N = (float)Ni;
    while (true) {
        px *= N; py *= N; pz *= N;
        const float fu = floorf(px), fv = floorf(py), fw = floorf(pz);
        px -= fu; py -= fv; pz -= fw;
        int u = (int)fu, v = (int)fv, w = (int)fw;
        u = min(max(u, 0), Ni - 1);
        v = min(max(v, 0), Ni - 1);
        w = min(max(w, 0), Ni - 1);
        const uint32_t slot = (((uint32_t)node * Ni + u) * Ni + v) * Ni + w;
        const int32_t skip = tr.child[slot];
        ++levels;
        if (skip == 0) {
            lf.slot = slot; lf.lx = px; lf.ly = py; lf.lz = pz;
            lf.cube_sz = cube_sz; lf.levels = levels;
            return;
        }
        cube_sz *= N;
        node += skip;
    }
}

constexpr int kFixBits = 22;   // integer descent handles up to 22 levels

// query_single_from_root (common.cuh:63-100) for a tree-space point.
// MARK (instrumentation, svoxt_count_touched): mark[slot] = 1 for every child word read
template <bool N2, bool MARK = false>
__device__ __forceinline__ void locate(const TreeDev& tr, float px, float py, float pz, Leaf& lf,
                                       uint8_t* mark = nullptr) {
    px = fmaxf(0.f, fminf(kClampHi, px));
    py = fmaxf(0.f, fminf(kClampHi, py));
    pz = fmaxf(0.f, fminf(kClampHi, pz));
    if constexpr (!N2) {
        descend_generic(tr, 0, px, py, pz, (float)tr.N, 0, lf);
    } else {
        const float S = (float)(1 << kFixBits);
        const uint32_t ux = (uint32_t)(px * S);   // exact: px*2^22 then truncation
        const uint32_t uy = (uint32_t)(py * S);
        const uint32_t uz = (uint32_t)(pz * S);
        int32_t node = 0;
        int k = 1;
        uint32_t slot;
        int32_t skip;
#pragma unroll 1
        for (;; ++k) {
            const int sh = kFixBits - k;
            const uint32_t cell = (((ux >> sh) & 1u) << 2) | (((uy >> sh) & 1u) << 1) | ((uz >> sh) & 1u);
            slot = ((uint32_t)node << 3) + cell;
            skip = tr.child[slot];
            if constexpr (MARK) mark[slot] = 1;
            if (skip == 0 || k == kFixBits) break;
            node += skip;
        }
        const float sc = __int_as_float((127 + k) << 23);   // 2^k
        const float fx = px * sc, fy = py * sc, fz = pz * sc;
        const float lx = fx - floorf(fx), ly = fy - floorf(fy), lz = fz - floorf(fz);
        if (skip == 0) {
            lf.slot = slot; lf.lx = lx; lf.ly = ly; lf.lz = lz;
            lf.cube_sz = sc; lf.levels = k;
        } else {
            // deeper than the fixed-point path resolves: finish in float
            descend_generic(tr, node + skip, lx, ly, lz, sc * 2.f, k, lf);
        }
    }
}

// Acceleration grid.  Cell (cx,cy,cz) of the 2^G grid caches the state of the
// root descent after (at most) G levels for any point inside the cell, in 4 bytes:
//   bit 31 set  : the descent ended in a leaf at depth (bits 27-30) <= G whose data word
//                 (feature row index) is bits 0-26 -- all ones: an empty leaf (a data word that
//                 is no row of the feature table; the kernels see an index >= M)
//   bit 31 clear: the descent is at internal node (bits 0-26) after G levels
// (trees with 2^27 - 1 rows or internal nodes and more get no grid).  The 2^(3G) cells are
// followed by one (child word, data word) pair per tree slot.
// Built by accel_build_kernel / accel_nodes_kernel from child/data; the march then needs
// one 4-byte load instead of up to G dependent 4-byte loads per step, plus one 8-byte load
// per level below the grid (none for the data word).
constexpr uint32_t kAccelLeaf = 0x80000000u;
constexpr uint32_t kAccelIdx = 0x07ffffffu;      // row / node bits; as a row: empty

// Where cell (x, y, z) of the 2^G grid lies in memory.  Row-major (z fastest), or -- `bricks`, r05 -- in bricks of
// 4 x 4 x 4 cells (256 bytes: two 128-byte lines), the bricks in row-major order: a ray's consecutive crossings, and
// the 64 rays of a tile at one crossing, then fall into a few lines whichever way the rays point, where row-major a
// step along x or y is a new line (1 KB / 256 KB away at G = 8).  Measured at 800 x 800 / depth 8 (G = 8): the marching
// wavefronts of fwd_roles_kernel, which wait for exactly this load, 0.2477 -> 0.2342 ms (requests to L2 -26 %, fetches
// -15 %); the one-kernel forward, which is short of issue slots rather than of memory, pays for the six extra integer
// operations per crossing, 0.2033 -> 0.2157 ms; rows of 32 floats on the depth-9 tree: no change at g 7, 0.837 -> 0.801 ms a
// level finer.  Nested bricks (4^3 in 16^3 ...) and 8^3 bricks: between the two; 2 x 4 x 4, 4 x 4 x 2, 4 x 2 x 4, 4 x 4 x 8 and
// 2 x 2 x 2: within noise of 4 x 4 x 4 or behind it.  So the layout is the caller's to choose per grid (svoxt_tree.accel_log2,
// SVOXT_ACCEL_BRICKS), uniform per launch.
__device__ __forceinline__ uint32_t accel_cell_index(uint32_t x, uint32_t y, uint32_t z, int G, bool bricks) {
    if (!bricks) return (((x << G) + y) << G) + z;
    const int Gb = G - 2;
    return ((((((x >> 2) << Gb) + (y >> 2)) << Gb) + (z >> 2)) << 6) | ((x & 3u) << 4) | ((y & 3u) << 2) | (z & 3u);
}
__device__ __forceinline__ void accel_cell_coords(uint32_t c, int G, bool bricks, uint32_t& x, uint32_t& y, uint32_t& z) {
    if (!bricks) {
        const uint32_t mask = (1u << G) - 1u;
        z = c & mask; y = (c >> G) & mask; x = c >> (2 * G);
    } else {
        const int Gb = G - 2;
        const uint32_t brick = c >> 6, mb = (1u << Gb) - 1u;
        z = ((brick & mb) << 2) | (c & 3u);
        y = (((brick >> Gb) & mb) << 2) | ((c >> 2) & 3u);
        x = ((brick >> (2 * Gb)) << 2) | ((c >> 4) & 3u);
    }
}

// The grid cell of a point: the clamp (in place: what follows uses the clamped point), the fixed-point coordinates the
// descent below the grid takes its child bits from, and the cell's index.
__device__ __forceinline__ uint32_t accel_point(const TreeDev& tr, float& px, float& py, float& pz,
                                                uint32_t& ux, uint32_t& uy, uint32_t& uz) {
    // (r03) the clamp as ONE median-of-three per axis: for every non-NaN p the value of fmaxf(0, fminf(hi, p))
    // (a zero may come out with the other sign: p * 2^22 truncates to the same 0 and f - floor(f) is +0 either
    // way); a NaN position cannot reach this point (no finite t makes one, and t = NaN ends the march).
    px = __builtin_amdgcn_fmed3f(px, 0.f, kClampHi);
    py = __builtin_amdgcn_fmed3f(py, 0.f, kClampHi);
    pz = __builtin_amdgcn_fmed3f(pz, 0.f, kClampHi);
    const float S = (float)(1 << kFixBits);
    ux = (uint32_t)(px * S);
    uy = (uint32_t)(py * S);
    uz = (uint32_t)(pz * S);
    const int G = tr.accel_g;
    const int gs = kFixBits - G;
    return accel_cell_index(ux >> gs, uy >> gs, uz >> gs, G, tr.accel_bricks);
}

// The leaf of a (clamped) point given its grid cell's word: the levels below the grid, the data word, the local coordinates.
template <bool MARK = false>
__device__ __forceinline__ void accel_resolve(const TreeDev& tr, uint32_t cell, float px, float py, float pz,
                                              uint32_t ux, uint32_t uy, uint32_t uz, Leaf& lf, int32_t& idx, uint8_t* mark = nullptr) {
    const int G = tr.accel_g;
    int k;
    uint32_t slot = 0xffffffffu;
    if (cell & kAccelLeaf) {
        k = (int)((cell >> 27) & 15u);
        idx = (int32_t)(cell & kAccelIdx);          // (an empty leaf's all-ones is no row of the table: M < 2^27 - 1 with a grid)
    } else {
        // below the grid: (child word, data word) pairs, so that reaching a leaf costs no
        // further dependent load for its data word
        const uint2* __restrict__ nodes = reinterpret_cast<const uint2*>(tr.accel + ((size_t)1 << (3 * G)));
        int32_t node = (int32_t)cell;
        int32_t skip;
        uint2 cd;
        k = G + 1;
#pragma unroll 1
        for (;; ++k) {
            const int sh = kFixBits - k;
            const uint32_t c3 = (((ux >> sh) & 1u) << 2) | (((uy >> sh) & 1u) << 1) | ((uz >> sh) & 1u);
            slot = ((uint32_t)node << 3) + c3;
            // ONE 8-byte load.  Written as `cd = nodes[slot]` the compiler splits it into two dword
            // loads and sinks the data word's out of the loop: a second dependent round trip per
            // leaf crossing, which is what the pairs exist to avoid (seen in the ISA: a
            // `global_load_dword ... offset:4` after the loop).  A relaxed wavefront-scope atomic
            // load is an ordinary global_load_dwordx2 that may not be split.
            const unsigned long long w64 = __hip_atomic_load(
                reinterpret_cast<const unsigned long long*>(nodes + slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            cd = make_uint2((uint32_t)w64, (uint32_t)(w64 >> 32));
            if constexpr (MARK) mark[((size_t)1 << (3 * G)) + slot] = 1;
            skip = (int32_t)cd.x;
            if (skip == 0 || k == kFixBits) break;
            node += skip;
        }
        if (skip != 0) {   // deeper than the fixed-point path resolves: generic float descent
            const float sc = __int_as_float((127 + k) << 23);
            const float fx = px * sc, fy = py * sc, fz = pz * sc;
            descend_generic(tr, node + skip, fx - floorf(fx), fy - floorf(fy), fz - floorf(fz),
                            sc * 2.f, k, lf);
            idx = tr.data[lf.slot];
            return;
        }
        idx = (int32_t)cd.y;
    }
    const float sc = __int_as_float((127 + k) << 23);   // 2^k
    const float fx = px * sc, fy = py * sc, fz = pz * sc;
    lf.slot = slot;
    // f - floorf(f) for 0 <= f < 2^22 is exact and below 1: v_fract_f32 (= min(f - floor(f), 1 - 2^-24)) is that value
    lf.lx = __builtin_amdgcn_fractf(fx); lf.ly = __builtin_amdgcn_fractf(fy); lf.lz = __builtin_amdgcn_fractf(fz);
    lf.cube_sz = sc;
    lf.levels = k;
}

// MARK: mark[cell] = 1 for the grid cell read, mark[n_cells + slot] = 1 for every (child, data) pair read
template <bool MARK = false>
__device__ __forceinline__ void locate_accel(const TreeDev& tr, float px, float py, float pz,
                                             Leaf& lf, int32_t& idx, uint8_t* mark = nullptr) {
    uint32_t ux, uy, uz;
    const uint32_t ci = accel_point(tr, px, py, pz, ux, uy, uz);
    const uint32_t cell = tr.accel[ci];
    if constexpr (MARK) mark[ci] = 1;
    accel_resolve<MARK>(tr, cell, px, py, pz, ux, uy, uz, lf, idx, mark);
}

struct Sample {
    Leaf leaf;
    int32_t idx;       // data word of the leaf: the feature row when `valid`
    float delta_t;     // chord of the leaf along the ray + step_size
    bool valid;        // 0 <= idx < M (anything else marks an empty leaf)
};

// The step across a leaf from a point inside it (leaf-local coordinates l in [0, 1), cube_sz = 2^level for N = 2):
// _dda_unit on the leaf-local point (rt_kernel.cu:273).  The point lies inside its leaf, so on every axis one of
// t1 = -c*inv and t2 = t1 + inv is <= 0: the entry distance max(0, min...) is 0 and `subcube_tmax - subcube_tmin`
// (:275) is subcube_tmax itself -- only the exit distance is evaluated.
template <bool N2>
__device__ __forceinline__ float leaf_delta_t(float lx, float ly, float lz, float cube_sz, const Ray& r, float step_size) {
    float sub_tmax = 1e9f;
    {
        float t1, t2;
        t1 = -lx * r.ix; t2 = t1 + r.ix; sub_tmax = fminf(sub_tmax, fmaxf(t1, t2));
        t1 = -ly * r.iy; t2 = t1 + r.iy; sub_tmax = fminf(sub_tmax, fmaxf(t1, t2));
        t1 = -lz * r.iz; t2 = t1 + r.iz; sub_tmax = fminf(sub_tmax, fmaxf(t1, t2));
    }
    float t_subcube;
    if constexpr (N2) {
        // cube_sz is a power of two: multiplying by its reciprocal is the
        // same correctly rounded result as the reference's division (:275).
        t_subcube = sub_tmax * __int_as_float((254 << 23) - __float_as_int(cube_sz));
    } else {
        t_subcube = sub_tmax / cube_sz;
    }
    return t_subcube + step_size;
}

// One leaf crossing: rt_kernel.cu:261-277.
// ACC: -1 = look at tr.accel at run time; 1 / 0 = the caller knows the acceleration grid is
// there / is not (a kernel instance without the other descent: fewer registers, and no false
// register hazards between the two paths' loads for the wait-count placement to trip over).
// MARK: see locate / locate_accel; without the grid the leaf's data word is mark[n_slots + slot].
template <bool N2, int ACC = -1, bool MARK = false>
__device__ __forceinline__ void march_step(const TreeDev& tr, const Ray& r, float step_size,
                                           float t, Sample& s, uint8_t* mark = nullptr, uint32_t n_slots = 0) {
    const float px = r.ox + t * r.dx;
    const float py = r.oy + t * r.dy;
    const float pz = r.oz + t * r.dz;
    // Locals, assigned to `s` once after the branches join: with the two branches storing
    // straight into different members of `s`, the compiler merges the stores into one through
    // a pointer phi, which pins slot and idx in scratch memory (a scratch round trip per step).
    Leaf lf;
    int32_t idx;
    if (N2 && (ACC == 1 || (ACC < 0 && tr.accel != nullptr))) {
        locate_accel<MARK>(tr, px, py, pz, lf, idx, mark);   // leaf.slot / leaf.levels are not reference-accurate here
    } else {
        locate<N2, MARK>(tr, px, py, pz, lf, mark);
        idx = tr.data[lf.slot];
        if constexpr (MARK) mark[n_slots + lf.slot] = 1;
    }
    s.leaf = lf;
    s.idx = idx;
    // `*data_idx_ptr >= features.size(0)` compares int32 with int64 (:269); a
    // negative index is therefore "valid" for the reference (and reads out of
    // bounds).  Treat it as empty instead of faulting.
    // (one unsigned compare: a negative index is a large unsigned one, and a table of 2^31 rows or more holds
    // every non-negative int32)
    s.valid = (uint32_t)s.idx < (tr.M > 0x7fffffffLL ? 0x80000000u : (uint32_t)tr.M);
    s.delta_t = leaf_delta_t<N2>(s.leaf.lx, s.leaf.ly, s.leaf.lz, s.leaf.cube_sz, r, step_size);
}

// expf with a fixed operation sequence (Cephes-style: n = rint(x*log2e), two-
// step Cody-Waite reduction, degree-5 polynomial, exact power-of-two scaling).
// Every step is a correctly rounded IEEE operation (mul, add, fma, rint), so
// the result is reproducible bit for bit on any IEEE machine -- oracle/ carries
// the same sequence, which makes the whole forward pass comparable exactly
// rather than to within "some ulps of some libm".  Accuracy ~1 ulp (the
// reference's CUDA expf is specified to 2 ulp).  Results below 2^-126
// (x < -87) are returned as 0, above FLT_MAX as +inf.
// PIN: keep the polynomial evaluation unconditional.  Left alone, the compiler sinks it under
// the selects at the end as a branch; the basic-block boundaries then stop the scheduler from
// overlapping the independent exponentials and double-precision sigmoids of one sample (a
// wavefront alone on its SIMD -- the tail of the list walk -- retires one dependent
// instruction every ~11 clocks).  Costs registers, so only where a kernel's tail pays for it.
template <bool PIN = false>
__device__ __forceinline__ float pexpf(float x) {
    // Branch-free: evaluate on a clamped argument, then select the special
    // cases (identical results to the early-return form in oracle/).
    const float xc = fminf(fmaxf(x, -87.0f), 88.72283905206835f);
    const float n = rintf(xc * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, xc);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float y = __builtin_fmaf(p, r * r, r);
    y = y + 1.0f;
    const int ni = (int)n;
    const int n1 = ni >> 1, n2 = ni - n1;
    y = y * __int_as_float((n1 + 127) << 23);
    y = y * __int_as_float((n2 + 127) << 23);
    if constexpr (PIN) asm volatile("" : "+v"(y));
    y = (x > 88.72283905206835f) ? __int_as_float(0x7f800000) : y;
    y = (x < -87.0f) ? 0.0f : y;
    return (x != x) ? x : y;
}

// Tolerance-grade arithmetic for the opt-in SVOXT_LISTS_NATIVE_MATH mode (shading only; never the stepping):
// e^x as v_exp_f32(x * log2 e).  v_exp_f32 is good to 1 ulp; the rounding of the product adds up to
// 2^-24 |x log2 e| to the exponent, i.e. a relative error of |x| * 6e-8 in the result (4e-7 at |x| = 6,
// 5e-6 at the float range's end) -- where CUDA's expf, which the reference calls, is specified to
// 2 ulp (rt_kernel.cu:280, 300, 304, 397, 408, 420, 461, 472, 476).  Results below 2^-126 flush to 0.
__device__ __forceinline__ float nexpf(float x) {
    return __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
}
// 1 / (1 + e^-x) with the hardware reciprocal (1 ulp) where the reference divides in double
__device__ __forceinline__ float nsigmoidf(float x) {
    return __builtin_amdgcn_rcpf(1.f + nexpf(-x));
}

// t += delta_t (rt_kernel.cu:321), guarded: if the step is too small to move t
// in float (step_size <= 0 on a degenerate crossing, or ~1e-8 of t) the
// reference loops forever; here the march ends instead of hanging the GPU.
// Never triggers for a step that advances, so results are unchanged.
__device__ __forceinline__ float march_advance(float t, float delta_t) {
    const float tn = t + delta_t;
    return (tn > t) ? tn : __int_as_float(0x7f800000);
}

// SH constants, `const float` in the reference (rt_kernel.cu:54-84).
__device__ __constant__ const float kC0 = 0.28209479177387814;
__device__ __constant__ const float kC1 = 0.4886025119029199;
__device__ __constant__ const float kC2[5] = {1.0925484305920792, -1.0925484305920792,
                                              0.31539156525252005, -1.0925484305920792,
                                              0.5462742152960396};
__device__ __constant__ const float kC3[7] = {-0.5900435899266435, 2.890611442640554,
                                              -0.4570457994644658, 0.3731763325901154,
                                              -0.4570457994644658, 1.445305721320277,
                                              -0.5900435899266435};
__device__ __constant__ const float kC4[9] = {2.5033429417967046,  -1.7701307697799304,
                                              0.9461746957575601,  -0.6690465435572892,
                                              0.10578554691520431, -0.6690465435572892,
                                              0.47308734787878004, -1.7701307697799304,
                                              0.6258357354491761};

__device__ __forceinline__ float dot3(const float* u, const float* v) {
    return u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
}

// maybe_precalc_basis (rt_kernel.cu:110-185).  BD is the number of basis
// functions; `out` must hold at least BD (<= 25) floats.
template <int BD_STATIC>
__device__ __forceinline__ void precalc_basis(int format, int basis_dim_rt, const TreeDev& tr,
                                              float x, float y, float z, float* out) {
    const int bd = BD_STATIC > 0 ? BD_STATIC : basis_dim_rt;
    if (format == FMT_SH) {
        out[0] = kC0;
        const float xx = x * x, yy = y * y, zz = z * z;
        const float xy = x * y, yz = y * z, xz = x * z;
        if (bd == 25) {
            out[16] = kC4[0] * xy * (xx - yy);
            out[17] = kC4[1] * yz * (3 * xx - yy);
            out[18] = kC4[2] * xy * (7 * zz - 1.f);
            out[19] = kC4[3] * yz * (7 * zz - 3.f);
            out[20] = kC4[4] * (zz * (35 * zz - 30) + 3);
            out[21] = kC4[5] * xz * (7 * zz - 3);
            out[22] = kC4[6] * (xx - yy) * (7 * zz - 1.f);
            out[23] = kC4[7] * xz * (xx - 3 * yy);
            out[24] = kC4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy));
        }
        if (bd == 25 || bd == 16) {
            out[9] = kC3[0] * y * (3 * xx - yy);
            out[10] = kC3[1] * xy * z;
            out[11] = kC3[2] * y * (4 * zz - xx - yy);
            out[12] = kC3[3] * z * (2 * zz - 3 * xx - 3 * yy);
            out[13] = kC3[4] * x * (4 * zz - xx - yy);
            out[14] = kC3[5] * z * (xx - yy);
            out[15] = kC3[6] * x * (xx - 3 * yy);
        }
        if (bd == 25 || bd == 16 || bd == 9) {
            out[4] = kC2[0] * xy;
            out[5] = kC2[1] * yz;
            out[6] = (float)((double)kC2[2] * (2.0 * (double)zz - (double)xx - (double)yy));  // :169
            out[7] = kC2[3] * xz;
            out[8] = kC2[4] * (xx - yy);
        }
        if (bd == 25 || bd == 16 || bd == 9 || bd == 4) {
            out[1] = -kC1 * y;
            out[2] = kC1 * z;
            out[3] = -kC1 * x;
        }
    } else if (format == FMT_SG) {
        const float dir[3] = {x, y, z};
        for (int i = 0; i < bd; ++i) {
            const float* lobe = tr.extra + (int64_t)i * tr.extra_cols;
            out[i] = pexpf(lobe[0] * (dot3(dir, lobe + 1) - 1.f)) / (float)bd;
        }
    } else if (format == FMT_ASG) {
        const float dir[3] = {x, y, z};
        for (int i = 0; i < bd; ++i) {
            const float* lobe = tr.extra + (int64_t)i * tr.extra_cols;
            const float S = dot3(dir, lobe + 8);
            const float dot_x = dot3(dir, lobe + 2);
            const float dot_y = dot3(dir, lobe + 5);
            out[i] = S * pexpf(-lobe[0] * dot_x * dot_x - lobe[1] * dot_y * dot_y) / (float)bd;
        }
    }
}

// The SG / ASG half of maybe_precalc_basis (rt_kernel.cu:110-185) for a constant number of lobes, for the kernels
// that keep a ray's basis values in registers (r03: `render_fwd_kernel` / `render_bwd_kernel<..., LOBES>` -- what they
// do with the values per sample is what they do with an SH basis).  Same operations in the same order as
// precalc_basis (dot3 = u0 v0 + u1 v1 + u2 v2), written with scalars and unrolled so that `out` is registers.
// one lobe's value (i < bd): the loop bodies of maybe_precalc_basis for SG / ASG
__device__ __forceinline__ float lobe_value(int format, const TreeDev& tr, float x, float y, float z, int i, int bd) {
    const float* lobe = tr.extra + (int64_t)i * tr.extra_cols;
    if (format == FMT_SG) {
        const float d = x * lobe[1] + y * lobe[2] + z * lobe[3];
        return pexpf(lobe[0] * (d - 1.f)) / (float)bd;
    }
    const float S = x * lobe[8] + y * lobe[9] + z * lobe[10];
    const float dot_x = x * lobe[2] + y * lobe[3] + z * lobe[4];
    const float dot_y = x * lobe[5] + y * lobe[6] + z * lobe[7];
    return S * pexpf(-lobe[0] * dot_x * dot_x - lobe[1] * dot_y * dot_y) / (float)bd;
}
template <int BD>
__device__ __forceinline__ void precalc_lobes(int format, const TreeDev& tr, float x, float y, float z, float* out) {
#pragma unroll
    for (int i = 0; i < BD; ++i) out[i] = lobe_value(format, tr, x, y, z, i, BD);
}

// Per-leaf view-direction rotation (rt_kernel.cu:283-291, :387-395): the basis
// is re-evaluated for ray_dir = M[idx] * vdir.
__device__ __forceinline__ void rotated_basis(const TreeDev& tr, int format, int basis_dim, int32_t idx,
                                              const float* vdir, float* basis) {
    const int d = tr.xform_dim;
    const float* m = tr.xform + (int64_t)idx * (d * d);
    const float x = m[0] * vdir[0] + m[1] * vdir[1] + m[2] * vdir[2];
    const float y = m[d] * vdir[0] + m[d + 1] * vdir[1] + m[d + 2] * vdir[2];
    const float z = m[2 * d] * vdir[0] + m[2 * d + 1] * vdir[1] + m[2 * d + 2] * vdir[2];
    precalc_basis<0>(format, basis_dim, tr, x, y, z, basis);
}

// the rotated view direction alone (rt_kernel.cu:284-288)
__device__ __forceinline__ void rotated_dir(const TreeDev& tr, int32_t idx, const float* vdir, float (&out)[3]) {
    const int d = tr.xform_dim;
    const float* m = tr.xform + (int64_t)idx * (d * d);
    out[0] = m[0] * vdir[0] + m[1] * vdir[1] + m[2] * vdir[2];
    out[1] = m[d] * vdir[0] + m[d + 1] * vdir[1] + m[d + 2] * vdir[2];
    out[2] = m[2 * d] * vdir[0] + m[2 * d + 1] * vdir[1] + m[2 * d + 2] * vdir[2];
}

template <int BD>
__device__ __forceinline__ void rotated_sh_basis(const TreeDev& tr, int32_t idx, const float* vdir, float* basis) {
    const int d = tr.xform_dim;
    const float* m = tr.xform + (int64_t)idx * (d * d);
    const float x = m[0] * vdir[0] + m[1] * vdir[1] + m[2] * vdir[2];
    const float y = m[d] * vdir[0] + m[d + 1] * vdir[1] + m[d + 2] * vdir[2];
    const float z = m[2 * d] * vdir[0] + m[2 * d + 1] * vdir[1] + m[2 * d + 2] * vdir[2];
    precalc_basis<BD>(FMT_SH, BD, tr, x, y, z, basis);
}

// -(sum_i basis[i] * row[i]) for exp(-sum) (rt_kernel.cu:293-300), the products rounded one by one and added in
// index order like the reference's `tmp += basis_fn[i] * tree_val[off + i]` from tmp = 0.  Two things about the
// instruction stream (r03 ISA of the shade rounds): each product goes through an opaque register, which keeps the
// compiler from pairing them into v_pk_mul_f32 -- the rows' channels start at odd floats, so every pair cost two
// v_mov to line its operands up (4 packed multiplies + 8 moves where 8 multiplies do); and the sum starts from
// the first product instead of 0 + product -- the same float unless the product is -0, and then the sums differ
// in the sign of a zero at most, which exp() of the negated sum does not see (e^-0 = e^+0 = 1).
template <int BD>
__device__ __forceinline__ float neg_sh_dot(const float* __restrict__ basis, const float* __restrict__ row) {
    float tmp = basis[0] * row[0];
    asm("" : "+v"(tmp));
#pragma unroll
    for (int i = 1; i < BD; ++i) {
        float pr = basis[i] * row[i];
        asm("" : "+v"(pr));
        tmp += pr;
    }
    return -tmp;
}

// The double-precision quotients of the reference's `w / (1.0 + expf(-x))` family (rt_kernel.cu:300,304,408,420,472,
// 476) for THEIR operand range, with the compiler's own operation sequence minus what that range makes an identity.
// hipcc expands n / d in double to  v_div_scale x2, v_rcp_f64, two Newton steps (4 fma), q = n r, rem = fma(-d, q, n),
// v_div_fmas(rem, r, q), v_div_fixup.  Here n is a float in {0} u [2^-149, 1] (a weight T (1 - att), or 1.0) and d =
// 1.0 + double(e) with e a float >= 0 or +inf, so d is in [1, 2^128] or +inf: no operand is scaled (v_div_scale
// pre-scales only a denormal or tiny operand, a quotient or reciprocal that would be denormal, or exponents 768 apart:
// none can occur -- the quotient is >= 2^-277), v_div_fmas without a scale flag is a plain fma, and what remains is the
// same rcp / fma / mul sequence and the same v_div_fixup (which delivers the n = 0, d = inf cases as the full sequence
// does).  Bit-identical to the `/` operator on that range: exp/div_check.hip sweeps all 2^31 patterns of e against it
// on the GPU (tests/test_gpu_div_exact.py).  9 double-precision instructions instead of 11 (8 for the reciprocal).
__device__ __forceinline__ double div_unit_range(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    const double q = n * r;
    return __builtin_amdgcn_div_fixup(__builtin_fma(__builtin_fma(-d, q, n), r, q), d, n);
}
__device__ __forceinline__ double rcp_unit_range(double d) {        // 1.0 / d, d as above
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return __builtin_amdgcn_div_fixup(__builtin_fma(__builtin_fma(-d, r, 1.0), r, r), d, 1.0);
}

// The reference's `w / (1.0 + expf(-x))` family is evaluated in double
// (rt_kernel.cu:300,304,408,420,472,476).
template <bool PIN = false>
__device__ __forceinline__ double sigmoid_d(float x) {
    return 1.0 / (1.0 + (double)pexpf<PIN>(-x));
}

// Feature row -> registers.  16-byte loads when the row stride allows it
// (K % 4 == 0 and torch allocations are >= 16-byte aligned).
template <int K>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&row)[K]) {
    if constexpr (K % 4 == 0) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
        for (int i = 0; i < K / 4; ++i) {
            const float4 v = p4[i];
            row[4 * i] = v.x; row[4 * i + 1] = v.y; row[4 * i + 2] = v.z; row[4 * i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i) row[i] = p[i];
    }
}

// Cooperative flush of the rows staged in this iteration.  The lanes with a
// sample have written their K values to consecutive staging rows (slot = rank
// among the active lanes, destination row in sidx[slot]); the wave now walks
// the `n` rows two at a time -- lanes 0..31 take row 2r, lanes 32..63 row 2r+1
// (one row per round when K > 32) -- so every atomic instruction covers
// contiguous K-float segments of the gradient table.  Rounds are independent
// (no per-round cross-lane bookkeeping), so the LDS reads and the atomics of
// several rounds overlap.
//
// Rows that hit the same leaf are NOT merged first: measured on the headline
// workload a wavefront's k-th samples land on 28-29 distinct leaves out of
// 36-50 active lanes, so merging saves < 1.7x atomics and its bookkeeping cost
// more than it saved.
template <int K, int KS>
__device__ __forceinline__ void flush_staged(const float* __restrict__ stage, const int32_t* __restrict__ sidx,
                                             int n, int lane, float* __restrict__ grad, int gstride) {
    constexpr int ROWS = (K <= 32) ? 2 : 1;           // staged rows per atomic instruction
    constexpr int LPR = 64 / ROWS;                    // lanes per row
    constexpr int CHUNKS = (K + LPR - 1) / LPR;       // instructions per row (K > 64 only)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int half = (ROWS == 2) ? (lane >> 5) : 0;
    const int j = (ROWS == 2) ? (lane & 31) : lane;
#pragma unroll 4
    for (int base = 0; base < n; base += ROWS) {
        const int rw = base + half;
        if (rw < n) {
            const int32_t ridx = sidx[rw];
#pragma unroll
            for (int ch = 0; ch < CHUNKS; ++ch) {
                const int col = j + ch * LPR;
                if (col < K) atomicAdd(grad + (int64_t)ridx * gstride + col, stage[rw * KS + col]);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

}  // namespace svoxt
