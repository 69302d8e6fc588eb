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
// Kernels (templates, defined in headers this file includes; launched from the C ABI below):
//   svoxt_lists.h         launch shape, sample lists (records, blocks, pool, LDS staging), hand-over layouts
//   svoxt_fwd_kernels.h   render_fwd_kernel (trace_ray; optionally records each ray's composited samples),
//                         render_fwd_generic_kernel, sigma_mask_kernel, march_rec_kernel + shade_tile_kernel /
//                         shade_chan_kernel + tail_chan_kernel (the forward as two kernels)
//   svoxt_misc_kernels.h  opacity_fwd_kernel (+ opacity_walk_kernel, opacity_merge_kernel: backward from lists),
//                         depth_kernel, count_fwd_kernel, count_touched_kernel, query_fwd_kernel, query_bwd_kernel,
//                         leaves_count / scan / scatter kernels, compact_rows_kernel, accel_build_kernel,
//                         accel_nodes_kernel
// Other translation units of the library: svoxt_bwd.hip (the backward: svoxt_bwd_kernels.h -- render_bwd_kernel,
// grad_merge_kernel, grad_fused_kernel, grad_wide_kernel, the generic fallbacks -- and its launch logic; its own
// unit so that the two halves compile side by side), svoxt_build.hip (octree from a point
// cloud, construct_tree), svoxt_motion.hip (motion variants, point skinning), svoxt_order.hip
// (coherent order for ray batches that are not images).
// The design rationale and the measurements behind each choice are in DESIGN.md 4 and, step by step, NOTEBOOK.md 5.
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

#include "svoxt_launch.h"
#include "svoxt_fwd_kernels.h"
#include "svoxt_misc_kernels.h"


// ===========================================================================
// C ABI
// ===========================================================================

using namespace svoxt;

namespace {
thread_local char g_err[512] = "";
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
    if (t->accel != nullptr) {
        const int32_t g = t->accel_log2 & ~SVOXT_ACCEL_BRICKS;
        if (g < 1 || g > 8 || ((t->accel_log2 & SVOXT_ACCEL_BRICKS) && g < 2))
            return fail(SVOXT_ERR_INVALID, "%s: accel_log2 must be in [1, 8] (with SVOXT_ACCEL_BRICKS: [2, 8])", fn);
    }
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
        if (r->order != nullptr) return fail(SVOXT_ERR_INVALID, "%s: rays.order does not go with camera mode", fn);
    } else if (r->Q > 0 && (r->origins == nullptr || r->dirs == nullptr || r->vdirs == nullptr)) {
        return fail(SVOXT_ERR_INVALID, "%s: rays.origins / dirs / vdirs is NULL", fn);
    }
    if (r->Q >= (int64_t)kBlock * 2147483647LL) return fail(SVOXT_ERR_INVALID, "%s: too many rays", fn);
    if (r->image_width < 0 || r->image_height < 0) return fail(SVOXT_ERR_INVALID, "%s: negative image extent", fn);
    if (r->order != nullptr && r->image_width > 0)
        return fail(SVOXT_ERR_INVALID, "%s: rays.order and the image hint are two answers to the same question: give one", fn);
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
    d.accel = use_accel ? reinterpret_cast<const uint32_t*>(t->accel) : nullptr;
    d.accel_g = use_accel ? (t->accel_log2 & ~SVOXT_ACCEL_BRICKS) : 0;
    d.accel_bricks = use_accel && (t->accel_log2 & SVOXT_ACCEL_BRICKS) != 0;
    // the exponentials table serves RGBA-style rows of 8 / 16 / 32 floats only
    d.etab = (t->exp_table != nullptr && (t->K == 8 || t->K == 16 || t->K == 32)) ? t->exp_table : nullptr;
    return d;
}

// Images of trees with more feature rows than this are walked in super-tiles (RaysDev.super_tiles).  The rule is the
// TREE'S size, not its features' bytes (measured r04, forward+backward, row-major -> super-tiles): depth 9 (4.7 M rows) at
// 1024 x 1024: K = 32 2.63 -> 2.40 ms, SH9 1.99 -> 1.71, K = 4 (72 MiB of features!) 1.15 -> 0.95; at 800 x 800 unchanged
// (+-0.5 %); depth 8 (0.67 M rows): SH9 800 x 800 +1.7 %, K = 32 800 x 800 +6 %, SH9 1600 x 1600 +1 % -- never a gain.
constexpr int64_t kSuperTileRows = (int64_t)1 << 21;
static int64_t g_super_tile_rows = kSuperTileRows;        // svoxt_set_super_tile_rows

RaysDev to_dev(const svoxt_rays* r, const svoxt_tree* t, const svoxt_sample_lists* l) {
    RaysDev d;
    d.origins = r->origins; d.dirs = r->dirs; d.vdirs = r->vdirs; d.Q = r->Q;
    // image hint: usable only if the batch is exactly a W x H image of 8x8 tiles
    const bool tiled = r->image_width > 0 && r->image_height > 0 && r->image_width % 8 == 0 &&
                       r->image_height % 8 == 0 && (int64_t)r->image_width * r->image_height == r->Q;
    d.tiles_per_row = tiled ? r->image_width / 8 : 0;
    d.tile0 = 0;
    d.c2w = r->c2w; d.fx = r->fx; d.fy = r->fy;
    d.order = r->order;
    d.width = r->image_width; d.height = r->image_height;
    // the walk of an image's tiles: what the lists were recorded with where they say so (SVOXT_LISTS_WALK_*), else the rule
    const int32_t forced = l != nullptr ? (l->flags & (SVOXT_LISTS_WALK_ROWMAJOR | SVOXT_LISTS_WALK_SUPER)) : 0;
    const bool super = forced ? (forced & SVOXT_LISTS_WALK_SUPER) != 0 : (t != nullptr && t->M > g_super_tile_rows);
    d.super_tiles = (tiled && super && r->Q < ((int64_t)1 << 28)) ? 1 : 0;   // (tiles < 2^22: inv_st)
    d.tile_rows = tiled ? r->image_height / 8 : 0;
    d.inv_st = tiled ? 1.0f / (8.0f * (float)d.tiles_per_row) : 0.f;
    d.inv_t = (tiled && r->Q < ((int64_t)1 << 28)) ? 1.0f / (float)d.tiles_per_row : 0.f;
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

#ifndef SVOXT_ROLES_BAND_ROWS
#define SVOXT_ROLES_BAND_ROWS 0          // tile rows per band of fwd_roles_kernel's marching order (0: the plain order)
#endif
inline unsigned finish_grid(unsigned nb) { return nb < 512u ? nb : 512u; }     // fwd_finish_kernel: runs of tiles, two workgroups per CU
// fwd_roles_kernel's launch shape: the marching workgroups' tile map (RolesMap), how many of them, and the grid
struct RolesLaunch { RolesMap map; int n_march; unsigned grid; };
inline RolesLaunch roles_launch(unsigned nb, int tiles_per_row) {
    RolesLaunch r;
    const int G = (int)((nb + 7) / 8);                              // groups of 8 tiles
    int gb = 0;
    if (SVOXT_ROLES_BAND_ROWS > 0) {
        gb = tiles_per_row > 0 ? (SVOXT_ROLES_BAND_ROWS * tiles_per_row + 7) / 8 : 32;
        if (gb < 1 || G <= 8 * gb) gb = 0;                           // (fewer than a band per XCD: nothing to arrange)
    }
    r.map.gb = gb;
    r.map.full = gb > 0 ? G / (8 * gb) : 0;                         // rounds of eight full bands
    const int rest = gb > 0 ? G - r.map.full * 8 * gb : 0;          // groups left for the last round
    r.map.gb_last = gb > 0 ? (rest + 7) / 8 : 0;
    r.n_march = gb > 0 ? 8 * (r.map.full * gb + r.map.gb_last) : (G + 7) / 8 * 8;
    int mx = 0;
    for (int x = 0; x < 8; ++x) r.map.cnt[x] = 0;
    for (int b = 0; b < r.n_march; ++b) {                            // (a few thousand iterations per launch)
        const int64_t t0 = roles_group_of(r.map, b) * 8;
        if (t0 < (int64_t)nb) r.map.cnt[b & 7] += (int)((int64_t)nb - t0 < 8 ? (int64_t)nb - t0 : 8);
    }
    for (int x = 0; x < 8; ++x) if (r.map.cnt[x] > mx) mx = r.map.cnt[x];
    r.grid = (unsigned)r.n_march + 8u * (unsigned)mx;
    return r;
}

// ... with sample lists: the recording forward as march + tile shade + tail launch, leaving the backward's hand-over
// (position-major: terms_state 3), and the per-tile backward over it -- the exact one-kernel form only.
template <bool N2>
bool launch_lobes_fwd_record(const TreeDev& tr, const RaysDev& rays, const Opts& opt, float* out, RecLists L, uint4* aux,
                             hipStream_t st, const uint32_t* sigma_mask, int32_t* tile_state, int tflags = 0) {
    if (L.terms == nullptr || !lobes_payload(opt, tr.K)) return false;
    const unsigned nb = nblocks(rays.Q);
    const bool acc = N2 && tr.accel != nullptr;
    if constexpr (N2) {
        // march and shade in one launch (fwd_roles_kernel<..., LOBES>) where SH takes it too: up to 16 lobes, grid at hand
        if (tile_state != nullptr && sigma_mask != nullptr && acc && opt.basis_dim <= 16 && nb < (1u << 30)) {
            const RolesLaunch rl = roles_launch(nb, rays.tiles_per_row);
            const int n_march = rl.n_march;
            const unsigned grid = rl.grid;
            const RolesMap rmap = rl.map;
#define SVOXT_LOBES_ROLES(BB)                                                                                       \
    {                                                                                                               \
        hipLaunchKernelGGL((fwd_roles_kernel<FMT_SH, BB, 1, true, true>), dim3(grid), dim3(512), 0, st,             \
                           tr, rays, opt, L, aux, out, sigma_mask, tile_state, n_march, (int)nb, tflags, rmap);     \
        hipLaunchKernelGGL((fwd_finish_kernel<FMT_SH, BB, true, true>), dim3(finish_grid(nb)), dim3(512), 0, st,    \
                           tr, rays, opt, L, aux, out, tile_state, (int)nb);                                        \
        return true;                                                                                                \
    }
            switch (opt.basis_dim) {
                case 1: SVOXT_LOBES_ROLES(1)
                case 4: SVOXT_LOBES_ROLES(4)
                case 9: SVOXT_LOBES_ROLES(9)
                case 16: SVOXT_LOBES_ROLES(16)
            }
#undef SVOXT_LOBES_ROLES
        }
    }
    if (sigma_mask != nullptr) {
        if (acc) hipLaunchKernelGGL((march_rec_kernel<N2, false, 1, true>), dim3(nb), dim3(kBlock), 0, st, tr, rays, opt, L, aux, sigma_mask);
        else hipLaunchKernelGGL((march_rec_kernel<N2, false, 0, true>), dim3(nb), dim3(kBlock), 0, st, tr, rays, opt, L, aux, sigma_mask);
    } else {
        if (acc) hipLaunchKernelGGL((march_rec_kernel<N2, false, 1>), dim3(nb), dim3(kBlock), 0, st, tr, rays, opt, L, aux, (const uint32_t*)nullptr);
        else hipLaunchKernelGGL((march_rec_kernel<N2, false, 0>), dim3(nb), dim3(kBlock), 0, st, tr, rays, opt, L, aux, (const uint32_t*)nullptr);
    }
#define SVOXT_LOBES_FWD(BB)                                                                                         \
    {                                                                                                               \
        hipLaunchKernelGGL((shade_tile_kernel<FMT_SH, BB, false, false, true, true>), dim3(nb), dim3(512), 0, st,   \
                           tr, rays, opt, L, aux, out, (const int32_t*)nullptr);                                    \
        hipLaunchKernelGGL((render_fwd_kernel<FMT_SH, 3, BB, N2, false, false, true, true>), dim3(nb), dim3(kBlock), \
                           0, st, tr, rays, opt, out, L, aux);                                                      \
        return true;                                                                                                \
    }
    switch (opt.basis_dim) {
        case 1: SVOXT_LOBES_FWD(1)
        case 4: SVOXT_LOBES_FWD(4)
        case 9: SVOXT_LOBES_FWD(9)
        case 16: SVOXT_LOBES_FWD(16)
        case 25: SVOXT_LOBES_FWD(25)
    }
#undef SVOXT_LOBES_FWD
    return false;
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
    } else if constexpr (!REC) {
        // SG / ASG payloads with 1 / 4 / 9 / 16 / 25 lobes and three channels (r03; they used to take the generic
        // kernels): a ray's basis values are formed once (precalc_lobes) and used like an SH basis
#define SVOXT_FWD_LOBES(BB)                                                                                     \
    hipLaunchKernelGGL((render_fwd_kernel<FMT_SH, 3, BB, N2, false, false, false, true>), dim3(nb), dim3(kBlock), \
                       0, st, tr, rays, opt, out, L, aux);                                                      \
    return true;
        if ((opt.format == FMT_SG || opt.format == FMT_ASG) && C == 3 && tr.K == 3 * opt.basis_dim + 1) {
            switch (opt.basis_dim) {
                case 1: SVOXT_FWD_LOBES(1)
                case 4: SVOXT_FWD_LOBES(4)
                case 9: SVOXT_FWD_LOBES(9)
                case 16: SVOXT_FWD_LOBES(16)
                case 25: SVOXT_FWD_LOBES(25)
            }
        }
#undef SVOXT_FWD_LOBES
    }
#undef SVOXT_FWD
    return false;
}

// two-kernel forward (march_rec_kernel + a shade kernel + tail launch of render_fwd_kernel), all
// components.  Default: on for RGBA-style rows of 8 / 16 / 32 floats (shade_chan_kernel: r02,
// 1024 x 1024 depth-9 K = 32), off for the 3-channel payloads (shade_tile_kernel:
// 800x800 depth-8 SH9 0.156 + 0.123 ms against 0.247 ms for the one-kernel forward).
// list_flags: SVOXT_LISTS_FWD_ONE_KERNEL / _TWO_KERNELS force it off / on where a shade kernel exists
// (the caller's choice, handed over with the lists: the library reads no environment).
// with_terms: a recording forward that is to leave the backward's (att, e_c) -- the tile shade kernel
// writes them 1 KB at a time for nothing, the one-kernel forward pays 0.06 ms for its scattered lines
// (r02: 0.145 + 0.13 ms against 0.316 ms) -- so then the two kernels are the default for 3 channels too.
bool fwd_split_enabled(const svoxt_tree* t, const svoxt_options* o, bool with_terms, int32_t list_flags) {
    if (list_flags & SVOXT_LISTS_FWD_ONE_KERNEL) return false;
    if (list_flags & SVOXT_LISTS_FWD_TWO_KERNELS) return true;
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
        if (!STOP && L.terms != nullptr)                                                                      \
            hipLaunchKernelGGL((shade_tile_kernel<F, BB, X, STOP, !STOP>), dim3(nb), dim3(512), 0, st,        \
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
            /* (r05) lists for a backward (!STOP) under a stop threshold: the instance that applies the rule, told to  \
               keep the overflow flags (the march did not stop: the backward's tail still needs them) */             \
            const bool stop_rt = !STOP && opt.stop_thresh > 0.f;                                               \
            if (stop_rt) {                                                                                     \
                if (fast) hipLaunchKernelGGL((shade_chan_kernel<KK, true, true>), dim3(nbc), dim3(256), 0, st, \
                                             tr, rays, opt, L, aux, out, 1);                                   \
                else if (tr.etab != nullptr)                                                                   \
                    hipLaunchKernelGGL((shade_chan_kernel<KK, true, false, true>), dim3(nbc), dim3(256), 0, st, \
                                       tr, rays, opt, L, aux, out, 1);                                         \
                else hipLaunchKernelGGL((shade_chan_kernel<KK, true, false>), dim3(nbc), dim3(256), 0, st,     \
                                        tr, rays, opt, L, aux, out, 1);                                        \
            } else                                                                                             \
            if (fast) hipLaunchKernelGGL((shade_chan_kernel<KK, STOP, true>), dim3(nbc), dim3(256), 0, st,    \
                                         tr, rays, opt, L, aux, out, 0);                                 \
            else if (tr.etab != nullptr)                                                                      \
                hipLaunchKernelGGL((shade_chan_kernel<KK, STOP, false, true>), dim3(nbc), dim3(256), 0, st,   \
                                   tr, rays, opt, L, aux, out, 0);                                       \
            else hipLaunchKernelGGL((shade_chan_kernel<KK, STOP, false>), dim3(nbc), dim3(256), 0, st,        \
                                    tr, rays, opt, L, aux, out, 0);                                      \
            if (fast) hipLaunchKernelGGL((tail_chan_kernel<KK, N2, true>), dim3(nb), dim3(256), 0, st,        \
                                         tr, rays, opt, aux, out, L);                                         \
            else hipLaunchKernelGGL((tail_chan_kernel<KK, N2, false>), dim3(nb), dim3(256), 0, st,            \
                                    tr, rays, opt, aux, out, L);                                              \
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

// (Measured r02 and removed in r03: the tiles cut into 2..7 ranges pipelined over two streams, so that
// the shade of one range runs beside the march of the next -- each cross-stream event dependency cost
// more than the overlap returned: 800x800 depth-8 SH9 0.29 -> 0.59 ms with 4 ranges.)
// March and shade as ONE launch (fwd_roles_kernel) + the fallback shade + the tail launch: 3-channel
// payloads of at most 28 floats on N = 2 trees, no view rotations, no stop rule, the sigma bitmask at hand.
template <bool N2, bool STOP>
bool launch_fwd_roles(const TreeDev& tr, const RaysDev& rays, const Opts& opt, float* out, RecLists L, uint4* aux,
                      hipStream_t st, const uint32_t* sigma_mask, int32_t* tile_state, int tflags, bool xf) {
    if constexpr (!N2 || STOP) {
        return false;
    } else {
        const unsigned nb = nblocks(rays.Q);
        if (nb >= (1u << 30)) return false;
        const RolesLaunch rl = roles_launch(nb, rays.tiles_per_row);
        const int n_march = rl.n_march;                                 // a multiple of 8: see fwd_roles_kernel
        const unsigned grid = rl.grid;                                  // + the shading workgroups (the active ones: one per tile)
        const RolesMap rmap = rl.map;
        // (r05: only with the acceleration grid -- the default for every N = 2 tree of 64 nodes and more; without one the
        // march and the shade stay two launches: the 18 grid-less instances of this kernel had no default caller)
        if (tr.accel == nullptr) return false;
        const bool wt = L.terms != nullptr;
#define SVOXT_ROLES(F, BB)                                                                                        \
        {                                                                                                         \
            if (wt) hipLaunchKernelGGL((fwd_roles_kernel<F, BB, 1, true>), dim3(grid), dim3(512), 0, st,          \
                                       tr, rays, opt, L, aux, out, sigma_mask, tile_state, n_march, (int)nb, tflags, rmap); \
            else hipLaunchKernelGGL((fwd_roles_kernel<F, BB, 1, false>), dim3(grid), dim3(512), 0, st,            \
                                    tr, rays, opt, L, aux, out, sigma_mask, tile_state, n_march, (int)nb, tflags, rmap); \
            /* the fallback shade of what the launch left unshaded + the tails of overflowed rays: one small launch */ \
            if (wt) hipLaunchKernelGGL((fwd_finish_kernel<F, BB, true>), dim3(finish_grid(nb)), dim3(512), 0, st, \
                                       tr, rays, opt, L, aux, out, tile_state, (int)nb);                          \
            else hipLaunchKernelGGL((fwd_finish_kernel<F, BB, false>), dim3(finish_grid(nb)), dim3(512), 0, st,   \
                                    tr, rays, opt, L, aux, out, tile_state, (int)nb);                             \
            return true;                                                                                          \
        }
        if (xf) {
            // (r04) per-leaf view rotations: SH 1 / 4 / 9 without hand-over (the lists' backward is grad_fused_kernel<..., XF>)
            if (opt.format != FMT_SH || opt.basis_dim > 9) return false;
#define SVOXT_ROLES_XF1(BB, AA, WW)                                                                               \
                hipLaunchKernelGGL((fwd_roles_kernel<FMT_SH, BB, AA, WW, false, true>), dim3(grid), dim3(512), 0, st, \
                                   tr, rays, opt, L, aux, out, sigma_mask, tile_state, n_march, (int)nb, tflags, rmap);
#define SVOXT_ROLES_XF(BB)                                                                                        \
            {                                                                                                     \
                if (wt) SVOXT_ROLES_XF1(BB, 1, true) else SVOXT_ROLES_XF1(BB, 1, false)                           \
                if (wt) hipLaunchKernelGGL((fwd_finish_kernel<FMT_SH, BB, true, false, true>), dim3(finish_grid(nb)), dim3(512), 0, st, \
                                           tr, rays, opt, L, aux, out, tile_state, (int)nb);                      \
                else hipLaunchKernelGGL((fwd_finish_kernel<FMT_SH, BB, false, false, true>), dim3(finish_grid(nb)), dim3(512), 0, st, \
                                        tr, rays, opt, L, aux, out, tile_state, (int)nb);                         \
                return true;                                                                                      \
            }
            switch (opt.basis_dim) {
                case 1: SVOXT_ROLES_XF(1)
                case 4: SVOXT_ROLES_XF(4)
                case 9: SVOXT_ROLES_XF(9)
            }
#undef SVOXT_ROLES_XF
#undef SVOXT_ROLES_XF1
            return false;
        }
        if (opt.format == FMT_RGBA && tr.K == 4) SVOXT_ROLES(FMT_RGBA, 0)
        if (opt.format == FMT_SH) {
            switch (opt.basis_dim) {
                case 1: SVOXT_ROLES(FMT_SH, 1)
                case 4: SVOXT_ROLES(FMT_SH, 4)
                case 9: SVOXT_ROLES(FMT_SH, 9)
                case 16: SVOXT_ROLES(FMT_SH, 16)      // (r03: 0.833 -> 0.805 ms forward+backward; SH25, at 129 registers, loses: 1.139 -> 1.194)
            }
        }
#undef SVOXT_ROLES
        return false;
    }
}

template <bool N2, bool STOP>
bool launch_fwd_split(const TreeDev& tr, const RaysDev& rays, const Opts& opt, float* out,
                      RecLists L, uint4* aux, bool xf, bool fast, hipStream_t st,
                      const uint32_t* sigma_mask = nullptr, int32_t* tile_state = nullptr, int tflags = 0) {
    const unsigned nb = nblocks(rays.Q);
    if (xf && !N2) return false;
    const bool acc = N2 && tr.accel != nullptr;
    if (tile_state != nullptr && sigma_mask != nullptr && !STOP &&
        launch_fwd_roles<N2, STOP>(tr, rays, opt, out, L, aux, st, sigma_mask, tile_state, tflags, xf))
        return true;
    if constexpr (!STOP) {
        if (sigma_mask != nullptr) {
            if (acc) hipLaunchKernelGGL((march_rec_kernel<N2, false, 1, true>), dim3(nb), dim3(kBlock), 0, st, tr, rays, opt, L, aux, sigma_mask);
            else hipLaunchKernelGGL((march_rec_kernel<N2, false, 0, true>), dim3(nb), dim3(kBlock), 0, st, tr, rays, opt, L, aux, sigma_mask);
        }
    }
    if (STOP || sigma_mask == nullptr) {
        if (acc) hipLaunchKernelGGL((march_rec_kernel<N2, STOP, 1>), dim3(nb), dim3(kBlock), 0, st, tr, rays, opt, L, aux, (const uint32_t*)nullptr);
        else hipLaunchKernelGGL((march_rec_kernel<N2, STOP, 0>), dim3(nb), dim3(kBlock), 0, st, tr, rays, opt, L, aux, (const uint32_t*)nullptr);
    }
    return launch_shade<N2, STOP>(tr, rays, opt, out, L, aux, xf, fast, nb, st);
}

}  // namespace

extern "C" {

int svoxt_abi_version(void) { return SVOXT_ABI_VERSION; }

int64_t svoxt_set_super_tile_rows(int64_t rows) {
    const int64_t before = g_super_tile_rows;
    g_super_tile_rows = rows < 0 ? kSuperTileRows : rows;
    return before;
}

int32_t svoxt_image_walk(const svoxt_tree* tree, const svoxt_rays* rays) {
    if (tree == nullptr || rays == nullptr) return -1;
    const RaysDev d = to_dev(rays, tree);
    if (d.tiles_per_row == 0) return 0;
    return d.super_tiles ? SVOXT_LISTS_WALK_SUPER : SVOXT_LISTS_WALK_ROWMAJOR;
}

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
    if (((uintptr_t)l->tile_state & 7u) != 0) return fail(SVOXT_ERR_INVALID, "%s: lists.tile_state must be 8-byte aligned (64-bit queue entries)", fn);
    if ((l->flags & SVOXT_LISTS_WALK_ROWMAJOR) && (l->flags & SVOXT_LISTS_WALK_SUPER))
        return fail(SVOXT_ERR_INVALID, "%s: lists.flags name two tile walks", fn);
    // (r05: any thresholds -- the lists hold every sample with sigma > 0, the forward's thresholds decide what it composites)
    (void)opt;
    return SVOXT_OK;
}

// a recording forward starts with an empty block table and pool (pooled lists only)
static int lists_begin(const svoxt_sample_lists* l, int64_t Q, hipStream_t st, const char* fn) {
    if (l->blocktab == nullptr || (l->flags & SVOXT_LISTS_BEGUN)) return SVOXT_OK;
    const size_t n = (size_t)(rec_rays(Q) / 64) * (l->max_samples / kRecBlock) * sizeof(int32_t);
    const size_t nc = sizeof(int32_t) * kSubPools * kSubPoolStride;
    const size_t ns = l->tile_state != nullptr ? (size_t)roles_state_words(rec_rays(Q) / 64) * sizeof(int32_t) : 0;   // states, counters, queues: all -1
    hipError_t e;
    if (reinterpret_cast<char*>(l->blocktab) + n == reinterpret_cast<char*>(l->pool_next)) {
        // counters right behind the table (and the tile states right behind the counters): one fill
        const bool all = ns > 0 && reinterpret_cast<char*>(l->pool_next) + nc == reinterpret_cast<char*>(l->tile_state);
        e = hipMemsetAsync(l->blocktab, 0xff, n + nc + (all ? ns : 0), st);
        if (e == hipSuccess && ns > 0 && !all) e = hipMemsetAsync(l->tile_state, 0xff, ns, st);
    } else {
        e = hipMemsetAsync(l->blocktab, 0xff, n, st);
        if (e == hipSuccess) e = hipMemsetAsync(l->pool_next, 0xff, nc, st);
        if (e == hipSuccess && ns > 0) e = hipMemsetAsync(l->tile_state, 0xff, ns, st);
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
    const RaysDev rd = to_dev(rays, tree, lists);
    const Opts od = to_dev(opt);
    const bool n2 = tree->N == 2;
    bool done = false;
    // two kernels (march, shade per tile) where there is room for the lists: the caller's,
    // or scratch (then the stop rule applies while marching -- those lists serve no backward)
    // what the lists (the caller's, or the scratch) say about the kernels that fill them, and the scratch
    // forwards' own `flags` argument (SVOXT_FWD_FAST_SIGMOID = SVOXT_LISTS_NATIVE_MATH)
    const int32_t lflags = flags | (lists != nullptr ? lists->flags : 0) | (scratch != nullptr ? scratch->flags : 0);
    const bool fast = (lflags & SVOXT_LISTS_NATIVE_MATH) != 0;
    const bool want_terms = lists != nullptr && lists->terms != nullptr && C == 3;
    if (lists != nullptr && lobes_payload(od, tree->K)) {
        // SG / ASG with sample lists: only as march + tile shade that leaves the backward's hand-over (svoxt_can_record)
        if (!want_terms || !full_comp(opt) || uses_xform(tree, opt) || tree->weight_accum != nullptr)
            return fail(SVOXT_ERR_UNSUPPORTED, "%s: SG / ASG payloads record sample lists only with lists.terms (the exact per-tile backward's hand-over), all components, no transformation_matrices, no weight accumulation", fn);
        const uint32_t* smask = (tree->sigma_mask != nullptr && tree->sigma_mask_thresh == 0.f)      // (lists: every sigma > 0)
                                    ? reinterpret_cast<const uint32_t*>(tree->sigma_mask) : nullptr;
        uint4* aux = reinterpret_cast<uint4*>(lists->aux);
        int32_t* states = (lists->blocktab != nullptr && !(lflags & SVOXT_LISTS_FWD_NO_OVERLAP))
                              ? reinterpret_cast<int32_t*>(lists->tile_state) : nullptr;
        done = n2 ? launch_lobes_fwd_record<true>(tr, rd, od, out, lists_dev(lists, rays->Q), aux, st, smask, states, (lflags >> 8) & 15)
                  : launch_lobes_fwd_record<false>(tr, rd, od, out, lists_dev(lists, rays->Q), aux, st, smask, nullptr);
        if (!done) return fail(SVOXT_ERR_UNSUPPORTED, "%s: no recording kernel for this SG / ASG payload", fn);
        return check_launch(fn);
    }
    if (fwd_split_enabled(tree, opt, want_terms, lflags) && full_comp(opt) && fwd_split_payload(tree, opt, C) &&
        (!uses_xform(tree, opt) || xform_special(tree, opt))) {
        const bool xf = uses_xform(tree, opt);
        // one bit per feature row, built for this feature table and this sigma_thresh (else ignored); lists for a
        // backward hold every sample with sigma > 0: their march wants the mask of threshold 0
        const uint32_t* smask = (tree->sigma_mask != nullptr && tree->sigma_mask_thresh == (lists != nullptr ? 0.f : opt->sigma_thresh))
                                    ? reinterpret_cast<const uint32_t*>(tree->sigma_mask) : nullptr;
        // (tile states: march and shade as ONE launch; pooled lists only -- their fill clears the states too)
        auto states = [&](const svoxt_sample_lists* l) -> int32_t* {
            return (l->blocktab != nullptr && !(lflags & SVOXT_LISTS_FWD_NO_OVERLAP)) ? reinterpret_cast<int32_t*>(l->tile_state) : nullptr;
        };
        if (lists != nullptr) {
            uint4* aux = reinterpret_cast<uint4*>(lists->aux);
            done = n2 ? launch_fwd_split<true, false>(tr, rd, od, out, lists_dev(lists, rays->Q), aux, xf, fast, st, smask, states(lists), (lflags >> 8) & 15)
                      : launch_fwd_split<false, false>(tr, rd, od, out, lists_dev(lists, rays->Q), aux, xf, fast, st, smask);
        } else if (scratch != nullptr && smask != nullptr &&
                   (opt->stop_thresh == 0.f || (opt->format == SVOXT_FORMAT_RGBA && (tree->K == 8 || tree->K == 16 || tree->K == 32)))) {
            // With stop_thresh = 0 the stop rule ends a ray only once its transmittance is exactly 0; every
            // later sample then has weight 0 * (1 - att) = 0 and the final rescale is by 1 / (1 - 0): the
            // outputs are the same without it, and the march needs no sigma value -- the bitmask will do.
            // (r05) Rows of 8 / 16 / 32 floats take this route under a stop threshold too: the rule is then applied
            // by the SHADE (shade_chan_kernel's STOP instance; the march lists the whole ray), which keeps the
            // bitmask and the exponentials table -- depth 9 / K = 32 at 1024 x 1024, fast=True forward: 1.05 -> 0.8x ms;
            // 3-channel payloads stay with the march that stops (800 x 800 / depth 8: 0.173 ms against 0.204).
            if ((rc = lists_begin(scratch, rays->Q, st, fn))) return rc;
            uint4* aux = reinterpret_cast<uint4*>(scratch->aux);
            done = n2 ? launch_fwd_split<true, false>(tr, rd, od, out, lists_dev(scratch, rays->Q), aux, xf, fast, st, smask, states(scratch), (lflags >> 8) & 15)
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
    // (r05) Scratch lists this forward will not use (no two-kernel forward for the payload, or the tree accumulates weights):
    // their block counters are still left "nothing taken", so that a caller who sizes its pool by them reads a defined state.
    if (scratch != nullptr && (rc = lists_begin(scratch, rays->Q, st, fn))) return rc;
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

static int sigma_mask_build(const svoxt_tree* tree, float sigma_thresh, void* mask, void* fill, int64_t fill_bytes, void* stream,
                            const char* fn) {
    int rc;
    if ((rc = check_tree(tree, fn))) return rc;
    if (fill_bytes < 0 || (fill_bytes > 0 && (fill == nullptr || ((uintptr_t)fill & 15u) != 0 || (fill_bytes & 15) != 0)))
        return fail(SVOXT_ERR_INVALID, "%s: fill is NULL or not 16-byte aligned, or fill_bytes not a multiple of 16", fn);
    if (tree->M == 0 && fill_bytes == 0) return SVOXT_OK;
    if (tree->M > 0 && (mask == nullptr || ((uintptr_t)mask & 7u) != 0))
        return fail(SVOXT_ERR_INVALID, "%s: mask is NULL or not 8-byte aligned", fn);
    const int64_t words = (tree->M + 63) / 64;
    const unsigned mb = (unsigned)((words + 15) / 16);
    const int64_t fv = fill_bytes / 16;
    const unsigned fb = (unsigned)((fv + 1023) / 1024);                      // 256 threads x 4 words
    hipLaunchKernelGGL(svoxt::sigma_mask_kernel, dim3(mb + fb), dim3(256), 0, (hipStream_t)stream,
                       tree->features, tree->M, tree->K, sigma_thresh, reinterpret_cast<unsigned long long*>(mask),
                       mb, reinterpret_cast<uint4*>(fill), fv);
    return check_launch(fn);
}

int svoxt_sigma_mask_build(const svoxt_tree* tree, float sigma_thresh, void* mask, void* stream) {
    return sigma_mask_build(tree, sigma_thresh, mask, nullptr, 0, stream, "svoxt_sigma_mask_build");
}

int svoxt_sigma_mask_build_fill(const svoxt_tree* tree, float sigma_thresh, void* mask, void* fill, int64_t fill_bytes, void* stream) {
    return sigma_mask_build(tree, sigma_thresh, mask, fill, fill_bytes, stream, "svoxt_sigma_mask_build_fill");
}

int svoxt_exp_table_build(const svoxt_tree* tree, float sigma_thresh, void* mask, float* table, void* stream) {
    const char* fn = "svoxt_exp_table_build";
    int rc;
    if ((rc = check_tree(tree, fn))) return rc;
    if (tree->K != 8 && tree->K != 16 && tree->K != 32)
        return fail(SVOXT_ERR_UNSUPPORTED, "%s: the table serves RGBA-style rows of 8, 16 or 32 floats", fn);
    if (tree->M == 0) return SVOXT_OK;
    if (table == nullptr || ((uintptr_t)table & 15u) != 0 || ((uintptr_t)tree->features & 15u) != 0)
        return fail(SVOXT_ERR_INVALID, "%s: table is NULL, or table / features not 16-byte aligned", fn);
    if (mask != nullptr && ((uintptr_t)mask & 7u) != 0) return fail(SVOXT_ERR_INVALID, "%s: mask not 8-byte aligned", fn);
    const int64_t threads = tree->M * (tree->K / 4);
    const dim3 grid((unsigned)((threads + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    uint8_t* mb = reinterpret_cast<uint8_t*>(mask);
    if (tree->K == 8) hipLaunchKernelGGL((svoxt::exp_table_kernel<8>), grid, dim3(256), 0, st, tree->features, tree->M, sigma_thresh, mb, table);
    else if (tree->K == 16) hipLaunchKernelGGL((svoxt::exp_table_kernel<16>), grid, dim3(256), 0, st, tree->features, tree->M, sigma_thresh, mb, table);
    else hipLaunchKernelGGL((svoxt::exp_table_kernel<32>), grid, dim3(256), 0, st, tree->features, tree->M, sigma_thresh, mb, table);
    return check_launch(fn);
}

int svoxt_fwd_fills_terms(const svoxt_tree* tree, const svoxt_options* opt, int32_t list_flags) {
    if (tree == nullptr || opt == nullptr || !svoxt_can_record(tree, opt)) return 0;
    const int C = svoxt_out_data_dim(opt, tree->K) - 1;
    if (uses_xform(tree, opt))      // (r04) view rotations: only the two-kernel / one-launch forward of SH 1 / 4 / 9 leaves a hand-over
        return (C == 3 && tree->N == 2 && opt->format == SVOXT_FORMAT_SH && opt->basis_dim <= 9 && (list_flags & SVOXT_LISTS_FWD_TWO_KERNELS) &&
                full_comp(opt) && fwd_split_payload(tree, opt, C) && xform_special(tree, opt)) ? 3 : 0;
    if (C != 3) return 0;
    if (lobes_payload(to_dev(opt), tree->K)) return 3;
    // 3: the two-kernel forward (tile shade kernel, position-major); 2: the one-kernel forward (lane-major lines)
    return (fwd_split_enabled(tree, opt, true, list_flags) && full_comp(opt) && fwd_split_payload(tree, opt, C)) ? 3 : 2;
}

int svoxt_can_record(const svoxt_tree* tree, const svoxt_options* opt) {
    if (tree == nullptr || opt == nullptr) return 0;
    if (!full_comp(opt)) return 0;        // (r05: any thresholds)
    if (uses_xform(tree, opt)) return xform_special(tree, opt) ? 1 : 0;
    const int C = svoxt_out_data_dim(opt, tree->K) - 1;
    if (opt->format == SVOXT_FORMAT_RGBA) return (C == 3 || C == 7 || C == 15 || C == 31) ? 1 : 0;
    if (opt->format == SVOXT_FORMAT_SH && C == 3 && tree->K == 3 * opt->basis_dim + 1)
        return (opt->basis_dim == 1 || opt->basis_dim == 4 || opt->basis_dim == 9 || opt->basis_dim == 16 ||
                opt->basis_dim == 25) ? 1 : 0;
    // SG / ASG (r03): 2 = lists can be recorded, but only together with the exact per-tile backward's hand-over
    // (lists.terms; svoxt_fwd_fills_terms says 3) on N = 2 trees without weight accumulation -- they serve that backward alone
    if (C == 3 && lobes_payload(to_dev(opt), tree->K) && tree->N == 2 && tree->weight_accum == nullptr) return 2;
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
    if (fwd_out != nullptr && (opt->sigma_thresh != 0.f || opt->stop_thresh != 0.f))
        return fail(SVOXT_ERR_UNSUPPORTED, "%s: fwd_out (the single-march backward) needs sigma_thresh == stop_thresh == 0: "
                    "with thresholds the forward's output is not the sum the backward differentiates", fn);
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
    if (tree->N == 2) hipLaunchKernelGGL((opacity_fwd_kernel<true>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays, tree), to_dev(opt), out);
    else hipLaunchKernelGGL((opacity_fwd_kernel<false>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays, tree), to_dev(opt), out);
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
    if (tree->N == 2) hipLaunchKernelGGL((opacity_fwd_kernel<true, true>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays, tree, lists), to_dev(opt), out, L, aux);
    else hipLaunchKernelGGL((opacity_fwd_kernel<false, true>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays, tree, lists), to_dev(opt), out, L, aux);
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
    const RaysDev rd = to_dev(rays, tree, lists);
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
    if (tree->N == 2) hipLaunchKernelGGL((depth_kernel<true>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays, tree), to_dev(opt), depth);
    else hipLaunchKernelGGL((depth_kernel<false>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays, tree), to_dev(opt), depth);
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
    if (tree->N == 2) hipLaunchKernelGGL((count_fwd_kernel<true>), dim3(nb), dim3(kBlock), 0, st, tr, to_dev(rays, tree), to_dev(opt), c);
    else hipLaunchKernelGGL((count_fwd_kernel<false>), dim3(nb), dim3(kBlock), 0, st, tr, to_dev(rays, tree), to_dev(opt), c);
    return check_launch(fn);
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
    if (tree->N == 2) hipLaunchKernelGGL((count_touched_kernel<true>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays, tree), to_dev(opt), row_mask, tree_mask, n_slots, lg);
    else hipLaunchKernelGGL((count_touched_kernel<false>), dim3(nb), dim3(kBlock), 0, st, to_dev(tree), to_dev(rays, tree), to_dev(opt), row_mask, tree_mask, n_slots, lg);
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

static int compact_rows(const float* src, float* src_clear, int64_t M, int32_t K, int32_t stride, float* dst, void* stream, const char* fn);

int svoxt_compact_rows(const float* src, int64_t M, int32_t K, int32_t stride, float* dst, void* stream) {
    return compact_rows(src, nullptr, M, K, stride, dst, stream, "svoxt_compact_rows");
}

int svoxt_compact_rows_clear(float* src, int64_t M, int32_t K, int32_t stride, float* dst, void* stream) {
    return compact_rows(src, src, M, K, stride, dst, stream, "svoxt_compact_rows_clear");
}

static int compact_rows(const float* src, float* src_clear, int64_t M, int32_t K, int32_t stride, float* dst, void* stream, const char* fn) {
    if (M < 0 || K < 1 || stride < K) return fail(SVOXT_ERR_INVALID, "%s: bad extents", fn);
    if (M == 0) return SVOXT_OK;
    if (src == nullptr || dst == nullptr) return fail(SVOXT_ERR_INVALID, "%s: src / dst is NULL", fn);
    typedef float v4f __attribute__((ext_vector_type(4)));
    const bool vec = K % 4 == 0 && stride % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
    const int64_t n = src_clear != nullptr ? (vec ? M * (stride / 4) : M * (int64_t)stride)       // (the clearing form covers the padded rows)
                                           : (vec ? M * (K / 4) : M * K);
    const int64_t want = (n + kBlock - 1) / kBlock;
    const unsigned nb = (unsigned)(want < 16384 ? want : 16384);
    if (src_clear != nullptr && src_clear == dst) return fail(SVOXT_ERR_INVALID, "%s: src and dst are the same buffer", fn);
    if (vec && src_clear != nullptr && stride == 32 && n < (int64_t)0xfff00000) {
        constexpr int U = 4;
        const unsigned g = (unsigned)((n + U * 256 - 1) / (U * 256));
        hipLaunchKernelGGL((compact_rows_clear_pow2_kernel<8, U>), dim3(g), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<float4*>(src_clear), (uint32_t)n, (int)(K / 4), reinterpret_cast<float4*>(dst));
    } else if (vec && src_clear != nullptr)
        hipLaunchKernelGGL((compact_rows_kernel<v4f, true>), dim3(nb), dim3(kBlock), 0, (hipStream_t)stream,
                           reinterpret_cast<v4f*>(src_clear), n, (int)(K / 4), (int)(stride / 4), reinterpret_cast<v4f*>(dst));
    else if (vec)
        hipLaunchKernelGGL((compact_rows_kernel<v4f>), dim3(nb), dim3(kBlock), 0, (hipStream_t)stream,
                           reinterpret_cast<const v4f*>(src), n, (int)(K / 4), (int)(stride / 4), reinterpret_cast<v4f*>(dst));
    else if (src_clear != nullptr)
        hipLaunchKernelGGL((compact_rows_kernel<float, true>), dim3(nb), dim3(kBlock), 0, (hipStream_t)stream,
                           src_clear, n, (int)K, (int)stride, dst);
    else
        hipLaunchKernelGGL((compact_rows_kernel<float>), dim3(nb), dim3(kBlock), 0, (hipStream_t)stream,
                           src, n, (int)K, (int)stride, dst);
    return check_launch(fn);
}

int64_t svoxt_accel_bytes(int32_t log2_res, int64_t n_internal) {
    log2_res &= ~SVOXT_ACCEL_BRICKS;                         // (the same size in either layout)
    if (log2_res < 1 || log2_res > 8 || n_internal < 0) return -1;
    return ((int64_t)sizeof(uint32_t) << (3 * log2_res)) + (int64_t)sizeof(uint2) * 8 * n_internal;
}

int svoxt_accel_build(const svoxt_tree* tree, int32_t log2_res, void* cells, void* stream) {
    const char* fn = "svoxt_accel_build";
    int rc;
    if ((rc = check_tree(tree, fn))) return rc;
    if (tree->N != 2) return fail(SVOXT_ERR_UNSUPPORTED, "%s: the acceleration grid exists for N == 2 only", fn);
    const bool bricks = (log2_res & SVOXT_ACCEL_BRICKS) != 0;
    log2_res &= ~SVOXT_ACCEL_BRICKS;
    if (log2_res < (bricks ? 2 : 1) || log2_res > 8)
        return fail(SVOXT_ERR_INVALID, "%s: log2_res must be in [1, 8] (with SVOXT_ACCEL_BRICKS: [2, 8])", fn);
    if (cells == nullptr || ((uintptr_t)cells & 7u) != 0) return fail(SVOXT_ERR_INVALID, "%s: cells is NULL or not 8-byte aligned", fn);
    if (tree->M >= (int64_t)kAccelIdx || tree->n_internal >= (int64_t)kAccelIdx)
        return fail(SVOXT_ERR_UNSUPPORTED, "%s: the 4-byte cells hold row and node indices below 2^27 - 1", fn);
    TreeDev tr = to_dev(tree);
    tr.accel = nullptr;
    const unsigned n = 1u << (3 * log2_res);
    hipLaunchKernelGGL(accel_build_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       (hipStream_t)stream, tr, (int)log2_res, bricks, reinterpret_cast<uint32_t*>(cells));
    const int64_t slots = tree->n_internal * 8;
    if (slots > 0)
        hipLaunchKernelGGL(accel_nodes_kernel, dim3((unsigned)((slots + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           (hipStream_t)stream, tr.child, tr.data, slots,
                           reinterpret_cast<uint2*>(reinterpret_cast<uint32_t*>(cells) + n));
    return check_launch(fn);
}

}  // extern "C"
