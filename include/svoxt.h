/* svoxt.h -- C ABI of libsvoxt_hip.so, the MI355X (gfx950) implementation of
 * svox_t's differentiable volume-render hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / pybind
 * types.  Each entry point replaces one function of the reference's pybind11
 * module `svox_t.csrc` (reference paths are relative to the svox_t checkout):
 *
 *   svoxt_volume_render_fwd   <- volume_render            svox_t/csrc/svox.cpp:55,128   (rt_kernel.cu:1362-1379)
 *   svoxt_volume_render_bwd   <- volume_render_backward   svox_t/csrc/svox.cpp:57,130   (rt_kernel.cu:1402-1426)
 *   svoxt_opacity_render_fwd  <- opacity_render           svox_t/csrc/svox.cpp:68,138   (rt_kernel.cu:1574-1591)
 *   svoxt_opacity_render_bwd  <- opacity_render_backward  svox_t/csrc/svox.cpp:69,139   (rt_kernel.cu:1593-1616)
 *   svoxt_render_depth        <- render_depth             svox_t/csrc/svox.cpp:63,134   (rt_kernel.cu:1506-1523)
 *   svoxt_query_fwd           <- query_vertical           svox_t/csrc/svox.cpp:45,119   (svox_kernel.cu:274-324)
 *   svoxt_query_bwd           <- query_vertical_backward  svox_t/csrc/svox.cpp:46,120   (svox_kernel.cu:380-402)
 *   svoxt_tree / svoxt_rays / svoxt_options
 *                             <- TreeSpec / RaysSpec / RenderOptions
 *                                                         svox_t/csrc/include/data_spec.hpp:52-145
 *
 * Entry points without a counterpart in the reference, all optional and
 * result-neutral (a caller that ignores them gets the same numbers, slower):
 *   svoxt_accel_build / svoxt_accel_bytes            cached prefix of the root descent
 *   svoxt_ray_order                                  coherent order for ray batches that are not images
 *   svoxt_volume_render_fwd_record / _bwd_replay     backward without tree traversal
 *   svoxt_can_record, svoxt_bwd_workspace_bytes, svoxt_compact_rows(_clear), svoxt_count_fwd,
 *   svoxt_query_leaves (the reference's mask compaction, made deterministic)
 *
 * Conventions
 *   - Every pointer marked "device" is HBM memory of the current HIP device;
 *     the caller owns all buffers (inputs are borrowed, outputs are written in
 *     place, nothing is allocated or freed by the library).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *     calls are asynchronous with respect to the host: they enqueue work and
 *     return; no call synchronises.
 *   - Return value 0 = success; otherwise one of SVOXT_ERR_* and
 *     svoxt_last_error() describes the failure (thread-local string).
 *     Unlike the reference (which only printf's launch errors,
 *     svox_t/csrc/include/common.cuh:108-111) launch failures are returned.
 *   - fp32 only (the reference dispatches fp64 nominally, but every Python
 *     path builds fp32 tensors).
 */
#ifndef SVOXT_H_
#define SVOXT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVOXT_ABI_VERSION 22

enum {
    SVOXT_OK = 0,
    SVOXT_ERR_INVALID = 1,      /* null pointer, bad extent, inconsistent options */
    SVOXT_ERR_UNSUPPORTED = 2,  /* valid for the reference, not implemented here  */
    SVOXT_ERR_HIP = 3           /* HIP runtime / launch failure                   */
};

/* enum DataFormat, svox_t/csrc/include/data_spec.hpp:45-50 */
enum {
    SVOXT_FORMAT_RGBA = 0,
    SVOXT_FORMAT_SH = 1,
    SVOXT_FORMAT_SG = 2,
    SVOXT_FORMAT_ASG = 3
};

/* TreeSpec (data_spec.hpp:67-111) as the kernels need it.  `data` is NOT the
 * feature storage: it holds, per leaf slot, the row index into `features`; a
 * slot is empty iff its index >= M (rt_kernel.cu:269). */
typedef struct svoxt_tree {
    const float*   features;     /* device [M, K] row-major; row = colour/SH coeffs..., sigma last */
    int64_t        M;
    int32_t        K;            /* data_dim */
    int32_t        N;            /* branching factor per axis (2 = octree) */
    const int32_t* data;         /* device [capacity, N, N, N, 1] feature-row index per slot */
    const int32_t* child;        /* device [capacity, N, N, N] relative child offset, 0 = leaf */
    int64_t        n_internal;   /* rows of child/data in use (TreeSpec.n_internal) */
    const float*   offset;       /* device [3]  world -> tree: p' = offset + scaling * p */
    const float*   scaling;      /* device [3] */
    const float*   extra_data;   /* device [extra_rows, extra_cols] SG/ASG lobes, or NULL */
    int32_t        extra_rows;
    int32_t        extra_cols;
    float*         weight_accum; /* device [capacity * N^3] or NULL (TreeSpec._weight_accum) */
    const float*   xform;        /* device [M,d,d] TreeSpec.transformation_matrices (per-leaf rotation of the view
                                    direction by the upper-left 3x3, rt_kernel.cu:283-291) or NULL;
                                    volume_render fwd/bwd only (ignored by the other entry points); d = xform_dim below */
    const void*    accel;        /* device, optional: acceleration grid built by svoxt_accel_build for THIS
                                    child/data content (N == 2), or NULL.  Pure cache: results are identical
                                    with or without it; rebuild after any change to child or data. */
    int32_t        accel_log2;   /* log2 of the grid resolution per axis the grid was built with, | SVOXT_ACCEL_BRICKS
                                    if it was built in that layout (the value handed to svoxt_accel_build) */
    int32_t        xform_dim;    /* rows = columns of each xform matrix: 3, or 4 (the [M,4,4] that
                                    warp_vertices / blend_transformation_matrix produce); 0 means 3 */
    const void*    sigma_mask;   /* device, optional: one bit per feature row, set iff the row's sigma (its last
                                    float) > sigma_mask_thresh, built by svoxt_sigma_mask_build for THIS feature
                                    content, or NULL.  Pure cache like accel (results are identical with or
                                    without it; rebuild after any change to the features): the two-kernel
                                    forward's march reads a bit instead of gathering sigma. Ignored unless
                                    sigma_mask_thresh == options.sigma_thresh. */
    float          sigma_mask_thresh;
    int32_t        reserved0;    /* 0 */
    const float*   exp_table;    /* device, optional (ABI v17; RGBA-style rows of K = 8 / 16 / 32 floats only): [M, K], entry (row, c)
                                    = the exponential exp(-features[row][c]) exactly as the kernels form it, for the K - 1
                                    feature columns, and sigma itself in the last column, built by svoxt_exp_table_build for
                                    THIS feature content, or NULL.  Pure cache like accel and sigma_mask (results are
                                    identical with or without it): the sigmoids of such a payload do not depend on the view,
                                    so the shade kernel of the two-kernel forward and both sweeps of the per-tile backward
                                    read the table instead of forming the exponential once per sample. */
} svoxt_tree;

/* RaysSpec (data_spec.hpp:52-65); with c2w set, CameraSpec (data_spec.hpp:113-126):
 * the image-mode entry points volume_render_image / volume_render_image_backward
 * (rt_kernel.cu:1382-1399, 1428-1452) are every render call below given a camera. */
typedef struct svoxt_rays {
    const float* origins;        /* device [Q, 3] */
    const float* dirs;           /* device [Q, 3] */
    const float* vdirs;          /* device [Q, 3] */
    int64_t      Q;
    int32_t      image_width;    /* optional hint (no counterpart in the reference): if the batch is a   */
    int32_t      image_height;   /* row-major W x H image (Q == W*H, both multiples of 8) the kernels walk
                                    it in 8x8 pixel tiles -- row-major, or in super-tiles of 8 x 8 tiles where the
                                    tree has more than 2^21 feature rows (a rule of M alone: a forward that
                                    records sample lists and the backward that walks them use the same walk);
                                    0 = no hint.  Results do not depend on it. */
    const float* c2w;            /* camera mode when non-NULL: device camera-to-world matrix, rows of 4 floats
                                    ([3,4] or [4,4] contiguous).  Ray q is then pixel (q % image_width,
                                    q / image_width) of a pinhole camera, generated inside the kernels as
                                    cam2world_ray + maybe_world2ndc do (rt_kernel.cu:1153-1190; the NDC warp
                                    applies iff options.ndc_width >= 0, and the view direction is the one
                                    before the warp, :1203); origins / dirs / vdirs are ignored, Q must be
                                    image_width * image_height, outputs are the [H, W, C+1] image. */
    float        fx, fy;         /* focal lengths in pixels (camera mode) */
    const int32_t* order;        /* optional (ABI v15, no counterpart in the reference): device int32 [Q], a permutation
                                    of 0 .. Q-1 (svoxt_ray_order's): launch thread i works on ray order[i], so
                                    that the 64 rays of a wavefront are the ones the permutation puts next to
                                    each other -- nothing is gathered or scattered, every ray's inputs and results
                                    stay at the ray's own index.  Sample lists recorded with an order must be
                                    replayed with the same one.  NULL: thread i takes ray i (or the image tiles
                                    above); not allowed together with the image hint or camera mode. */
} svoxt_rays;

/* RenderOptions (data_spec.hpp:129-145), same fields in the same order. */
typedef struct svoxt_options {
    float   step_size;
    float   background_brightness;
    int32_t format;
    int32_t basis_dim;
    int32_t ndc_width;
    int32_t ndc_height;
    float   ndc_focal;
    int32_t min_comp;
    int32_t max_comp;
    float   sigma_thresh;
    float   stop_thresh;
} svoxt_options;

int         svoxt_abi_version(void);
const char* svoxt_last_error(void);

/* get_out_data_dim (rt_kernel.cu:1352-1358): number of output columns C+1 of
 * volume_render for a feature width K. Returns -1 for invalid options. */
int svoxt_out_data_dim(const svoxt_options* opt, int32_t K);

/* out: device [Q, C+1], fully written (no pre-zeroing needed). */
int svoxt_volume_render_fwd(const svoxt_tree* tree, const svoxt_rays* rays,
                            const svoxt_options* opt, float* out, void* stream);

/* The same result (bit for bit) with device scratch: with svoxt_fwd_workspace_bytes(Q, S)
 * bytes (S >= 8, 96 is a good value) the forward can run as two kernels -- one that only
 * steps the rays through the tree and lists each ray's samples (up to S; longer rays finish in a
 * tail launch), one that shades the lists: per 64-ray tile with eight wavefronts sharing the
 * work (3-channel payloads; only on request: svoxt_volume_render_fwd_scratch with SVOXT_LISTS_FWD_TWO_KERNELS), or with the channels of a row on
 * the lanes of a wavefront (RGBA-style rows of 8 / 16 / 32 floats: the default for them) --
 * instead of one kernel as long as its longest ray (NOTEBOOK.md 5).  Other payloads,
 * tree->weight_accum, a NULL or too small workspace: exactly svoxt_volume_render_fwd.  Contents of the workspace on return are unspecified.
 * (Reference: volume_render, rt_kernel.cu:1362-1379; trace_ray :222-328.) */
/* flags: 0, or an OR of the SVOXT_LISTS_* values below (how the scratch lists are filled); among them
 * SVOXT_FWD_FAST_SIGMOID (= SVOXT_LISTS_NATIVE_MATH) -- an opt-in tolerance mode for
 * RGBA-style rows of 8 / 16 / 32 floats: bit-exact stepping, but expf and the per-channel quotient
 * w / (1 + exp(-x)) -- double precision in the reference (rt_kernel.cu:280, 300-305) -- are taken with the
 * hardware's v_exp_f32 / v_rcp_f32.  Outputs within 1e-5 relative (tests at full size); no longer bit for bit. */
#define SVOXT_FWD_FAST_SIGMOID 1
int64_t svoxt_fwd_workspace_bytes(int64_t Q, int32_t max_samples);
int svoxt_volume_render_fwd_ws(const svoxt_tree* tree, const svoxt_rays* rays,
                               const svoxt_options* opt, float* out,
                               void* workspace, int64_t workspace_bytes, int32_t flags, void* stream);
/* The same with the scratch given as a svoxt_sample_lists (declared below; rec / aux, optionally
 * pooled through blocktab): what svoxt_volume_render_fwd_ws does with a dense workspace, for callers
 * that want the pool.  The lists come back in an unspecified state: they serve no backward (early
 * termination applies while they are written). */
struct svoxt_sample_lists;
int svoxt_volume_render_fwd_scratch(const svoxt_tree* tree, const svoxt_rays* rays,
                                    const svoxt_options* opt, float* out,
                                    const struct svoxt_sample_lists* scratch, int32_t flags, void* stream);

/* grad_out: device [Q, grad_cols] with grad_cols = C+1.
 * grad_features: device [M, grad_stride] floats of which columns 0..K-1 are the
 *           gradient; zeroed by this call on `stream`, then accumulated with
 *           float atomics (as the reference: zeros_like + atomicAdd,
 *           rt_kernel.cu:1415,413,486).  grad_stride = 0 or K is the
 *           reference's dense [M, K]; a stride that makes rows start on 64-byte
 *           boundaries (e.g. 32 for K = 28: rows of 128 bytes) lets each row's
 *           atomics hit two memory-side requests (the rate those are taken at is
 *           what bounds the backward): 2 instead of 2.75 on average for the
 *           one-kernel backward, 2 instead of 3.7 for the two-kernel one (its second
 *           kernel 0.30 -> 0.20 ms); svoxt_compact_rows turns the result into the
 *           dense table (0.03 ms for 75 MB).
 * workspace: device scratch of `workspace_bytes` bytes, or NULL.  With
 *           svoxt_bwd_workspace_bytes(Q, S) bytes the first pass records up to
 *           S composited samples per ray (8 bytes each) and the second pass
 *           replays them instead of traversing the tree a second time; rays
 *           with more samples than S march the remainder, so any S >= 0 gives
 *           the same result.  Contents on return are unspecified. */
int64_t svoxt_bwd_workspace_bytes(int64_t Q, int32_t max_samples);
int svoxt_volume_render_bwd(const svoxt_tree* tree, const svoxt_rays* rays,
                            const svoxt_options* opt, const float* grad_out,
                            int32_t grad_cols, float* grad_features, int32_t grad_stride,
                            void* workspace, int64_t workspace_bytes, void* stream);

/* Sample lists (no counterpart in the reference).  When a forward will be
 * followed by a backward, the forward can record which samples each ray
 * composited -- 8-byte records (feature row, step length), the k-th record of the ray
 * handled by launch thread t at rec[block(t / 64, k / 8)][t % 64][k % 8] (a lane's 8
 * consecutive records are one 64-byte line; `block` is the identity for dense lists and
 * goes through blocktab for pooled ones, below), and aux[q] = (count, overflow flag,
 * resume point, final transmittance) -- and the backward then replays
 * those samples instead of traversing the tree again (twice, in the reference:
 * rt_kernel.cu:365-494).  Rays with more than max_samples composited samples
 * march the remainder, so the result does not depend on max_samples.  The lists hold EVERY
 * sample with sigma > 0 to the end of the ray -- what the reference's backward takes, which ignores
 * both thresholds (rt_kernel.cu:382, 456) -- and the recording forward applies its own rules while it
 * composites (ABI v20: sigma > sigma_thresh, :279; nothing after T <= stop_thresh, :313-319), so any
 * thresholds are served (`fast=True`); a sigma bitmask handed to a recording forward must be the one of
 * threshold 0.  Needs one of the specialised payloads (svoxt_can_record returns 1; with tree->xform
 * set: SH payloads on N = 2 trees); the lists are valid for the tree, features'
 * sign of sigma, rays and options they were recorded with.
 * svoxt_can_record returns 2 (r03) for SG / ASG payloads with 1 / 4 / 9 / 16 / 25 lobes and three channels on
 * N = 2 trees: their lists are recorded only TOGETHER WITH the exact per-tile backward's hand-over (lists.terms
 * given; svoxt_fwd_fills_terms says 3) and serve that backward alone (svoxt_volume_render_bwd_replay with
 * coef_bytes < 0, terms_state 3, fwd_out NULL); any other combination is SVOXT_ERR_UNSUPPORTED -- call the
 * marching svoxt_volume_render_bwd instead. */
/* svoxt_sample_lists.flags (ABI v16): how the kernels that write and read these lists work.
 *   SVOXT_LISTS_NATIVE_MATH      opt-in TOLERANCE mode for RGBA-style rows of 8 / 16 / 32 floats (= the `flags`
 *                                value SVOXT_FWD_FAST_SIGMOID of the scratch forwards): the stepping -- which
 *                                leaves a ray crosses, the step lengths, the lists -- stays bit-exact, the SHADING
 *                                arithmetic uses the hardware's exponential (v_exp_f32 on x * log2 e) and
 *                                reciprocal (v_rcp_f32) where the reference has expf and a double-precision
 *                                quotient (rt_kernel.cu:280, 300-305, 397, 408-425, 461-476): the channel-lane
 *                                shade / tail kernels of the forward and both sweeps of the per-tile backward.
 *                                Outputs within 1e-5 relative (+1e-6 absolute), gradients within 1e-5 of the
 *                                tight scale (tests at full size); not bit for bit.  A backward must be given the
 *                                flags its forward recorded with.  Ignored by every other payload / kernel.
 *   SVOXT_LISTS_FWD_ONE_KERNEL   the forward that fills these lists runs as one kernel / as march + shade
 *   SVOXT_LISTS_FWD_TWO_KERNELS  kernels where both exist (neither: the library's default for the payload:
 *                                two kernels for rows of 8 / 16 / 32 floats and for 3-channel forwards that
 *                                also fill `terms`).  Result-neutral. */
#define SVOXT_LISTS_NATIVE_MATH 1
#define SVOXT_LISTS_FWD_ONE_KERNEL 2
#define SVOXT_LISTS_FWD_TWO_KERNELS 4
/*   SVOXT_LISTS_FWD_NO_OVERLAP   march and shade of the two-kernel forward as two launches even where lists.tile_state
 *                                would let one launch carry both (below).  Result-neutral. */
#define SVOXT_LISTS_FWD_NO_OVERLAP 8
/*   SVOXT_LISTS_GRAD_ZEROED      (read by svoxt_volume_render_bwd_replay) grad_features is all zeros on entry and the call
 *                                need not fill it: the caller keeps a padded gradient scratch between steps and turns
 *                                it into the dense [M, K] gradient with svoxt_compact_rows_clear, which leaves it zeroed
 *                                again -- a fill of M * grad_stride floats per step less.  The caller answers for the
 *                                zeros (a scratch a failed call may have left half written must be filled again). */
#define SVOXT_LISTS_GRAD_ZEROED 16
/*   SVOXT_LISTS_BEGUN            (read by the recording forwards; ABI v18) blocktab, pool_next and tile_state hold -1
 *                                in every word already (svoxt_sigma_mask_build_fill on the same stream): the call
 *                                does not fill them again. */
#define SVOXT_LISTS_BEGUN 32
/*   SVOXT_LISTS_WALK_ROWMAJOR    (ABI v20) the order in which an image's 8 x 8 tiles are walked when these lists are written
 *   SVOXT_LISTS_WALK_SUPER       and read (svoxt_rays.image_width): lists are indexed by (tile, lane), so the backward that
 *                                replays them must walk the image the way the recording forward did.  Neither bit: the
 *                                library's rule (a function of tree->M and svoxt_set_super_tile_rows) at the time of EACH
 *                                call -- the caller answers for both calls seeing the same rule.  One bit set (what
 *                                svoxt_image_walk returns for the recording call: pass it to both): that walk, whatever
 *                                the rule says by the time of the backward. */
#define SVOXT_LISTS_WALK_ROWMAJOR 64
#define SVOXT_LISTS_WALK_SUPER 128
/* Test and measurement switches of the one-launch forward (fwd_roles_kernel; ABI v17), all result-neutral: its shading
 * workgroups drop every third tile they take / give up after one poll / treat what they load for every fifth tile as a
 * stale read would look -- the fallback launch must then deliver the same pixels, lists and hand-over -- and
 * SVOXT_LISTS_FWD_AGENT_FENCE: the march -> shade hand-over with an agent-scope release before the queue entry is
 * stored and an agent-scope acquire before the first load (the microarchitecture guide's valid form in full; the
 * default rests on same-XCD queues, acknowledged stores and sc1 loads, checked by a per-tile checksum). */
#define SVOXT_LISTS_TEST_DROP 256
#define SVOXT_LISTS_TEST_NOPOLL 512
#define SVOXT_LISTS_TEST_STALE 1024
#define SVOXT_LISTS_FWD_AGENT_FENCE 2048
typedef struct svoxt_sample_lists {
    void*   rec;           /* device, 64-byte aligned, max_samples * ceil(Q / 64) * 64 * 8 bytes: record k of the
                              ray handled by launch thread t lives at rec[t / 64][k / 8][t % 64][k % 8] (8 bytes
                              each): a lane's 8 consecutive records are one 64-byte line, written at once */
    void*   aux;           /* device, Q * 16 bytes: count | overflow, resume point, final transmittance, pad */
    int32_t max_samples;   /* S: a multiple of 8 in [8, 4096] */
    void*   coef;          /* device, coef_bytes >= max_samples * Q * 16 bytes (twice that with tree->xform:
                              the second half takes each sample's rotated view direction), 16-byte
                              aligned, or NULL.  Only read by
                              svoxt_volume_render_bwd_replay: with it (3-channel payloads of at most 32
                              floats per row, N = 2) the backward runs as two kernels -- the list walk
                              leaves each sample's contribution in factored form in rec / coef (rec is
                              overwritten), and a second kernel adds them up per 8x8 tile in LDS, so a
                              gradient row goes to memory once per tile instead of once per sample. */
    int64_t coef_bytes;    /* size of coef (a buffer too small for the route is ignored; a buffer that is given
                              selects the two-kernel form); -1 with coef NULL: take the per-tile route if it
                              can run as ONE kernel (no view rotations, fwd_out given: list walk and merge
                              fused, nothing goes through coef), else the one-kernel backward */
    void*   terms;         /* device, 16-byte aligned, terms_bytes >= 2 * the bytes of rec (16 per record slot), or NULL:
                              (att, e_0, e_1, e_2) of every 3-channel sample as the exact backward (fwd_out NULL,
                              the ONE-kernel form) needs them.  Given to svoxt_volume_render_fwd_record, the
                              one-kernel forward fills it (svoxt_fwd_fills_terms says whether it will); the
                              backward, told so by terms_state = 2 or 3 (what svoxt_fwd_fills_terms returned:
                              the layout the forward wrote), then gathers no feature row and forms no
                              exponential in either of its sweeps.  terms_state = 0 (3-channel payloads, since ABI v17): the
                              buffer is ignored and both sweeps gather the rows.  Same bits every way.
                              RGBA rows of 8 / 16 / 32 floats (C = 7 / 15 / 31): scratch between the two sweeps of
                              their exact backwards, 8 bytes per record slot for the per-tile kernel (attenuation,
                              second-pass total_color), 4 for the per-ray one; NULL or smaller: the per-ray list
                              replay that forms every sigmoid twice. */
    int64_t terms_bytes;
    /* Pooled lists (optional, ABI v12).  blocktab NULL: dense -- rec holds max_samples slots for every
     * ray, as described above.  blocktab given: device int32 [ceil(Q / 64) * max_samples / 8]; entry
     * (tile, b) names the 4 KB block of `rec` that holds records 8b .. 8b+7 of the tile's 64 rays
     * (-1: none).  The forward clears the table and hands blocks out of rec's pool_blocks blocks
     * (a positive multiple of 32: the pool is cut into 32 parts, a tile draws from part tile % 32;
     * pool_next: device int32 [32 * 16], one counter per part, 64 bytes apart, each = blocks handed out - 1
     * (word 1, ABI v18: 1 once a recording kernel has filled a ray's list to its capacity -- what the tail launches
     * read instead of every ray's list length; -1 after the fill);
     * placed right behind blocktab, table and counters are cleared with one fill) the first time a ray
     * of a tile starts block b; a ray that finds
     * the pool used up stops recording there exactly as one that reaches max_samples does (it marches
     * the rest).  rec then needs pool_blocks * 4096 bytes -- the samples that exist, not the cap times
     * the rays -- and max_samples (<= 512) only caps a single ray. */
    void*   blocktab;
    int64_t pool_blocks;
    void*   pool_next;
    int32_t terms_state;   /* see terms */
    int32_t flags;         /* 0 or an OR of SVOXT_LISTS_* (above) */
    void*   tile_state;    /* optional (pooled lists only): device int32 [17 * ceil(Q / 64) + 514], 8-byte aligned (its ready
                              queues hold 64-bit entries; a misaligned pointer is SVOXT_ERR_INVALID), scratch.  Given, the two-kernel
                              forward of 3-channel payloads (N = 2, no view rotations, sigma bitmask at hand) runs its march
                              and its shade as ONE launch: tiles are shaded in the order their marches finish, beside the
                              marches still running (per-tile states, then per-XCD ready queues of 64-bit entries -- tile id and a
                              checksum of the tile's lists -- and their counters: fwd_roles_kernel; NOTEBOOK.md step 33).  Cleared by the
                              forward together with the block table (one fill when it lies right behind pool_next's
                              counters).  Same lists, terms and pixels bit for bit.  NULL: two launches. */
} svoxt_sample_lists;

int svoxt_can_record(const svoxt_tree* tree, const svoxt_options* opt);
/* What svoxt_volume_render_fwd_record does with lists that carry `terms` for this tree / options / lists.flags:
 * 0 = nothing, 2 / 3 = fills them (in the layout of that number: pass it back as terms_state to the backward) */
int svoxt_fwd_fills_terms(const svoxt_tree* tree, const svoxt_options* opt, int32_t list_flags);

/* The sigma bitmask of svoxt_tree.sigma_mask (no counterpart in the reference): bytes for M rows, and the
 * build -- bit (row & 31) of 32-bit word (row >> 5) = features[row * K + K - 1] > sigma_thresh.  One read of a
 * 64-byte line per row (0.05 ms for 4.7 M rows of 32 floats); worth it where the feature table no longer fits
 * the Infinity Cache (see NOTEBOOK.md, step 26). */
int64_t svoxt_sigma_mask_bytes(int64_t M);
int svoxt_sigma_mask_build(const svoxt_tree* tree, float sigma_thresh, void* mask, void* stream);
/* ... and, in the same launch (ABI v18), fill_bytes bytes at `fill` (device, 16-byte aligned, a multiple of 16) set to
 * 0xff: the state a recording forward's pooled lists start from (blocktab, pool_next and tile_state, allocated in one
 * piece).  A caller that builds the mask right before a recording forward hands the lists over with
 * SVOXT_LISTS_BEGUN and saves that forward's own fill launch. */
int svoxt_sigma_mask_build_fill(const svoxt_tree* tree, float sigma_thresh, void* mask, void* fill, int64_t fill_bytes,
                                void* stream);
/* The exponentials table of svoxt_tree.exp_table (ABI v17; no counterpart in the reference) for RGBA-style rows of
 * 8 / 16 / 32 floats: table <- device float [M, K], 16-byte aligned, and -- in the same pass over the feature table --
 * the sigma bitmask (mask as for svoxt_sigma_mask_build, or NULL).  Rebuild after any change to the features. */
int svoxt_exp_table_build(const svoxt_tree* tree, float sigma_thresh, void* mask, float* table, void* stream);
int svoxt_volume_render_fwd_record(const svoxt_tree* tree, const svoxt_rays* rays,
                                   const svoxt_options* opt, float* out,
                                   const svoxt_sample_lists* lists, void* stream);
int svoxt_volume_render_bwd_replay(const svoxt_tree* tree, const svoxt_rays* rays,
                                   const svoxt_options* opt, const float* grad_out,
                                   int32_t grad_cols, float* grad_features, int32_t grad_stride,
                                   const svoxt_sample_lists* lists, const float* fwd_out, void* stream);
/* fwd_out: device [Q, C+1] = the output svoxt_volume_render_fwd_record produced
 * with these lists, or NULL.  NULL: two list walks per ray, every gradient
 * contribution bit-identical to the reference's formulas.  Given: ONE walk --
 * the reference's first pass only computes accum = sum_c g_c * out_c and the
 * final transmittance, both known from the forward; the value of accum then
 * differs from the sequentially accumulated one by float rounding (~1e-7 of the
 * summed magnitudes), which reaches the sigma column only (well inside the 1e-5
 * parity tolerance); colour gradients stay bit-identical. */

/* Copy the first K columns of src [M, stride] into dense dst [M, K] (streaming,
 * non-temporal): turns a strided gradient buffer into the reference's layout. */
int svoxt_compact_rows(const float* src, int64_t M, int32_t K, int32_t stride, float* dst, void* stream);
/* The same copy, and all of src [M, stride] is zero afterwards (SVOXT_LISTS_GRAD_ZEROED above). */
int svoxt_compact_rows_clear(float* src, int64_t M, int32_t K, int32_t stride, float* dst, void* stream);

/* out: device [Q, 1] = accumulated opacity (alpha). */
int svoxt_opacity_render_fwd(const svoxt_tree* tree, const svoxt_rays* rays,
                             const svoxt_options* opt, float* out, void* stream);

/* grad_out: device [Q, 1].  Same kernel family as volume_render_bwd with zero
 * colour channels (the reference launches render_ray_backward_kernel,
 * rt_kernel.cu:1607): only the sigma column of grad_features is non-zero. */
int svoxt_opacity_render_bwd(const svoxt_tree* tree, const svoxt_rays* rays,
                             const svoxt_options* opt, const float* grad_out,
                             float* grad_features, void* stream);

/* The same pair with sample lists (any thresholds since ABI v20: recorded is every sample with sigma > 0,
 * composited what the forward's thresholds leave): the forward records each ray's
 * samples, the backward walks the lists instead of marching twice and adds the
 * contributions up per 8x8 tile before they go to memory (rec and aux are rewritten:
 * the lists serve one backward).  Every contribution is the reference's
 * delta_t * delta_scale * grad_output * T_ray (rt_kernel.cu:486-490 without colour terms). */
int svoxt_opacity_render_fwd_record(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt,
                                    float* out, const svoxt_sample_lists* lists, void* stream);
int svoxt_opacity_render_bwd_replay(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt,
                                    const float* grad_out, float* grad_features, int32_t grad_stride,
                                    const svoxt_sample_lists* lists, void* stream);

/* depth: device [Q, 1] = delta_scale * t of the first sample with
 * sigma > sigma_thresh, 0 if none. */
int svoxt_render_depth(const svoxt_tree* tree, const svoxt_rays* rays,
                       const svoxt_options* opt, float* depth, void* stream);

/* Nearest-leaf point query.
 * points  : device [Q, 3] (world coordinates; pass offset=0, scaling=1 for local)
 * values  : device [Q, K]   feature row of the leaf; zeros for an empty leaf
 *           (the reference leaves such rows uninitialised, svox_kernel.cu:282)
 * node_ids: device [Q] int64 packed leaf id node*N^3 + u*N^2 + v*N + w
 * data_ids: device [Q] int64 feature row index, -1 for an empty leaf
 * hit_mask: device [n_internal * N^3] uint8 or NULL; set to 1 for every leaf
 *           slot hit (caller pre-zeroes).  The unique-leaf list the reference
 *           returns as leaf_node is the sorted set of non-zero entries. */
int svoxt_query_fwd(const svoxt_tree* tree, const float* points, int64_t Q,
                    float* values, int64_t* node_ids, int64_t* data_ids,
                    uint8_t* hit_mask, void* stream);

/* Unique-leaf list of a query (the reference's leaf_node, svox_kernel.cu:240-269,
 * 304-320): compacts hit_mask [n_slots] into leaf_node [U, 4] int64 rows
 * (node, u, v, w) in increasing packed-id order and writes U to *count (device).
 * leaf_node must have room for min(Q, n_slots) rows; workspace:
 * svoxt_query_leaves_workspace_bytes(n_slots) device bytes. */
int64_t svoxt_query_leaves_workspace_bytes(int64_t n_slots);
int svoxt_query_leaves(const uint8_t* hit_mask, int64_t n_slots, int32_t N, int64_t* leaf_node,
                       int64_t* count, void* workspace, void* stream);

/* grad_out: device [Q, K]; grad_features: device [M, K], zeroed by this call. */
int svoxt_query_bwd(const svoxt_tree* tree, const float* points, int64_t Q,
                    const float* grad_out, float* grad_features, void* stream);

/* Roofline counters: marches every ray exactly as
 * volume_render_fwd does and adds to counters[5] (device int64, caller
 * pre-zeroes): rays that enter the cube, leaf crossings S, child words read
 * sum(L), crossings with a valid feature index, composited samples. */
int svoxt_count_fwd(const svoxt_tree* tree, const svoxt_rays* rays,
                    const svoxt_options* opt, int64_t* counters, void* stream);

/* Instrumentation for the roofline (no counterpart in the reference): what ONE forward march of
 * this ray batch touches.  row_mask: device [2 * M] bytes, pre-zeroed -- [idx] = 1 for every valid
 * leaf's feature row (read by the forward), [M + idx] = 1 where a sample is composited (read again
 * by the backward).  tree_mask: device bytes, pre-zeroed -- with tree->accel: [cell] for the grid
 * cells and [n_cells + slot] for the (child, data) pairs read; without: [slot] for child words and
 * [n_slots + slot] for data words (N = 2; other N: data words only).  longest: device int64,
 * pre-zeroed <- the most leaf crossings of any ray.  The march is the production one. */
int svoxt_count_touched(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt,
                        uint8_t* row_mask, uint8_t* tree_mask, int64_t* longest, void* stream);

/* Instrumentation: counters [2] (device int64, caller-zeroed) or NULL.  While set, the one-kernel
 * per-tile backwards (grad_fused_kernel; grad_wide_kernel for rows of 8 / 16 / 32 floats) add to
 * counters[0] the number of 64-byte atomic requests they send to the gradient table and to counters[1]
 * the number of (tile, window of 16 list positions, feature row) groups they belong to.  Process-wide;
 * meant for bench.py. */
int svoxt_set_bwd_counters(int64_t* counters);

/* Instrumentation (ABI v17): words [32] (device int64, caller-zeroed) or NULL.  While set, the one-kernel
 * per-tile backwards run their CHECKED instances: every index into an LDS array, the lists' block pool
 * and the feature / gradient tables is compared with its extent before it is used; a violation adds one
 * to words[2 + site] (sites: svoxt_bwd_kernels.h, grad_fused_kernel / grad_wide_kernel) and the index is
 * clamped, so the checked run cannot itself leave its arrays; words[31] counts the tiles the checked
 * instances worked on (0: the route taken has no checked instance).  Results are those of the production
 * instances.  Process-wide; meant for tests (the memory fault of round 3). */
int svoxt_set_bwd_check(int64_t* words);
/* (ABI v19) Images of trees with more than `rows` feature rows (M) are walked in super-tiles of 8 x 8 tiles instead of
 * row-major (svoxt_rays.image_width); default 2^21 rows; < 0 restores the default.  Process-wide; a forward that
 * records sample lists and the backward that walks them must run under the same value (tests use 0 to exercise the
 * super-tile walk on small, ragged images).  Returns the value in force before the call. */
int64_t svoxt_set_super_tile_rows(int64_t rows);
/* (ABI v20) The walk the rule above gives this tree / ray batch right now: SVOXT_LISTS_WALK_ROWMAJOR or
 * SVOXT_LISTS_WALK_SUPER for a batch declared an image of 8 x 8 tiles, 0 otherwise (-1: NULL argument).  OR it into
 * svoxt_sample_lists.flags of a recording forward and of the backward over the same lists. */
int32_t svoxt_image_walk(const svoxt_tree* tree, const svoxt_rays* rays);

/* ---- One training step, planned by the library (ABI v21; no counterpart in the reference) -------------------------
 *
 * Everything above is mechanism; WHICH mechanisms a forward + backward pair should use for a payload (sample lists
 * pooled or not, the sigma bitmask and the exponentials table, the forward's hand-over to the backward, march and
 * shade as one launch, the per-tile backward, padded gradient rows) is policy.  These three calls are that policy --
 * the route svox_t_amd's VolumeRenderer takes with every switch at its default -- for hosts that do not want to
 * re-implement it (INTEGRATION.md route C).  Memory stays the caller's: one workspace, sized by the plan.
 *
 *   svoxt_step_plan      host only: decides the route for this tree / batch / options and lays out the workspace.
 *                        pool_blocks: 4 KB list blocks to provide (0: a first guess, 12 per 64-ray tile).  After a
 *                        step, (max over i of step.lists.pool_next[16 i] + 1) * 32 is what the batch used (device
 *                        words; read them whenever convenient): at or above lists.pool_blocks the pool ran dry --
 *                        results are unaffected (such rays march their remainder) but slower; plan the next step
 *                        with more.  A plan stays valid while tree->M / K / N / xform, the batch's shape (Q, image
 *                        size, order given or not), the options' format / basis_dim do not change.
 *   svoxt_step_forward   volume_render (rt_kernel.cu:1362-1379): out [Q, C+1]; leaves in the workspace what the
 *                        backward of the SAME features / rays / options replays.  Thresholds: any (the lists hold
 *                        every sample with sigma > 0).  tree->accel is used if given (build it once per topology);
 *                        tree->sigma_mask / exp_table are ignored -- both are built here, per call, in the workspace.
 *   svoxt_step_backward  volume_render_backward (rt_kernel.cu:1402-1426): grad_features dense [M, K], every element
 *                        written (zero fill included).  Exact arithmetic (every contribution the reference's formula).
 * The workspace must be 256-byte aligned and may be a different allocation for every step (the plan holds offsets),
 * but the backward needs the bytes its forward left.  No call synchronises. */
typedef struct svoxt_step {
    svoxt_sample_lists lists;      /* pointers are (re)bound to the workspace by each call */
    int64_t workspace_bytes;       /* device bytes the two calls need */
    int32_t records;               /* 1: the forward records sample lists and the backward replays them; 0: both march */
    int32_t grad_cols, grad_stride;
    int32_t uses_mask, uses_table;
    int64_t off_mask, off_table, off_tables, tables_bytes, off_rec, off_aux, off_terms, terms_bytes, off_grad_rows,
            off_bwd_ws, bwd_ws_bytes, nt;
    /* a payload one step away from a specialised one is rendered AS it, with dummy channels (one or two channels with a
       basis -> three; RGBA-style rows of 2 .. 31 floats of another width -> 4 / 8 / 16 / 32): zeros in front of sigma and in
       the upstream gradient; the caller sees its own shapes, the same bits */
    int32_t pad_K, pad_real, pad_dummy, pad_w;       /* pad_K == 0: not padded */
    int64_t off_pad_features, off_pad_out, off_pad_gout, off_pad_grad;
} svoxt_step;
int svoxt_step_plan(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt, int64_t pool_blocks,
                    svoxt_step* step);
int svoxt_step_forward(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt, float* out,
                       svoxt_step* step, void* workspace, void* stream);
int svoxt_step_backward(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt, const float* grad_out,
                        float* grad_features, svoxt_step* step, void* workspace, void* stream);

/* Acceleration grid (no counterpart in the reference).  A 2^g x 2^g x 2^g table
 * that caches, per cell, where the root->leaf descent of common.cuh:63-100
 * stands after g levels (the leaf and its data word if it ended earlier), so a
 * march step costs one 4-byte load plus the levels below g instead of a
 * dependent load per level.  Behind the cells the buffer holds one (child word,
 * data word) pair per tree slot, so that the levels below g cost one 8-byte load
 * each and the leaf's data word none.  svoxt_accel_bytes returns the buffer size
 * for a resolution and a tree (4 bytes per cell + 64 bytes per internal node), -1
 * if log2_res is outside [1, 8]; svoxt_accel_build fills `cells` (8-byte aligned) from
 * tree->child / tree->data; trees with 2^27 - 1 feature rows or internal nodes and more
 * are refused (SVOXT_ERR_UNSUPPORTED: a cell has 27 index bits) -- render them without a grid. */
/* (ABI v22) log2_res | SVOXT_ACCEL_BRICKS, g >= 2: the cells lie in bricks of 4 x 4 x 4 (two 128-byte lines) instead of
 * row-major, so that a ray's consecutive crossings and a tile's 64 rays share lines whichever way they point.  It pays
 * where a kernel waits for the cell and nothing else -- the marching wavefronts of the one-launch recording forward of
 * 3-channel payloads: 0.248 -> 0.234 ms at 800 x 800 / depth 8 -- and costs the one-kernel forward, short of issue slots,
 * its six extra integer operations per crossing (0.203 -> 0.216 ms): build the grid a training step renders through in
 * bricks, a grid for forward-only rendering of 3-channel payloads row-major.  Rows of 8 / 16 / 32 floats, whose forward
 * is march + shade as two kernels either way: bricks, and a level finer than the tree's size suggests (depth 9, 578 MB
 * of features, 1024 x 1024: g 7 row-major 0.837 ms forward, g 8 in bricks 0.801).  tree->accel_log2 carries the flag to
 * the kernels; same size. */
#define SVOXT_ACCEL_BRICKS 0x100
int64_t svoxt_accel_bytes(int32_t log2_res, int64_t n_internal);
int     svoxt_accel_build(const svoxt_tree* tree, int32_t log2_res, void* cells, void* stream);

/* Ray ordering for batches that are not images (no counterpart in the reference).  The
 * marching kernels put 64 consecutive rays on one wavefront; a batch in arbitrary order
 * (rays drawn at random from many cameras) then diverges at every step.  svoxt_ray_order
 * writes to perm [Q] (device, int32) the permutation that sorts the batch by the Morton
 * code (7 bits per axis) of the point where each ray enters the tree's cube (rays that miss it last;
 * the order of the rays of one cell is not fixed from run to run): gather
 * origins / dirs / vdirs with it, render the sorted batch through any entry point above and
 * scatter the output rows back -- results are per ray and do not depend on the order.
 * workspace: device, svoxt_ray_order_workspace_bytes(Q) bytes (-1: Q out of range or no
 * device).  Batches declared as images (image_width > 0, camera mode) are refused: they are
 * walked in 8x8 tiles already. */
int64_t svoxt_ray_order_workspace_bytes(int64_t Q);
int     svoxt_ray_order(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt,
                        int32_t* perm, void* workspace, int64_t workspace_bytes, void* stream);
/* The gathers and the scatter that go with it: origins / dirs / vdirs [Q, 3] of the sorted batch
 * (row i = row perm[i] of the caller's) in one launch, and rows of `cols` floats either way --
 * scatter = 0: dst[i, :] = src[perm[i], :]; scatter = 1: dst[perm[i], :] = src[i, :]. */
int     svoxt_gather_rays(const svoxt_rays* rays, const int32_t* perm, float* origins, float* dirs, float* vdirs,
                          void* stream);
int     svoxt_permute_rows(const float* src, const int32_t* perm, float* dst, int64_t n, int32_t cols,
                           int32_t scatter, void* stream);

/* Is an undeclared batch a row-major pinhole image (no counterpart in the reference; ABI v22)?  A caller of the
 * reference's API hands over 800 x 800 rays as [640 000, 3] arrays and says nothing; walked in 8 x 8 pixel tiles such a
 * batch needs no sort and its tiles are coherent in direction too.  Two small launches over the first 65 536 rays write
 * result[0 .. 4] (device, int32, room for SVOXT_IMAGE_PROBE_WORDS words: the rest is scratch): [0] 1 if every origin
 * equals the first; [1] how many jumps of the L1 distance between consecutive directions were found (at most 2: a jump
 * is more than four times the median of the first 1 024 distances); [2], [3] their positions; [4] `ticket`, so that a
 * host that reads the words later, from a buffer it reuses, can tell whose answer it holds.  An image of width W has
 * its first jump at W - 1 and its second at 2 W - 1; the caller then sets image_width / image_height (which must be
 * multiples of 8 with W * H = Q).  A hint like every other: results do not depend on what the caller concludes.
 * Q >= 4096; not for camera mode. */
#define SVOXT_IMAGE_PROBE_WORDS 256
int     svoxt_image_probe(const svoxt_rays* rays, int32_t* result, int32_t ticket, void* stream);

/* ---- Octree construction from a point cloud -------------
 *
 * The reference builds a frame's octree by repeating, depth-1 times,
 * `tree[points].refine()` (helpers.py:101-109: a point query, svox_kernel.cu:45-94,
 * its unique-leaf list, :240-324, then N3Tree.refine, svox.py:488-560) on a fresh
 * N = 2 tree, and then `tree.construct_tree(points)` (svox.py:160-161,
 * svox_kernel.cu:110-121: data[leaf of point i] = i).  The two calls below produce
 * the same child / parent_depth / data tables in one pipeline, with one host
 * read in the middle (the node count, to size the tables) instead of one per round:
 * occupancy bitmaps per level instead of repeated descents, integer prefix sums
 * for the node numbering (nodes of a level in increasing packed-parent order,
 * i.e. the order refine() assigns when given the sorted unique-leaf list).
 * Where several points share a finest leaf, the reference keeps whichever thread
 * wrote last; here the smallest point index wins.
 *
 * points          : device [P, 3] fp32, world coordinates
 * offset, scaling : device fp32[3] (as in svoxt_tree)
 * depth           : D in [1, 10]; internal nodes at levels 0..D-1, finest leaf side 2^-D
 * workspace       : svoxt_build_workspace_bytes(depth) device bytes, carried from
 *                   _count to _emit unchanged
 * n_internal      : _count writes the node count to this device int64; _emit takes
 *                   the value the host read back, = rows of the three tables
 * child, data     : device int32 [n_internal, 2, 2, 2]; parent_depth: [n_internal, 2];
 *                   every row is written (no pre-initialisation needed)
 * empty_index     : value for leaf slots that hold no point (any value >= P) */
int64_t svoxt_build_workspace_bytes(int32_t depth);
int svoxt_build_count(const float* points, int64_t P, const float* offset, const float* scaling,
                      int32_t depth, void* workspace, int64_t workspace_bytes,
                      int64_t* n_internal, void* stream);
int svoxt_build_emit(const float* points, int64_t P, const float* offset, const float* scaling,
                     int32_t depth, const void* workspace, int64_t workspace_bytes,
                     int32_t* child, int32_t* data, int32_t* parent_depth,
                     int64_t n_internal, int32_t empty_index, void* stream);

/* N3Tree.refine for an explicit selector (svox.py:520-546; any N): leaf i of
 * leaf_node (device [n_leaves, 4] int64 rows (node, u, v, w), unique, each a leaf
 * slot of a node < filled) becomes internal node filled + i: child[leaf] = offset to
 * it, its N^3 slots inherit the leaf's data word and have child 0, parent_depth =
 * (packed leaf id -- or node_id[i] when node_id is given -- , depth of the leaf's node
 * + 1).  Rows [filled, filled + n_leaves) of the three tables must exist (capacity). */
int svoxt_refine(const int64_t* leaf_node, int64_t n_leaves, int32_t N, int64_t filled, int64_t capacity,
                 int32_t* child, int32_t* data, int32_t* parent_depth, const int32_t* node_id, void* stream);

/* construct_tree (svox.py:160-161, svox_kernel.cu:110-121, 341-352) on an existing
 * tree of any N: data[leaf containing point i] = i; of several points in one
 * leaf the smallest index is kept (the reference: whichever wrote last).
 * tree->data is written; tree->features is not read. */
int svoxt_construct_tree(const svoxt_tree* tree, const float* points, int64_t P, void* stream);

/* ---- Motion variants of the march --------------------------
 *
 * svoxt_motion_render: motion_render (rt_kernel.cu:698-778, 1480-1504).  For each ray,
 * the first sample with sigma > sigma_thresh gives
 *   out       device [Q, J]  distance from the hit point to each of the J = extra_rows
 *                            joint positions extra_data[j][0:3]
 *   depth     device [Q]     t * delta_scale
 *   hit_point device [Q, 3]  (as the reference computes it: transform_coord_world of
 *                            the LEAF-LOCAL coordinates, rt_kernel.cu:745,757)
 *   data_idx  device [Q] int64  feature row of the hit leaf
 * and zeros in all four when nothing is hit. */
int svoxt_motion_render(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt,
                        float* out, float* depth, float* hit_point, int64_t* data_idx, void* stream);

/* TreeSpec.joint_features / skinning_weights / joint_index (data_spec.hpp:66-110) */
typedef struct svoxt_motion {
    const float*   joint_features;    /* device [n_joints, feature_dim] */
    int32_t        n_joints;
    int32_t        feature_dim;       /* 1..32 (the reference's fixed tmp_data_dim, rt_kernel.cu:904) */
    const float*   skinning_weights;  /* device [M, n_bind]: weight of each bound joint per feature row */
    const int32_t* joint_index;       /* device [M, n_bind]: which joints; entries outside [0, n_joints) are skipped */
    int32_t        n_bind;
} svoxt_motion;

/* motion_feature_render (rt_kernel.cu:886-981, 1525-1543): out device [Q, feature_dim] =
 * sum_samples weight * sigmoid(sum_j skinning_weight_j * joint_features[joint_index_j]) + T * background
 * (zeros for a ray that misses the cube, :913-919); early stop as volume_render.
 * workspace: svoxt_motion_workspace_bytes(M, feature_dim) device bytes, 16-byte aligned
 * (the per-row blend of the joint features, evaluated once per call; for the backward
 * also the gradient wrt it).  -1 if feature_dim is outside [1, 32]. */
int64_t svoxt_motion_workspace_bytes(int64_t M, int32_t feature_dim);
int svoxt_motion_feature_render_fwd(const svoxt_tree* tree, const svoxt_motion* motion, const svoxt_rays* rays,
                                    const svoxt_options* opt, float* out, void* workspace,
                                    int64_t workspace_bytes, void* stream);

/* Gradient of the above wrt joint_features: grad_joint_features device
 * [n_joints, feature_dim], zeroed by this call.  This is the derivative of the forward
 * (thresholds ignored, as in every backward of the reference); the reference's own
 * motion_feature_render_backward (rt_kernel.cu:983-1061) accumulates into an
 * uninitialised local indexed by bone, so it defines no result to reproduce. */
int svoxt_motion_feature_render_bwd(const svoxt_tree* tree, const svoxt_motion* motion, const svoxt_rays* rays,
                                    const svoxt_options* opt, const float* grad_out,
                                    float* grad_joint_features, void* workspace, int64_t workspace_bytes,
                                    void* stream);

/* ---- Linear blend skinning of points: the producer of transformation_matrices ------
 *
 * warp_vertices (svox_kernel.cu:123-154, 354-378; svox.py:58-76, 974-981):
 *   matrix_out[q]   device [Q, 4, 4] = sum_j (w_qj > 0) w_qj * matrices[joint_qj] on rows 0..2,
 *                   row 3 = (0, 0, 0, 1); 16-byte aligned
 *   vertices_out[q] device [Q, 3]    = matrix_out[q][0:3, 0:3] * points[q] + matrix_out[q][0:3, 3]
 * matrices: device [n_joints, 4, 4]; skinning_weights [Q, n_bind]; joint_index [Q, n_bind] int32
 * (entries outside [0, n_joints) are skipped).  matrix_out is what volume_render takes as
 * svoxt_tree.xform with xform_dim = 4 when rows of `features` are points. */
int svoxt_warp_vertices(const float* matrices, int32_t n_joints, const float* points, int64_t Q,
                        const float* skinning_weights, const int32_t* joint_index, int32_t n_bind,
                        float* vertices_out, float* matrix_out, void* stream);

/* warp_vertices_backward (svox_kernel.cu:156-211, 404-436): from grad_vertices [Q, 3] and
 * grad_matrix_out [Q, 4, 4] to grad_points [Q, 3], grad_matrices [n_joints, 4, 4] (zeroed by
 * this call, float atomics) and grad_skinning_weights [Q, n_bind] (every entry written). */
int svoxt_warp_vertices_bwd(const float* matrices, int32_t n_joints, const float* points, int64_t Q,
                            const float* skinning_weights, const int32_t* joint_index, int32_t n_bind,
                            const float* grad_vertices, const float* grad_matrix_out,
                            float* grad_points, float* grad_matrices, float* grad_skinning_weights,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SVOXT_H_ */
