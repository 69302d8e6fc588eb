// svoxt_bwd.hip -- volume_render_backward (trace_ray_backward, rt_kernel.cu:331-496, 675-694, 1402-1426) and
// opacity_render_backward (:1593-1616) behind the C ABI: which kernel of svoxt_bwd_kernels.h serves a payload / list
// combination.  A translation unit of its own so that the two halves of the library's kernels compile side by side
// (svoxt_kernels.hip keeps the forward, the queries and the utilities).  DESIGN.md 4 (the measurements behind it: NOTEBOOK.md 5).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see build.py).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "svoxt_launch.h"

#pragma clang fp contract(off)

#include "svoxt_bwd_kernels.h"

using namespace svoxt;

namespace {
int64_t* g_bwd_counters = nullptr;      // svoxt_set_bwd_counters (instrumentation)
int64_t* g_bwd_check = nullptr;         // svoxt_set_bwd_check (instrumentation): the per-tile backwards run their CHECK instances

// specialised kernels with per-leaf view rotations: SH payloads on N = 2 trees
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

bool launch_lobes_bwd_tiles(const TreeDev& tr, const RaysDev& rays, const Opts& opt, const float* grad_out, float* grad,
                            int gstride, RecLists L, const uint4* aux, hipStream_t st, int terms_state) {
    if (L.terms == nullptr || terms_state != 3 || !lobes_payload(opt, tr.K) || g_bwd_counters != nullptr) return false;
    const unsigned nb = nblocks(rays.Q);
#define SVOXT_LOBES_BWD(BB)                                                                                         \
    {                                                                                                               \
        hipLaunchKernelGGL((render_bwd_kernel<FMT_SH, 3, BB, true, true, false, true, false, true>), dim3(nb), dim3(kBlock), \
                           0, st, tr, rays, opt, grad_out, grad, gstride, L, aux, (const float*)nullptr, (float4*)nullptr); \
        hipLaunchKernelGGL((grad_fused_kernel<FMT_SH, BB, true, false, 3, true>), dim3(nb), dim3(512), 0, st,       \
                           tr, rays, opt, grad_out, L, aux, (const float*)nullptr, grad, gstride);                  \
        return true;                                                                                                \
    }
    switch (opt.basis_dim) {
        case 1: SVOXT_LOBES_BWD(1)
        case 4: SVOXT_LOBES_BWD(4)
        case 9: SVOXT_LOBES_BWD(9)
        case 16: SVOXT_LOBES_BWD(16)
        case 25: SVOXT_LOBES_BWD(25)
    }
#undef SVOXT_LOBES_BWD
    return false;
}

// two-kernel backward: SH 1/4/9 (also with view rotations) and RGBA with 3 channels
// (K <= 32) on N = 2 trees
bool launch_bwd_gather(const TreeDev& tr, const RaysDev& rays, const Opts& opt, int C,
                       const float* grad_out, float* grad, int gstride, RecLists L, const uint4* aux,
                       const float* fwd_out, float4* coef, bool xf, hipStream_t st, int terms_state = 0,
                       bool native = false) {
    const unsigned nb = nblocks(rays.Q);
    if (C > 3) {
        // RGBA-style rows of 8 / 16 / 32 floats: the exact per-tile form only (one kernel, a float per
        // list slot in L.terms); tails of overflowed rays first, as for the 3-channel fused kernel
        if (opt.format != FMT_RGBA || coef != nullptr || xf || fwd_out != nullptr || L.terms == nullptr) return false;
#define SVOXT_WIDE(KK)                                                                                        \
    {                                                                                                         \
        hipLaunchKernelGGL((render_bwd_kernel<FMT_RGBA, KK - 1, 0, true, true, false, true>), dim3(nb), dim3(kBlock), 0, st, \
                           tr, rays, opt, grad_out, grad, gstride, L, aux, (const float*)nullptr, (float4*)nullptr); \
        unsigned long long* ctr = reinterpret_cast<unsigned long long*>(g_bwd_counters);                      \
        unsigned long long* chkw = reinterpret_cast<unsigned long long*>(g_bwd_check);                        \
        if (chkw != nullptr && !native)                                                                       \
            hipLaunchKernelGGL((grad_wide_kernel<KK, false, false, true>), dim3(nb), dim3(512), 0, st, tr, rays, opt, grad_out, L, aux, \
                               grad, gstride, chkw);                                                          \
        else if (chkw != nullptr)                                                                             \
            return false;                                                                                     \
        else if (ctr != nullptr)                                                                              \
            hipLaunchKernelGGL((grad_wide_kernel<KK, false, true>), dim3(nb), dim3(512), 0, st, tr, rays, opt, grad_out, L, aux, \
                               grad, gstride, ctr);                                                           \
        else if (native)                                                                                      \
            hipLaunchKernelGGL((grad_wide_kernel<KK, true>), dim3(nb), dim3(512), 0, st, tr, rays, opt, grad_out, L, aux, \
                               grad, gstride, (unsigned long long*)nullptr);                                  \
        else if (tr.etab != nullptr)                                                                          \
            hipLaunchKernelGGL((grad_wide_kernel<KK, false, false, false, true>), dim3(nb), dim3(512), 0, st, tr, rays, opt, grad_out, L, aux, \
                               grad, gstride, (unsigned long long*)nullptr);                                  \
        else                                                                                                  \
            hipLaunchKernelGGL((grad_wide_kernel<KK, false>), dim3(nb), dim3(512), 0, st, tr, rays, opt, grad_out, L, aux, \
                               grad, gstride, (unsigned long long*)nullptr);                                  \
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
    const bool fused = coef == nullptr;
    if (fused && xf && (fwd_out != nullptr || opt.format != FMT_SH || opt.basis_dim > 9)) return false;   // (the exact one-kernel form only)
    // four wavefronts per tile and tables of 1024 (measured: one wavefront per tile 0.41 ms,
    // two 0.33, four 0.30 before step 15; tables of 512 / 256 cost more passes than they buy)
#define SVOXT_GATHER(F, BB)                                                                                   \
    if (fused) {   /* tails of overflowed rays (a tail-only launch), then list walk and merge in one kernel */ \
        hipLaunchKernelGGL((render_bwd_kernel<F, 3, BB, true, true, false, true>), dim3(nb), dim3(kBlock), 0, st, \
                           tr, rays, opt, grad_out, grad, gstride, L, aux, fwd_out, (float4*)nullptr);   \
        unsigned long long* ctr = reinterpret_cast<unsigned long long*>(g_bwd_counters);                      \
        unsigned long long* chkw = reinterpret_cast<unsigned long long*>(g_bwd_check);                        \
        if (chkw != nullptr) {   /* the checked instance of the route that would run: the default route (the forward's      \
                                    position-major hand-over) for every payload, the other routes for SH9 */  \
            if (L.terms != nullptr && terms_state == 3 && fwd_out == nullptr)                                 \
                hipLaunchKernelGGL((grad_fused_kernel<F, BB, true, false, 3, false, true>), dim3(nb), dim3(512), 0, st, \
                                   tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride, chkw);            \
            else if constexpr (F == FMT_SH && BB == 9) {                                                      \
                if (fwd_out != nullptr)                                                                       \
                    hipLaunchKernelGGL((grad_fused_kernel<F, BB, false, false, 0, false, true>), dim3(nb), dim3(512), 0, st, \
                                       tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride, chkw);        \
                else if (L.terms != nullptr && terms_state == 2)                                              \
                    hipLaunchKernelGGL((grad_fused_kernel<F, BB, true, false, 2, false, true>), dim3(nb), dim3(512), 0, st, \
                                       tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride, chkw);        \
                else                                                                                          \
                    hipLaunchKernelGGL((grad_fused_kernel<F, BB, true, false, 0, false, true>), dim3(nb), dim3(512), 0, st, \
                                       tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride, chkw); \
            } else return false;                                                                              \
            return true;                                                                                      \
        }                                                                                                     \
        if (fwd_out != nullptr && ctr == nullptr)                                                             \
            hipLaunchKernelGGL((grad_fused_kernel<F, BB, false>), dim3(nb), dim3(512), 0, st,                 \
                               tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride);                 \
        else if (ctr == nullptr && L.terms != nullptr && terms_state == 2)                                    \
            hipLaunchKernelGGL((grad_fused_kernel<F, BB, true, false, 2>), dim3(nb), dim3(512), 0, st,        \
                               tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride);                      \
        else if (ctr == nullptr && L.terms != nullptr && terms_state == 3)                                    \
            hipLaunchKernelGGL((grad_fused_kernel<F, BB, true, false, 3>), dim3(nb), dim3(512), 0, st,        \
                               tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride);                      \
        else if (ctr == nullptr)   /* (no hand-over from the forward: both sweeps gather the rows) */         \
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
    // (r04) view rotations as ONE kernel: the rays whose list overflowed whole by the per-ray kernel, every other ray by
    // grad_fused_kernel<..., XF> (no checked / counting instance: svoxt_set_bwd_check and _counters do not see this route)
#define SVOXT_FUSED_XF(BB)                                                                                        \
    {                                                                                                             \
        hipLaunchKernelGGL((render_bwd_kernel<FMT_SH, 3, BB, true, true, true>), dim3(nb), dim3(kBlock), 0, st,   \
                           tr, rays, opt, grad_out, grad, gstride, L, aux, fwd_out, reinterpret_cast<float4*>(kOnlyOverflowed)); \
        if (L.terms != nullptr && terms_state == 3)   /* the forward's hand-over: exponentials of each record's own basis */ \
            hipLaunchKernelGGL((grad_fused_kernel<FMT_SH, BB, true, false, 3, false, false, true>), dim3(nb), dim3(512), 0, st, \
                               tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride);                          \
        else                                                                                                      \
            hipLaunchKernelGGL((grad_fused_kernel<FMT_SH, BB, true, false, 0, false, false, true>), dim3(nb), dim3(512), 0, st, \
                               tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride);                          \
        return true;                                                                                              \
    }
    if (xf) {
        if (opt.format != FMT_SH) return false;
        if (fused) {
            switch (opt.basis_dim) {
                case 1: SVOXT_FUSED_XF(1)
                case 4: SVOXT_FUSED_XF(4)
                case 9: SVOXT_FUSED_XF(9)
            }
            return false;
        }
        switch (opt.basis_dim) {
            case 1: SVOXT_GATHER_XF(1)
            case 4: SVOXT_GATHER_XF(4)
            case 9: SVOXT_GATHER_XF(9)
        }
        return false;
    }
#undef SVOXT_FUSED_XF
    if (opt.format == FMT_RGBA) { SVOXT_GATHER(FMT_RGBA, 0) }
    // SH16 / SH25 (rows of 49 / 76 floats, r03): the one-kernel per-tile form only, and only over the hand-over a
    // recording forward left (terms_state 2 / 3): the kernel then never holds a feature row
#define SVOXT_GATHER_WIDE(BB)                                                                                 \
    {                                                                                                         \
        if (!fused || g_bwd_counters != nullptr || fwd_out != nullptr || L.terms == nullptr ||                \
            (terms_state != 2 && terms_state != 3)) return false;                                             \
        hipLaunchKernelGGL((render_bwd_kernel<FMT_SH, 3, BB, true, true, false, true>), dim3(nb), dim3(kBlock), 0, st, \
                           tr, rays, opt, grad_out, grad, gstride, L, aux, fwd_out, (float4*)nullptr);        \
        unsigned long long* chkw = reinterpret_cast<unsigned long long*>(g_bwd_check);                        \
        if (chkw != nullptr && terms_state == 2)                                                              \
            return false;                                                                                     \
        else if (chkw != nullptr)                                                                             \
            hipLaunchKernelGGL((grad_fused_kernel<FMT_SH, BB, true, false, 3, false, true>), dim3(nb), dim3(512), 0, st, \
                               tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride, chkw);                \
        else if (terms_state == 2)                                                                            \
            hipLaunchKernelGGL((grad_fused_kernel<FMT_SH, BB, true, false, 2>), dim3(nb), dim3(512), 0, st,   \
                               tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride);                      \
        else                                                                                                  \
            hipLaunchKernelGGL((grad_fused_kernel<FMT_SH, BB, true, false, 3>), dim3(nb), dim3(512), 0, st,   \
                               tr, rays, opt, grad_out, L, aux, fwd_out, grad, gstride);                      \
        return true;                                                                                          \
    }
    if (opt.format == FMT_SH) {
        switch (opt.basis_dim) {
            case 1: SVOXT_GATHER(FMT_SH, 1)
            case 4: SVOXT_GATHER(FMT_SH, 4)
            case 9: SVOXT_GATHER(FMT_SH, 9)
            case 16: SVOXT_GATHER_WIDE(16)
            case 25: SVOXT_GATHER_WIDE(25)
        }
    }
#undef SVOXT_GATHER_WIDE
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
    } else if constexpr (!REPLAY) {
#define SVOXT_BWD_LOBES(BB)                                                                                        \
    hipLaunchKernelGGL((render_bwd_kernel<FMT_SH, 3, BB, N2, false, false, false, false, true>), dim3(nb), dim3(kBlock), \
                       0, st, tr, rays, opt, grad_out, grad, gstride, L, aux, fwd_out);                            \
    return true;
        if ((opt.format == FMT_SG || opt.format == FMT_ASG) && C == 3 && tr.K == 3 * opt.basis_dim + 1) {
            switch (opt.basis_dim) {
                case 1: SVOXT_BWD_LOBES(1)
                case 4: SVOXT_BWD_LOBES(4)
                case 9: SVOXT_BWD_LOBES(9)
                case 16: SVOXT_BWD_LOBES(16)
                case 25: SVOXT_BWD_LOBES(25)
            }
        }
#undef SVOXT_BWD_LOBES
    }
#undef SVOXT_BWD
#undef SVOXT_BWD1
    return false;
}

}  // namespace

namespace svoxt {

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
    // (SVOXT_LISTS_GRAD_ZEROED: the caller's buffer is the scratch svoxt_compact_rows_clear left zeroed)
    if (tree->M > 0 && !(lists != nullptr && (lists->flags & SVOXT_LISTS_GRAD_ZEROED))) {
        const hipError_t e = hipMemsetAsync(grad_features, 0, sizeof(float) * (size_t)tree->M * gs, st);
        if (e != hipSuccess) return fail(SVOXT_ERR_HIP, "%s: hipMemsetAsync: %s", fn, hipGetErrorString(e));
    }
    if (rays->Q == 0 || tree->M == 0) return SVOXT_OK;
    const TreeDev tr = to_dev(tree);
    const RaysDev rd = to_dev(rays, tree, lists);
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
            else if (lists->coef == nullptr && lists->coef_bytes < 0 && tree->K <= 32 && fwd_out == nullptr && C == 3 && n2)
                done = launch_bwd_gather(tr, rd, od, C, grad_out, grad_features, gs, lists_dev(lists, rays->Q),
                                         reinterpret_cast<const uint4*>(lists->aux), fwd_out, nullptr, true, st,
                                         lists->terms_state == 3 ? 3 : 0);     // list walk + merge as one kernel (over the forward's hand-over, if it left one)
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
            // hand-over per list slot: 16 bytes for 3-channel payloads; RGBA rows of 8 / 16 / 32 floats: 8 for the
            // per-tile kernel (grad_wide_kernel), 4 for the per-ray one (render_bwd_kernel<ONEPASS>)
            const bool wide = opt->format == SVOXT_FORMAT_RGBA && C > 3;
            const RecLists ll = lists_dev(lists, rays->Q, wide ? 8 : 16);
            const RecLists l1 = wide ? lists_dev(lists, rays->Q, 4) : ll;
            const uint4* laux = reinterpret_cast<const uint4*>(lists->aux);
            // coef_bytes < 0 (and no coef): the per-tile route if it can run fused, which needs no buffer
            if (lobes_payload(od, tree->K)) {
                // SG / ASG: the one-kernel per-tile backward over the forward's hand-over, or nothing
                if (n2 && lists->coef_bytes < 0)
                    done = launch_lobes_bwd_tiles(tr, rd, od, grad_out, grad_features, gs, ll, laux, st, lists->terms_state);
                if (!done) return fail(SVOXT_ERR_UNSUPPORTED, "%s: SG / ASG sample lists serve the per-tile backward only (N = 2, lists.terms filled by the forward: terms_state 3, coef_bytes < 0)", fn);
                return check_launch(fn);
            }
            const bool have_coef = lists->coef != nullptr &&
                                   lists->coef_bytes >= (int64_t)lists->max_samples * rays->Q * 16;
            // (rows wider than 32 floats: SH16 / SH25 over the forward's hand-over, one kernel -- launch_bwd_gather decides)
            if (n2 && (tree->K <= 32 || !have_coef) && (have_coef || lists->coef_bytes < 0))
                // (ll.terms: the exact one-kernel form's hand-over buffer; terms_state 2 = the forward filled it)
                done = launch_bwd_gather(tr, rd, od, C, grad_out, grad_features, gs, ll, laux,
                                         fwd_out, have_coef ? reinterpret_cast<float4*>(lists->coef) : nullptr, false, st,
                                         (lists->terms_state == 2 || lists->terms_state == 3) ? lists->terms_state : 1,
                                         (lists->flags & SVOXT_LISTS_NATIVE_MATH) != 0);
            if (!done) done = n2 ? launch_bwd_special<true, true>(tr, rd, od, C, grad_out, grad_features, gs, l1, laux, fwd_out, st)
                      : launch_bwd_special<false, true>(tr, rd, od, C, grad_out, grad_features, gs, l1, laux, fwd_out, st);
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

}  // namespace svoxt

extern "C" {

int svoxt_set_bwd_counters(int64_t* counters) {
    g_bwd_counters = counters;
    return SVOXT_OK;
}

int svoxt_set_bwd_check(int64_t* words) {
    g_bwd_check = words;
    return SVOXT_OK;
}

}  // extern "C"
