// svoxt_launch.h -- host-side helpers shared by the translation units that launch the render kernels
// (svoxt_kernels.hip: forward, queries, utilities; svoxt_bwd.hip: the backward): the kernels' view of the caller's
// sample lists, payload predicates, and the backward's entry behind the C ABI.  Not part of the public C ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svoxt.h"
#include "svoxt_device.h"
#include "svoxt_host.h"
#include "svoxt_lists.h"

namespace svoxt {

inline int fail(int code, const char* fmt, const char* a = "", const char* b = "") { return set_error(code, fmt, a, b); }

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
    // term_bytes (16; 8 / 4 for the backwards of wide rows) per record slot, 16-byte aligned, or not at all
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
inline bool uses_xform(const svoxt_tree* t, const svoxt_options* o) {
    return t->xform != nullptr && o->format != SVOXT_FORMAT_RGBA;
}

inline bool full_comp(const svoxt_options* o) {
    return o->format == SVOXT_FORMAT_RGBA || (o->min_comp == 0 && o->max_comp == o->basis_dim - 1);
}

// can the specialised kernels serve this tree / options pair with its view rotations?
inline bool xform_special(const svoxt_tree* t, const svoxt_options* o) {
    return t->N == 2 && o->format == SVOXT_FORMAT_SH && t->K == 3 * o->basis_dim + 1 &&
           (o->basis_dim == 1 || o->basis_dim == 4 || o->basis_dim == 9 || o->basis_dim == 16 || o->basis_dim == 25);
}

// SG / ASG payloads with an SH-sized lobe count and three channels (r03): a ray's basis values are formed once
// (precalc_lobes) and used like an SH basis by the <LOBES> instances of the FMT_SH kernels.
inline bool lobes_payload(const Opts& opt, int K) {
    return (opt.format == FMT_SG || opt.format == FMT_ASG) && K == 3 * opt.basis_dim + 1 &&
           (opt.basis_dim == 1 || opt.basis_dim == 4 || opt.basis_dim == 9 || opt.basis_dim == 16 || opt.basis_dim == 25);
}

// volume_render_backward / opacity_render_backward behind the C ABI's entry points (svoxt_bwd.hip)
int bwd_common(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt,
               const float* grad_out, int32_t grad_cols, float* grad_features, int32_t grad_stride,
               void* workspace, int64_t workspace_bytes, const svoxt_sample_lists* lists,
               const float* fwd_out, void* stream, const char* fn);

}  // namespace svoxt
