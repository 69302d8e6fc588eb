// svoxt_step.hip -- one training step (recording forward + replaying backward) planned by the LIBRARY (ABI v21).
//
// The entry points of include/svoxt.h are mechanisms: sample lists, the sigma bitmask, the exponentials table, the
// hand-over between forward and backward, padded gradient rows, which backward arrangement serves which payload.
// Which of them a step should use was, until r05, written down once -- in the Python operator layer
// (svox_t_amd/csrc/__init__.py) -- so a plain C / C++ host (INTEGRATION.md route C) got the fast path only by
// re-implementing that policy.  The three calls below ARE that policy, over the same entry points, with every byte
// still owned by the caller: svoxt_step_plan lays out ONE workspace, svoxt_step_forward / _backward run the default
// route for the payload (what VolumeRenderer.forward + backward take with every switch at its default).
// Host code only, but for the two column-copy kernels of the padded payloads (below).
#include <cstring>
#include <hip/hip_runtime.h>

#include "svoxt_host.h"
#include "svoxt_lists.h"            // roles_state_words (the size of svoxt_sample_lists.tile_state)

using namespace svoxt;

namespace {

constexpr int64_t kAlign = 256;
inline int64_t up(int64_t x) { return (x + kAlign - 1) / kAlign * kAlign; }
inline bool wide_rows(const svoxt_tree* t, const svoxt_options* o) {
    return o->format == SVOXT_FORMAT_RGBA && (t->K == 8 || t->K == 16 || t->K == 32);
}
inline bool sh_like(const svoxt_options* o) {
    return o->format == SVOXT_FORMAT_SH || o->format == SVOXT_FORMAT_SG || o->format == SVOXT_FORMAT_ASG;
}
inline bool is_tiled(const svoxt_rays* r) {
    return r->image_width > 0 && r->image_height > 0 && r->image_width % 8 == 0 && r->image_height % 8 == 0 &&
           (int64_t)r->image_width * r->image_height == r->Q;
}
// per-leaf view rotations whose recording forward runs march and shade as one launch (SH 1 / 4 / 9)
inline bool xf_roles(const svoxt_tree* t, const svoxt_options* o) {
    return t->xform != nullptr && t->N == 2 && t->weight_accum == nullptr && o->format == SVOXT_FORMAT_SH &&
           (o->basis_dim == 1 || o->basis_dim == 4 || o->basis_dim == 9) && t->K == 3 * o->basis_dim + 1;
}
inline char* at(void* ws, int64_t off) { return static_cast<char*>(ws) + off; }

// Payloads one step away from a specialised one are rendered AS it, with dummy channels (what svox_t_amd/csrc/__init__.py
// does under PAD_PAYLOADS): one or two channels with a basis -> three, RGBA-style rows of 2 .. 31 floats of
// another width -> 4 / 8 / 16 / 32.  Columns of zeros go in front of the last column (sigma / alpha / the sigma gradient).
struct PadLayout { int Kp, real, dummy, w; };
inline bool pad_layout(const svoxt_tree* t, const svoxt_options* o, PadLayout* p) {
    const int K = t->K, bd = o->basis_dim;
    if (o->format == SVOXT_FORMAT_RGBA) {
        if (K < 2 || K > 32 || K == 4 || K == 8 || K == 16 || K == 32) return false;
        const int Kp = K < 4 ? 4 : K < 8 ? 8 : K < 16 ? 16 : 32;
        *p = PadLayout{Kp, K - 1, Kp - K, 1};
        return true;
    }
    if (!(bd == 1 || bd == 4 || bd == 9 || bd == 16 || bd == 25) || (K - 1) % bd != 0 || o->min_comp != 0 || o->max_comp != bd - 1) return false;
    const int C = (K - 1) / bd;
    if (C < 1 || C >= 3) return false;
    *p = PadLayout{3 * bd + 1, C * bd, (3 - C) * bd, bd};
    return true;
}

// dst [n, dc] <- src [n, sc]: the first `real` columns and the last one; zeros between them where dc > sc, the columns between
// dropped where dc < sc
__global__ void __launch_bounds__(256)
copy_cols_kernel(const float* __restrict__ src, int sc, float* __restrict__ dst, int dc, int64_t n, int real) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * dc) return;
    const int64_t row = i / dc;
    const int c = (int)(i - row * dc);
    float v;
    if (c < real) v = src[row * sc + c];
    else if (c == dc - 1) v = src[row * sc + sc - 1];
    else v = 0.f;                                    // (padding only: a dropped column is never a destination)
    dst[i] = v;
}
inline int copy_cols(bool /* pad */, const float* src, int sc, float* dst, int dc, int64_t n, int real, void* stream, const char* fn) {
    if (n == 0) return SVOXT_OK;
    const unsigned nb = (unsigned)((n * dc + 255) / 256);
    hipLaunchKernelGGL(copy_cols_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, src, sc, dst, dc, n, real);
    return check_launch(fn);
}

}  // namespace

extern "C" {

int svoxt_step_plan(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt, int64_t pool_blocks,
                    svoxt_step* step) {
    const char* fn = "svoxt_step_plan";
    int rc;
    if (step == nullptr) return set_error(SVOXT_ERR_INVALID, "%s: step is NULL", fn);
    if ((rc = check_tree(tree, fn)) || (rc = check_rays(rays, fn)) || (rc = check_opts(opt, tree, fn, true))) return rc;
    std::memset(step, 0, sizeof(*step));
    svoxt_tree padded_tree;
    PadLayout pl;
    const int64_t K_caller = tree->K;
    const int cols_caller = svoxt_out_data_dim(opt, tree->K);
    if (cols_caller < 2) return set_error(SVOXT_ERR_INVALID, "%s: bad output width", fn);
    if (pad_layout(tree, opt, &pl)) {                      // planned as the specialised payload it is rendered as
        padded_tree = *tree;
        padded_tree.K = pl.Kp;
        tree = &padded_tree;
        step->pad_K = pl.Kp; step->pad_real = pl.real; step->pad_dummy = pl.dummy; step->pad_w = pl.w;
    }
    const int64_t Q = rays->Q, M = tree->M, K = tree->K;
    const int cols = svoxt_out_data_dim(opt, tree->K);
    const bool coherent = is_tiled(rays) || rays->order != nullptr || rays->c2w != nullptr;
    const bool wide = wide_rows(tree, opt);
    int rec = Q > 0 && M > 0 ? svoxt_can_record(tree, opt) : 0;
    if (rec == 2 && !coherent) rec = 0;            // SG / ASG lists serve the per-tile backward alone
    step->records = rec != 0;
    step->grad_cols = cols;
    step->grad_stride = (K <= 8 || K % 16 == 0) ? (int32_t)K : (int32_t)((K + 15) / 16 * 16);
    int64_t off = 0;
    auto take = [&](int64_t bytes) { const int64_t o = off; off += up(bytes > 0 ? bytes : 1); return o; };
    step->off_pad_features = step->off_pad_out = step->off_pad_gout = step->off_pad_grad = -1;
    if (step->pad_K) {
        step->off_pad_features = take(4 * M * K);
        step->off_pad_out = take(4 * Q * cols);
        step->off_pad_gout = take(4 * Q * cols);
        step->off_pad_grad = take(4 * M * K);
    }
    (void)K_caller; (void)cols_caller;
    if (step->grad_stride != K) step->off_grad_rows = take(4 * M * step->grad_stride); else step->off_grad_rows = -1;
    if (!step->records) {
        // no lists for this payload / options: both calls march; the backward gets its own scratch lists
        step->bwd_ws_bytes = svoxt_bwd_workspace_bytes(Q, 96);
        step->off_bwd_ws = step->bwd_ws_bytes > 0 ? take(step->bwd_ws_bytes) : -1;
        step->workspace_bytes = off > 0 ? off : kAlign;
        return SVOXT_OK;
    }
    const int S = tree->xform == nullptr ? 192 : 96;                  // pooled lists make a high cap free; view rotations: see _list_cap
    const int64_t tiles = (Q + 63) / 64, full = tiles * (S / 8);
    int64_t blocks = pool_blocks > 0 ? pool_blocks : (full < tiles * 12 ? full : tiles * 12);
    if (blocks > full) blocks = full;
    blocks = (blocks + 31) / 32 * 32;
    const bool roles_xf = xf_roles(tree, opt);
    int32_t lflags = roles_xf ? SVOXT_LISTS_FWD_TWO_KERNELS : 0;
    const int fills = svoxt_fwd_fills_terms(tree, opt, lflags);
    step->uses_mask = (wide || fills == 3 || roles_xf) ? 1 : 0;
    step->uses_table = wide ? 1 : 0;
    const int64_t nt = (full + 1) / 2 * 2;                            // (even: the 64-bit queue entries of tile_state stay aligned)
    const int64_t table_words = (nt + 32 * 16 + roles_state_words(tiles) + 3) / 4 * 4;
    step->off_mask = step->uses_mask ? take(svoxt_sigma_mask_bytes(M)) : -1;
    step->off_table = step->uses_table ? take(4 * M * K) : -1;
    step->off_tables = take(4 * table_words);
    step->tables_bytes = 4 * table_words;
    step->off_rec = take(blocks * 4096);
    step->off_aux = take(((Q + 63) / 64 * 64) * 16);
    // (att, e0, e1, e2) per record slot where the forward leaves the backward's hand-over; for rows of 8 / 16 / 32
    // floats the scratch between the two sweeps of their per-tile backward (8 bytes per slot)
    step->terms_bytes = fills ? blocks * 512 * 16 : (wide ? blocks * 512 * 8 : 0);
    step->off_terms = step->terms_bytes > 0 ? take(step->terms_bytes) : -1;
    step->workspace_bytes = off;
    svoxt_sample_lists& l = step->lists;
    l.max_samples = S;
    l.coef = nullptr; l.coef_bytes = 0;
    l.pool_blocks = blocks;
    l.terms_state = fills;
    l.terms_bytes = fills ? step->terms_bytes : 0;
    l.flags = lflags | (svoxt_image_walk(tree, rays) > 0 ? svoxt_image_walk(tree, rays) : 0);
    step->nt = nt;
    return SVOXT_OK;
}

// the lists' pointers for this workspace (the plan holds offsets: a workspace may move between steps)
static void bind_lists(svoxt_step* s, void* ws) {
    svoxt_sample_lists& l = s->lists;
    int32_t* tables = reinterpret_cast<int32_t*>(at(ws, s->off_tables));
    l.blocktab = tables;
    l.pool_next = tables + s->nt;
    l.tile_state = tables + s->nt + 32 * 16;
    l.rec = at(ws, s->off_rec);
    l.aux = at(ws, s->off_aux);
    l.terms = s->off_terms >= 0 ? at(ws, s->off_terms) : nullptr;
}

int svoxt_step_forward(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt, float* out,
                       svoxt_step* step, void* workspace, void* stream) {
    const char* fn = "svoxt_step_forward";
    if (step == nullptr || workspace == nullptr || tree == nullptr) return set_error(SVOXT_ERR_INVALID, "%s: NULL argument", fn);
    if (((uintptr_t)workspace & 255u) != 0) return set_error(SVOXT_ERR_INVALID, "%s: workspace must be 256-byte aligned", fn);
    int rc;
    svoxt_tree padded_tree;
    float* out_caller = out;
    const int cols_caller = opt != nullptr ? svoxt_out_data_dim(opt, tree->K) : 0;
    if (step->pad_K) {
        if (rays == nullptr || opt == nullptr) return set_error(SVOXT_ERR_INVALID, "%s: NULL argument", fn);
        float* fp = reinterpret_cast<float*>(at(workspace, step->off_pad_features));
        if ((rc = copy_cols(true, tree->features, tree->K, fp, step->pad_K, tree->M, step->pad_real, stream, fn))) return rc;
        padded_tree = *tree;
        padded_tree.features = fp;
        padded_tree.K = step->pad_K;
        tree = &padded_tree;
        out = reinterpret_cast<float*>(at(workspace, step->off_pad_out));
    }
    auto finish = [&](int r) -> int {
        if (r != SVOXT_OK || !step->pad_K) return r;
        return copy_cols(false, out, step->grad_cols, out_caller, cols_caller, rays->Q, step->pad_real / step->pad_w, stream, fn);
    };
    if (!step->records) return finish(svoxt_volume_render_fwd(tree, rays, opt, out, stream));
    bind_lists(step, workspace);
    svoxt_sample_lists& l = step->lists;
    l.flags &= ~SVOXT_LISTS_BEGUN;
    svoxt_tree t = *tree;
    if (step->uses_mask) {
        // built by every forward: the features' content is the caller's to change between steps.  Lists for a backward
        // hold every sample with sigma > 0: the mask of threshold 0, whatever opt->sigma_thresh says.
        void* mask = at(workspace, step->off_mask);
        if (step->uses_table) {
            if ((rc = svoxt_exp_table_build(tree, 0.f, mask, reinterpret_cast<float*>(at(workspace, step->off_table)), stream))) return rc;
            t.exp_table = reinterpret_cast<const float*>(at(workspace, step->off_table));
        } else {
            // the same launch leaves the lists' tables in the state the forward starts from (-1 everywhere)
            if ((rc = svoxt_sigma_mask_build_fill(tree, 0.f, mask, at(workspace, step->off_tables), step->tables_bytes, stream))) return rc;
            l.flags |= SVOXT_LISTS_BEGUN;
        }
        t.sigma_mask = mask;
        t.sigma_mask_thresh = 0.f;
    }
    if (l.terms_state == 0) { l.terms = nullptr; l.terms_bytes = 0; }       // (wide rows: the backward's scratch, not the forward's)
    rc = svoxt_volume_render_fwd_record(&t, rays, opt, out, &l, stream);
    if (step->off_terms >= 0) l.terms = at(workspace, step->off_terms);
    return finish(rc);
}

int svoxt_step_backward(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt, const float* grad_out,
                        float* grad_features, svoxt_step* step, void* workspace, void* stream) {
    const char* fn = "svoxt_step_backward";
    if (step == nullptr || workspace == nullptr || tree == nullptr || opt == nullptr || rays == nullptr)
        return set_error(SVOXT_ERR_INVALID, "%s: NULL argument", fn);
    if (grad_features == nullptr && tree->M > 0) return set_error(SVOXT_ERR_INVALID, "%s: grad_features is NULL", fn);
    int rc;
    svoxt_tree padded_tree;
    float* grad_caller = grad_features;
    const int32_t K_caller = tree->K;
    if (step->pad_K) {
        // (the padded feature table is the one the forward left in the workspace: the lists' hand-over was formed from it)
        float* gp = reinterpret_cast<float*>(at(workspace, step->off_pad_gout));
        const int cols_caller = svoxt_out_data_dim(opt, tree->K);
        if ((rc = copy_cols(true, grad_out, cols_caller, gp, step->grad_cols, rays->Q, step->pad_real / step->pad_w, stream, fn))) return rc;
        grad_out = gp;
        padded_tree = *tree;
        padded_tree.features = reinterpret_cast<const float*>(at(workspace, step->off_pad_features));
        padded_tree.K = step->pad_K;
        tree = &padded_tree;
        grad_features = reinterpret_cast<float*>(at(workspace, step->off_pad_grad));
    }
    const int64_t M = tree->M;
    const int32_t K = tree->K, gs = step->grad_stride;
    float* rows = step->off_grad_rows >= 0 ? reinterpret_cast<float*>(at(workspace, step->off_grad_rows)) : grad_features;
    if (!step->records) {
        rc = svoxt_volume_render_bwd(tree, rays, opt, grad_out, step->grad_cols, rows, gs,
                                     step->off_bwd_ws >= 0 ? at(workspace, step->off_bwd_ws) : nullptr, step->bwd_ws_bytes, stream);
    } else {
        bind_lists(step, workspace);
        svoxt_sample_lists l = step->lists;
        l.flags &= ~(SVOXT_LISTS_BEGUN | SVOXT_LISTS_GRAD_ZEROED);
        const bool coherent = is_tiled(rays) || rays->order != nullptr || rays->c2w != nullptr;
        const bool wide = wide_rows(tree, opt);
        const bool wide_sh = K > 32 && sh_like(opt) && (opt->basis_dim == 16 || opt->basis_dim == 25) && K == 3 * opt->basis_dim + 1 &&
                             l.terms != nullptr && (l.terms_state == 2 || l.terms_state == 3) && tree->xform == nullptr;
        const bool gather = (K <= 32 || wide_sh) && step->grad_cols == 4 && tree->N == 2 && coherent;
        const bool lobes = opt->format == SVOXT_FORMAT_SG || opt->format == SVOXT_FORMAT_ASG;
        if (lobes && !(gather && tree->xform == nullptr && l.terms != nullptr && l.terms_state == 3)) {
            // (SG / ASG lists serve the exact per-tile backward only: svoxt_step_plan recorded none otherwise)
            return set_error(SVOXT_ERR_INVALID, "%s: the step was planned for a coherent batch", fn);
        }
        const bool fused = gather && (tree->xform == nullptr || (opt->format == SVOXT_FORMAT_SH && opt->basis_dim <= 9));
        const bool wide_tile = wide && tree->N == 2 && tree->xform == nullptr && coherent;
        if (wide) {                                     // scratch between the sweeps of the wide rows' exact backwards
            l.terms_state = 0;
            l.terms_bytes = step->terms_bytes;
        }
        if (fused || wide_tile) l.coef_bytes = -1;      // list walk and per-tile merge as one kernel
        svoxt_tree t = *tree;
        if (wide_tile && step->uses_table) t.exp_table = reinterpret_cast<const float*>(at(workspace, step->off_table));
        rc = svoxt_volume_render_bwd_replay(&t, rays, opt, grad_out, step->grad_cols, rows, gs, &l, nullptr, stream);
    }
    if (rc != SVOXT_OK) return rc;
    if (rows != grad_features && (rc = svoxt_compact_rows(rows, M, K, gs, grad_features, stream))) return rc;
    if (step->pad_K) return copy_cols(false, grad_features, K, grad_caller, K_caller, M, step->pad_real, stream, fn);
    return SVOXT_OK;
}

}  // extern "C"
