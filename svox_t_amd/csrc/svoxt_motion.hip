// svoxt_motion.hip -- the motion variants of the march:
//   motion_render           first sample with sigma > sigma_thresh -> distances to the joints
//                           (rt_kernel.cu:698-778, 837-861, 1480-1504)
//   motion_feature_render   compositing of skinning-weight-blended joint features
//                           (rt_kernel.cu:886-981, 1064-1081, 1525-1543)
//   ... and its gradient wrt joint_features (the derivative of that forward; the
//   reference's own backward, :983-1061, reads an uninitialised local and indexes it
//   by bone instead of channel, SURVEY.md A17, so it defines nothing to match)
// Same ray preamble / leaf stepping (svoxt_device.h) and the same numerical
// contract as the render kernels: every value is bit-identical to the oracle's;
// the backward's sums differ from it only by float accumulation order.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "svoxt_host.h"

#pragma clang fp contract(off)

namespace svoxt {

constexpr int kMotionBlock = 64;

struct MotionDev {
    const float* __restrict__ joint_features;     // [n_joints, F]
    int n_joints, F;
    const float* __restrict__ skinning_weights;   // [M, B]
    const int32_t* __restrict__ joint_index;      // [M, B]
    int B;
};

// motion_trace_ray.  `hit_point` is transform_coord_world (common.cuh:54-60) of the
// LEAF-LOCAL coordinates: the reference's `pos` was rewritten in place by
// query_single_from_root (:745) before it is mapped back (:757) -- kept as is.
template <bool N2>
__global__ void __launch_bounds__(kMotionBlock)
motion_render_kernel(TreeDev tr, RaysDev rays, Opts opt, int J, float* __restrict__ out,
                     float* __restrict__ depth, float* __restrict__ hit_point, int64_t* __restrict__ data_idx) {
    const int64_t q = ray_of_thread(rays, (int64_t)blockIdx.x * kMotionBlock + threadIdx.x);
    if (q >= rays.Q) return;
    float* o = out + q * J;
    for (int j = 0; j < J; ++j) o[j] = 0.f;                  // torch::zeros (:1489-1492)
    depth[q] = 0.f;
    hit_point[3 * q + 0] = 0.f; hit_point[3 * q + 1] = 0.f; hit_point[3 * q + 2] = 0.f;
    data_idx[q] = 0;
    Ray r;
    if (!setup_ray(tr, rays, opt, q, r)) return;
    const int K = tr.K;
    float t = r.tmin;
    while (t < r.tmax) {
        Sample s;
        march_step<N2>(tr, r, opt.step_size, t, s);
        if (s.valid) {
            const float sigma = tr.features[(int64_t)s.idx * K + (K - 1)];
            if (sigma > opt.sigma_thresh) {
                const float px = (s.leaf.lx - tr.offset[0]) / tr.scaling[0];
                const float py = (s.leaf.ly - tr.offset[1]) / tr.scaling[1];
                const float pz = (s.leaf.lz - tr.offset[2]) / tr.scaling[2];
                hit_point[3 * q + 0] = px; hit_point[3 * q + 1] = py; hit_point[3 * q + 2] = pz;
                depth[q] = t * r.delta_scale;
                for (int i = 0; i < J; ++i) {
                    const float* jp = tr.extra + (int64_t)i * tr.extra_cols;
                    const float d0 = px - jp[0], d1 = py - jp[1], d2 = pz - jp[2];
                    o[i] = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
                }
                data_idx[q] = (int64_t)s.idx;
                return;
            }
        }
        t = march_advance(t, s.delta_t);
    }
}

// pos_joint_feature (rt_kernel.cu:946-952) depends on the feature row alone, not on the
// ray: it is evaluated once per row into blended[M, FMAX] (FMAX = F rounded up to
// 4/8/16/32, pad columns zero), in the reference's order of operations, and the
// march then reads one aligned row per sample instead of n_bind scattered joint
// rows.  (First version blended per sample: 64 dword gathers per sample, forward
// 1.37 ms on the headline tree against 0.3 ms for a volume_render of that width.)
// (r04) What the march needs of a blended value is a function of the row alone too: MODE 1 (forward) leaves
// exp(-value) -- the forward's quotient weight / (1 + e) still depends on the sample --, MODE 2 (backward) the float
// sigmoid itself (rt_kernel.cu:1047-1050: sigmoid * (1 - sigmoid) * weight * grad): the same operations on the same
// operands, once per row instead of once per sample and channel (800 x 800 / depth 8 / 16 features: forward 0.58 ->
// see profiles/r04_motion_timing.txt).
template <int FMAX, int MODE>
__global__ void __launch_bounds__(256)
motion_blend_kernel(MotionDev mo, int64_t M, float* __restrict__ blended) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= M) return;
    float pjf[FMAX];
#pragma unroll
    for (int k = 0; k < FMAX; ++k) pjf[k] = 0.f;
    const float* sw = mo.skinning_weights + row * mo.B;
    const int32_t* ji = mo.joint_index + row * mo.B;
    for (int j = 0; j < mo.B; ++j) {
        const float w = sw[j];
        const int32_t joint = ji[j];
        if (w > 0.f && joint >= 0 && joint < mo.n_joints) {
            const float* jf = mo.joint_features + (int64_t)joint * mo.F;
#pragma unroll
            for (int k = 0; k < FMAX; ++k)
                if (k < mo.F) pjf[k] += w * jf[k];
        }
    }
#pragma unroll
    for (int k = 0; k < FMAX; ++k) {
        if constexpr (MODE == 1) pjf[k] = pexpf(-pjf[k]);
        else if constexpr (MODE == 2) pjf[k] = (float)sigmoid_d(pjf[k]);
    }
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f* dst = reinterpret_cast<v4f*>(blended + row * FMAX);
#pragma unroll
    for (int k = 0; k < FMAX; k += 4) dst[k / 4] = v4f{pjf[k], pjf[k + 1], pjf[k + 2], pjf[k + 3]};
}

// motion_feature_trace_ray: accumulators in registers
template <bool N2, int FMAX>
__global__ void __launch_bounds__(kMotionBlock)
motion_feature_fwd_kernel(TreeDev tr, int F, const float* __restrict__ blended, RaysDev rays, Opts opt,
                          float* __restrict__ out) {
    const int64_t q = ray_of_thread(rays, (int64_t)blockIdx.x * kMotionBlock + threadIdx.x);
    if (q >= rays.Q) return;
    float* o = out + q * F;
    Ray r;
    if (!setup_ray(tr, rays, opt, q, r)) {
        for (int j = 0; j < F; ++j) o[j] = 0.f;              // zeros, not the background (:913-919)
        return;
    }
    const int K = tr.K;
    float acc[FMAX];
#pragma unroll
    for (int k = 0; k < FMAX; ++k) acc[k] = 0.f;
    float light = 1.f;
    float t = r.tmin;
    bool stopped = false;
    while (t < r.tmax) {
        Sample s;
        march_step<N2>(tr, r, opt.step_size, t, s);
        if (s.valid) {
            const float sigma = tr.features[(int64_t)s.idx * K + (K - 1)];
            if (sigma > opt.sigma_thresh) {
                const float att = pexpf(-s.delta_t * r.delta_scale * sigma);
                const float weight = light * (1.f - att);
                float pjf[FMAX];
                load_row<FMAX>(blended + (int64_t)s.idx * FMAX, pjf);
#pragma unroll
                for (int k = 0; k < FMAX; ++k)
                    if (k < F) acc[k] = (float)((double)acc[k] + (double)weight / (1.0 + (double)pjf[k]));      // (pjf: exp(-value), MODE 1)
                light *= att;
                if (light <= opt.stop_thresh) {
                    const float scale = (float)(1.0 / (1.0 - (double)light));
#pragma unroll
                    for (int k = 0; k < FMAX; ++k) acc[k] *= scale;
                    stopped = true;
                    break;
                }
            }
        }
        t = march_advance(t, s.delta_t);
    }
#pragma unroll
    for (int k = 0; k < FMAX; ++k)
        if (k < F) o[k] = stopped ? acc[k] : acc[k] + light * opt.background_brightness;
}

// Gradient, stage 1: d/d blended[row][k] = sum over the samples of that row of
// weight * sigmoid' * grad_output[k].  Wave-synchronous march; rows are staged in LDS
// and leave as shaped atomics (flush_staged, as the render backward) into
// grad_blended[M, FMAX].
template <bool N2, int FMAX>
__global__ void __launch_bounds__(kMotionBlock)
motion_feature_bwd_kernel(TreeDev tr, int F, const float* __restrict__ blended, RaysDev rays, Opts opt,
                          const float* __restrict__ grad_out, float* __restrict__ grad_blended) {
    constexpr int KS = FMAX | 1;
    __shared__ float stage[64 * KS];
    __shared__ int32_t sidx[64];
    const int lane = threadIdx.x & 63;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const int64_t q = ray_of_thread(rays, (int64_t)blockIdx.x * kMotionBlock + threadIdx.x);
    Ray r;
    bool alive = q < rays.Q;
    if (alive) alive = setup_ray(tr, rays, opt, q, r);
    if (!__any(alive)) return;
    const int K = tr.K;
    float g[FMAX];
#pragma unroll
    for (int k = 0; k < FMAX; ++k) g[k] = (alive && k < F) ? grad_out[q * F + k] : 0.f;
    float light = 1.f;
    float t = alive ? r.tmin : 0.f;
    const float tmax = alive ? r.tmax : -1.f;
    while (__any(t < tmax)) {
        bool active = false;
        int32_t idx = -1;
        float weight = 0.f;
        if (t < tmax) {
            Sample s;
            march_step<N2>(tr, r, opt.step_size, t, s);
            t = march_advance(t, s.delta_t);
            if (s.valid) {
                const float sigma = tr.features[(int64_t)s.idx * K + (K - 1)];
                if (sigma > 0.f) {                            // thresholds ignored, as in every backward of the reference
                    const float att = pexpf(-s.delta_t * sigma * r.delta_scale);
                    weight = light * (1.f - att);
                    light *= att;
                    active = true;
                    idx = s.idx;
                }
            }
        }
        const unsigned long long amask = __ballot(active);
        if (amask == 0ull) continue;
        if (active) {
            float pjf[FMAX];
            load_row<FMAX>(blended + (int64_t)idx * FMAX, pjf);
            const int slot = __popcll(amask & lane_lt);
            sidx[slot] = idx;
            float* st = stage + slot * KS;
#pragma unroll
            for (int k = 0; k < FMAX; ++k) {
                const float sg = pjf[k];                      // (the row's float sigmoid: motion_blend_kernel<..., 2>)
                st[k] = weight * sg * (1.f - sg) * g[k];      // pad columns: g = 0
            }
        }
        flush_staged<FMAX, KS>(stage, sidx, __popcll(amask), lane, grad_blended, FMAX);
    }
}

// Gradient, stage 2: grad_joint_features[joint_index[row][j]][k] += skinning_weight[row][j] *
// grad_blended[row][k].  Every row adds into the same few joint rows: per-workgroup
// LDS table (ds_add_f32), one global atomic per non-zero cell per workgroup;
// USE_LDS = false: straight global atomics (tables over 64 KiB).
template <int FMAX, bool USE_LDS>
__global__ void __launch_bounds__(256)
motion_reduce_kernel(MotionDev mo, int64_t M, const float* __restrict__ grad_blended, float* __restrict__ grad) {
    extern __shared__ float table[];                          // [n_joints * F] when USE_LDS
    const int F = mo.F;
    const int cells = mo.n_joints * F;
    if constexpr (USE_LDS) {
        for (int i = threadIdx.x; i < cells; i += 256) table[i] = 0.f;
        __syncthreads();
    }
    float* dst = USE_LDS ? table : grad;
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row < M) {
        float gb[FMAX];
        load_row<FMAX>(grad_blended + row * FMAX, gb);
        bool any = false;
#pragma unroll
        for (int k = 0; k < FMAX; ++k) any |= gb[k] != 0.f;
        if (any) {
            const float* sw = mo.skinning_weights + row * mo.B;
            const int32_t* ji = mo.joint_index + row * mo.B;
            for (int j = 0; j < mo.B; ++j) {
                const float w = sw[j];
                const int32_t joint = ji[j];
                if (w > 0.f && joint >= 0 && joint < mo.n_joints) {
#pragma unroll
                    for (int k = 0; k < FMAX; ++k)
                        if (k < F && gb[k] != 0.f) atomicAdd(dst + joint * F + k, w * gb[k]);
                }
            }
        }
    }
    if constexpr (USE_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < cells; i += 256) {
            const float v = table[i];
            if (v != 0.f) atomicAdd(grad + i, v);
        }
    }
}

// ---------------------------------------------------------------------------
// Linear blend skinning of points: warp_vertices (svox_kernel.cu:123-154, 354-378)
// and its backward (:156-211, 404-436).  The per-point [4, 4] matrices it returns
// are what callers hand to volume_render as transformation_matrices.
// ---------------------------------------------------------------------------

// blended[m][n] = sum_j (w_j > 0) w_j * matrices[joint_j][m][n], m < 3, in the reference's order
__device__ __forceinline__ void blend_matrix(const float* __restrict__ matrices, int n_joints,
                                             const float* __restrict__ sw, const int32_t* __restrict__ ji, int B,
                                             float (&mo)[12]) {
#pragma unroll
    for (int e = 0; e < 12; ++e) mo[e] = 0.f;
    for (int j = 0; j < B; ++j) {
        const float w = sw[j];
        const int32_t joint = ji[j];
        if (w > 0.f && joint >= 0 && joint < n_joints) {
            const float* m = matrices + (int64_t)joint * 16;
#pragma unroll
            for (int e = 0; e < 12; ++e) mo[e] += w * m[e];       // rows 0..2 of a row-major 4x4: elements 0..11
        }
    }
}

__global__ void __launch_bounds__(256)
warp_vertices_kernel(const float* __restrict__ matrices, int n_joints, const float* __restrict__ points, int64_t Q,
                     const float* __restrict__ sw, const int32_t* __restrict__ ji, int B,
                     float* __restrict__ vertices_out, float* __restrict__ matrix_out) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) return;
    float mo[12];
    blend_matrix(matrices, n_joints, sw + q * B, ji + q * B, B, mo);
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f* dst = reinterpret_cast<v4f*>(matrix_out + q * 16);
    dst[0] = v4f{mo[0], mo[1], mo[2], mo[3]};
    dst[1] = v4f{mo[4], mo[5], mo[6], mo[7]};
    dst[2] = v4f{mo[8], mo[9], mo[10], mo[11]};
    dst[3] = v4f{0.f, 0.f, 0.f, 1.f};                              // :148
    const float x = points[3 * q], y = points[3 * q + 1], z = points[3 * q + 2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        vertices_out[3 * q + i] = x * mo[4 * i] + y * mo[4 * i + 1] + z * mo[4 * i + 2] + mo[4 * i + 3];   // :151-153
}

// grad_points and grad_skinning_weights are per point (written, not accumulated);
// grad_matrices [n_joints, 4, 4] meets all points: per-workgroup LDS table, one global
// atomic per touched cell per workgroup (USE_LDS = false: tables over 64 KiB).
template <bool USE_LDS>
__global__ void __launch_bounds__(256)
warp_vertices_bwd_kernel(const float* __restrict__ matrices, int n_joints, const float* __restrict__ points,
                         int64_t Q, const float* __restrict__ sw_all, const int32_t* __restrict__ ji_all, int B,
                         const float* __restrict__ grad_vertices, const float* __restrict__ grad_matrix_out,
                         float* __restrict__ grad_points, float* __restrict__ grad_matrices,
                         float* __restrict__ grad_sw) {
    extern __shared__ float table[];                              // [n_joints * 16] when USE_LDS
    const int cells = n_joints * 16;
    if constexpr (USE_LDS) {
        for (int i = threadIdx.x; i < cells; i += 256) table[i] = 0.f;
        __syncthreads();
    }
    float* dst = USE_LDS ? table : grad_matrices;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q < Q) {
        const float* sw = sw_all + q * B;
        const int32_t* ji = ji_all + q * B;
        float mo[12];
        blend_matrix(matrices, n_joints, sw, ji, B, mo);
        float mg[12];                                             // rows 0..2 of matrices_grad_out[q]
#pragma unroll
        for (int e = 0; e < 12; ++e) mg[e] = grad_matrix_out[q * 16 + e];
        const float g0 = grad_vertices[3 * q], g1 = grad_vertices[3 * q + 1], g2 = grad_vertices[3 * q + 2];
        const float gv[3] = {g0, g1, g2};
        const float x = points[3 * q], y = points[3 * q + 1], z = points[3 * q + 2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            grad_points[3 * q + i] = g0 * mo[i] + g1 * mo[4 + i] + g2 * mo[8 + i];       // :193
        float tg[12];                                             // tmp_grad_matrix (:194-197)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            tg[4 * i] = gv[i] * x; tg[4 * i + 1] = gv[i] * y; tg[4 * i + 2] = gv[i] * z; tg[4 * i + 3] = gv[i];
        }
        for (int j = 0; j < B; ++j) {
            const float w = sw[j];
            const int32_t joint = ji[j];
            float gs = 0.f;                                       // torch::zeros (:419)
            if (w > 0.f && joint >= 0 && joint < n_joints) {
                const float* m = matrices + (int64_t)joint * 16;
                // the reference adds into grad_skinning_weights[q][j] first over its loop at
                // :176-185 (matrices_grad_out), then over the one at :200-208 (tmp_grad_matrix)
#pragma unroll
                for (int e = 0; e < 12; ++e) gs += m[e] * mg[e];
#pragma unroll
                for (int e = 0; e < 12; ++e) gs += m[e] * tg[e];
#pragma unroll
                for (int e = 0; e < 12; ++e) {
                    atomicAdd(dst + joint * 16 + e, w * mg[e]);
                    atomicAdd(dst + joint * 16 + e, w * tg[e]);
                }
            }
            grad_sw[q * B + j] = gs;
        }
    }
    if constexpr (USE_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < cells; i += 256) {
            const float v = table[i];
            if (v != 0.f) atomicAdd(grad_matrices + i, v);
        }
    }
}

}  // namespace svoxt

using namespace svoxt;

namespace {

int check_motion(const svoxt_motion* m, const svoxt_tree* t, const char* fn) {
    if (m == nullptr) return set_error(SVOXT_ERR_INVALID, "%s: motion is NULL", fn);
    if (m->joint_features == nullptr || m->n_joints < 1)
        return set_error(SVOXT_ERR_INVALID, "%s: joint_features is NULL or empty", fn);
    if (m->feature_dim < 1 || m->feature_dim > 32)
        return set_error(SVOXT_ERR_INVALID, "%s: joint feature dim must be in [1, 32] (the reference's tmp_data_dim)", fn);
    if (m->n_bind < 1 || (t->M > 0 && (m->skinning_weights == nullptr || m->joint_index == nullptr)))
        return set_error(SVOXT_ERR_INVALID, "%s: skinning_weights / joint_index is NULL or n_bind < 1", fn);
    return SVOXT_OK;
}

MotionDev to_dev(const svoxt_motion* m) {
    MotionDev d;
    d.joint_features = m->joint_features; d.n_joints = m->n_joints; d.F = m->feature_dim;
    d.skinning_weights = m->skinning_weights; d.joint_index = m->joint_index; d.B = m->n_bind;
    return d;
}

unsigned blocks_of(int64_t Q, int block) { return (unsigned)((Q + block - 1) / block); }

int fmax_of(int F) { return F <= 4 ? 4 : F <= 8 ? 8 : F <= 16 ? 16 : 32; }

// run BODY with the compile-time FMAX matching the runtime feature width
#define SVOXT_MOTION_DISPATCH(F, BODY)                         \
    switch (fmax_of(F)) {                                      \
        case 4: { constexpr int FMAX = 4; BODY } break;        \
        case 8: { constexpr int FMAX = 8; BODY } break;        \
        case 16: { constexpr int FMAX = 16; BODY } break;      \
        default: { constexpr int FMAX = 32; BODY } break;      \
    }

}  // namespace

extern "C" {

int svoxt_motion_render(const svoxt_tree* tree, const svoxt_rays* rays, const svoxt_options* opt,
                        float* out, float* depth, float* hit_point, int64_t* data_idx, void* stream) {
    const char* fn = "svoxt_motion_render";
    int rc;
    if ((rc = check_tree(tree, fn)) || (rc = check_rays(rays, fn)) || (rc = check_opts(opt, tree, fn, false))) return rc;
    if (tree->extra_data == nullptr || tree->extra_rows < 1 || tree->extra_cols < 3)
        return set_error(SVOXT_ERR_INVALID, "%s: needs extra_data [n_joints, >= 3] (the joint positions)", fn);
    if (rays->Q == 0) return SVOXT_OK;
    if (out == nullptr || depth == nullptr || hit_point == nullptr || data_idx == nullptr)
        return set_error(SVOXT_ERR_INVALID, "%s: an output pointer is NULL", fn);
    const TreeDev tr = to_dev(tree);
    const unsigned nb = blocks_of(rays->Q, kMotionBlock);
    if (tree->N == 2)
        hipLaunchKernelGGL((motion_render_kernel<true>), dim3(nb), dim3(kMotionBlock), 0, (hipStream_t)stream,
                           tr, to_dev(rays, tree), to_dev(opt), (int)tree->extra_rows, out, depth, hit_point, data_idx);
    else
        hipLaunchKernelGGL((motion_render_kernel<false>), dim3(nb), dim3(kMotionBlock), 0, (hipStream_t)stream,
                           tr, to_dev(rays, tree), to_dev(opt), (int)tree->extra_rows, out, depth, hit_point, data_idx);
    return check_launch(fn);
}

int64_t svoxt_motion_workspace_bytes(int64_t M, int32_t feature_dim) {
    if (M < 0 || feature_dim < 1 || feature_dim > 32) return -1;
    return 2 * M * fmax_of(feature_dim) * (int64_t)sizeof(float) + 32;
}

static int motion_common(const char* fn, const svoxt_tree* tree, const svoxt_motion* motion, const svoxt_rays* rays,
                         const svoxt_options* opt, void* workspace, int64_t workspace_bytes, float** blended) {
    int rc;
    if ((rc = check_tree(tree, fn)) || (rc = check_rays(rays, fn)) || (rc = check_opts(opt, tree, fn, false)) ||
        (rc = check_motion(motion, tree, fn))) return rc;
    if (workspace == nullptr || ((uintptr_t)workspace & 15) != 0 ||
        workspace_bytes < svoxt_motion_workspace_bytes(tree->M, motion->feature_dim))
        return set_error(SVOXT_ERR_INVALID, "%s: workspace is NULL, not 16-byte aligned, or smaller than svoxt_motion_workspace_bytes", fn);
    *blended = reinterpret_cast<float*>(workspace);
    return SVOXT_OK;
}

int svoxt_motion_feature_render_fwd(const svoxt_tree* tree, const svoxt_motion* motion, const svoxt_rays* rays,
                                    const svoxt_options* opt, float* out, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
    const char* fn = "svoxt_motion_feature_render_fwd";
    float* blended = nullptr;
    int rc;
    if ((rc = motion_common(fn, tree, motion, rays, opt, workspace, workspace_bytes, &blended))) return rc;
    if (rays->Q == 0) return SVOXT_OK;
    if (out == nullptr) return set_error(SVOXT_ERR_INVALID, "%s: out is NULL", fn);
    hipStream_t st = (hipStream_t)stream;
    const TreeDev tr = to_dev(tree);
    const MotionDev mo = to_dev(motion);
    const RaysDev rd = to_dev(rays, tree);
    const Opts od = to_dev(opt);
    const int F = motion->feature_dim;
    const unsigned nb = blocks_of(rays->Q, kMotionBlock);
    SVOXT_MOTION_DISPATCH(F,
        if (tree->M > 0)
            hipLaunchKernelGGL((motion_blend_kernel<FMAX, 1>), dim3(blocks_of(tree->M, 256)), dim3(256), 0, st, mo, tree->M, blended);
        if (tree->N == 2)
            hipLaunchKernelGGL((motion_feature_fwd_kernel<true, FMAX>), dim3(nb), dim3(kMotionBlock), 0, st, tr, F, blended, rd, od, out);
        else
            hipLaunchKernelGGL((motion_feature_fwd_kernel<false, FMAX>), dim3(nb), dim3(kMotionBlock), 0, st, tr, F, blended, rd, od, out);
    )
    return check_launch(fn);
}

int svoxt_motion_feature_render_bwd(const svoxt_tree* tree, const svoxt_motion* motion, const svoxt_rays* rays,
                                    const svoxt_options* opt, const float* grad_out, float* grad_joint_features,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
    const char* fn = "svoxt_motion_feature_render_bwd";
    float* blended = nullptr;
    int rc;
    if ((rc = motion_common(fn, tree, motion, rays, opt, workspace, workspace_bytes, &blended))) return rc;
    if (grad_joint_features == nullptr) return set_error(SVOXT_ERR_INVALID, "%s: grad_joint_features is NULL", fn);
    hipStream_t st = (hipStream_t)stream;
    const int F = motion->feature_dim;
    const size_t jbytes = sizeof(float) * (size_t)motion->n_joints * F;
    hipError_t e = hipMemsetAsync(grad_joint_features, 0, jbytes, st);
    if (e != hipSuccess) return set_error(SVOXT_ERR_HIP, "%s: %s", fn, hipGetErrorString(e));
    if (rays->Q == 0 || tree->M == 0) return SVOXT_OK;
    if (grad_out == nullptr) return set_error(SVOXT_ERR_INVALID, "%s: grad_out is NULL", fn);
    const int fmax = fmax_of(F);
    float* grad_blended = blended + tree->M * fmax;
    e = hipMemsetAsync(grad_blended, 0, sizeof(float) * (size_t)tree->M * fmax, st);
    if (e != hipSuccess) return set_error(SVOXT_ERR_HIP, "%s: %s", fn, hipGetErrorString(e));
    const TreeDev tr = to_dev(tree);
    const MotionDev mo = to_dev(motion);
    const RaysDev rd = to_dev(rays, tree);
    const Opts od = to_dev(opt);
    const unsigned nb = blocks_of(rays->Q, kMotionBlock);
    const unsigned nbm = blocks_of(tree->M, 256);
    const bool lds = jbytes <= 65536;
    SVOXT_MOTION_DISPATCH(F,
        hipLaunchKernelGGL((motion_blend_kernel<FMAX, 2>), dim3(nbm), dim3(256), 0, st, mo, tree->M, blended);
        if (tree->N == 2)
            hipLaunchKernelGGL((motion_feature_bwd_kernel<true, FMAX>), dim3(nb), dim3(kMotionBlock), 0, st, tr, F, blended, rd, od, grad_out, grad_blended);
        else
            hipLaunchKernelGGL((motion_feature_bwd_kernel<false, FMAX>), dim3(nb), dim3(kMotionBlock), 0, st, tr, F, blended, rd, od, grad_out, grad_blended);
        if (lds)
            hipLaunchKernelGGL((motion_reduce_kernel<FMAX, true>), dim3(nbm), dim3(256), jbytes, st, mo, tree->M, grad_blended, grad_joint_features);
        else
            hipLaunchKernelGGL((motion_reduce_kernel<FMAX, false>), dim3(nbm), dim3(256), 0, st, mo, tree->M, grad_blended, grad_joint_features);
    )
    return check_launch(fn);
}

static int check_warp(const char* fn, const float* matrices, int32_t n_joints, const float* points, int64_t Q,
                      const float* sw, const int32_t* ji, int32_t n_bind) {
    if (Q < 0 || n_joints < 1 || n_bind < 1) return set_error(SVOXT_ERR_INVALID, "%s: bad extents", fn);
    if (matrices == nullptr || (Q > 0 && (points == nullptr || sw == nullptr || ji == nullptr)))
        return set_error(SVOXT_ERR_INVALID, "%s: matrices / points / skinning_weights / joint_index is NULL", fn);
    return SVOXT_OK;
}

int svoxt_warp_vertices(const float* matrices, int32_t n_joints, const float* points, int64_t Q,
                        const float* skinning_weights, const int32_t* joint_index, int32_t n_bind,
                        float* vertices_out, float* matrix_out, void* stream) {
    const char* fn = "svoxt_warp_vertices";
    int rc;
    if ((rc = check_warp(fn, matrices, n_joints, points, Q, skinning_weights, joint_index, n_bind))) return rc;
    if (Q == 0) return SVOXT_OK;
    if (vertices_out == nullptr || matrix_out == nullptr || ((uintptr_t)matrix_out & 15) != 0)
        return set_error(SVOXT_ERR_INVALID, "%s: an output is NULL or matrix_out is not 16-byte aligned", fn);
    hipLaunchKernelGGL(warp_vertices_kernel, dim3(blocks_of(Q, 256)), dim3(256), 0, (hipStream_t)stream,
                       matrices, (int)n_joints, points, Q, skinning_weights, joint_index, (int)n_bind,
                       vertices_out, matrix_out);
    return check_launch(fn);
}

int svoxt_warp_vertices_bwd(const float* matrices, int32_t n_joints, const float* points, int64_t Q,
                            const float* skinning_weights, const int32_t* joint_index, int32_t n_bind,
                            const float* grad_vertices, const float* grad_matrix_out,
                            float* grad_points, float* grad_matrices, float* grad_skinning_weights,
                            void* stream) {
    const char* fn = "svoxt_warp_vertices_bwd";
    int rc;
    if ((rc = check_warp(fn, matrices, n_joints, points, Q, skinning_weights, joint_index, n_bind))) return rc;
    if (grad_matrices == nullptr) return set_error(SVOXT_ERR_INVALID, "%s: grad_matrices is NULL", fn);
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = sizeof(float) * 16 * (size_t)n_joints;
    const hipError_t e = hipMemsetAsync(grad_matrices, 0, bytes, st);
    if (e != hipSuccess) return set_error(SVOXT_ERR_HIP, "%s: %s", fn, hipGetErrorString(e));
    if (Q == 0) return SVOXT_OK;
    if (grad_vertices == nullptr || grad_matrix_out == nullptr || grad_points == nullptr || grad_skinning_weights == nullptr)
        return set_error(SVOXT_ERR_INVALID, "%s: a gradient pointer is NULL", fn);
    const unsigned nb = blocks_of(Q, 256);
    if (bytes <= 65536)
        hipLaunchKernelGGL((warp_vertices_bwd_kernel<true>), dim3(nb), dim3(256), bytes, st, matrices, (int)n_joints,
                           points, Q, skinning_weights, joint_index, (int)n_bind, grad_vertices, grad_matrix_out,
                           grad_points, grad_matrices, grad_skinning_weights);
    else
        hipLaunchKernelGGL((warp_vertices_bwd_kernel<false>), dim3(nb), dim3(256), 0, st, matrices, (int)n_joints,
                           points, Q, skinning_weights, joint_index, (int)n_bind, grad_vertices, grad_matrix_out,
                           grad_points, grad_matrices, grad_skinning_weights);
    return check_launch(fn);
}

}  // extern "C"
