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
template <int FMAX>
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
                    if (k < F) acc[k] = (float)((double)acc[k] + (double)weight / (1.0 + (double)pexpf(-pjf[k])));
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
                const float sg = (float)sigmoid_d(pjf[k]);
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
                           tr, to_dev(rays), to_dev(opt), (int)tree->extra_rows, out, depth, hit_point, data_idx);
    else
        hipLaunchKernelGGL((motion_render_kernel<false>), dim3(nb), dim3(kMotionBlock), 0, (hipStream_t)stream,
                           tr, to_dev(rays), to_dev(opt), (int)tree->extra_rows, out, depth, hit_point, data_idx);
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
    const RaysDev rd = to_dev(rays);
    const Opts od = to_dev(opt);
    const int F = motion->feature_dim;
    const unsigned nb = blocks_of(rays->Q, kMotionBlock);
    SVOXT_MOTION_DISPATCH(F,
        if (tree->M > 0)
            hipLaunchKernelGGL((motion_blend_kernel<FMAX>), dim3(blocks_of(tree->M, 256)), dim3(256), 0, st, mo, tree->M, blended);
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
    const RaysDev rd = to_dev(rays);
    const Opts od = to_dev(opt);
    const unsigned nb = blocks_of(rays->Q, kMotionBlock);
    const unsigned nbm = blocks_of(tree->M, 256);
    const bool lds = jbytes <= 65536;
    SVOXT_MOTION_DISPATCH(F,
        hipLaunchKernelGGL((motion_blend_kernel<FMAX>), dim3(nbm), dim3(256), 0, st, mo, tree->M, blended);
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

}  // extern "C"
