// svoxt_oracle.cpp -- CPU restatement of svox_t's volume-render hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under svox_t_amd/ may import, link or
// call this file; it is the checker for the HIP kernels (tests/, the smoke
// check in __graft_entry__.py and the cpu_baseline leg of bench.py).
//
// PARITY UNPINNED for the render / query arithmetic, in the sense of the task
// statement: the reference ships no tests and no golden vectors for this path,
// its CUDA sources cannot be built here (no nvcc) and every CPU render entry of
// its Python asserts, so no output of the reference itself exists to compare
// with.  This file is a restatement of the source semantics, held in place by
// (a) closed-form known answers, (b) fp64 finite-difference checks of its own
// backward, (c) fixtures captured by importing the reference's Python on CPU
// (tests/golden/make_golden.py: tree topology from the reference's refine, SH
// polynomials and row layout from its sh.py, format table, world->tree
// transform) and (d) an independent vectorised PyTorch renderer with autograd
// (oracle/torch_renderer.py).  See DESIGN.md section 5.
//
// Every function cites the reference lines it follows (paths relative to
// /root/reference).  Arithmetic is written so that each C++ expression has
// the same operand types, association and rounding points as the reference
// expression; the file must be compiled with -ffp-contract=off so that no
// multiply-add is fused (the HIP kernels keep the stepping arithmetic
// bit-identical to this file).
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -fopenmp).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---- svox_t/csrc/include/data_spec.hpp:45-50 -------------------------------
enum { FORMAT_RGBA = 0, FORMAT_SH = 1, FORMAT_SG = 2, FORMAT_ASG = 3 };

// ---- svox_t/csrc/include/data_spec.hpp:129-145 (field order kept) ----------
struct RenderOptions {
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

// Tree as the kernels see it (svox_t/csrc/include/data_spec_packed.cuh:57-100)
template <typename T>
struct Tree {
    const T* features;      // [M, K]
    int64_t M;
    int K;
    const int32_t* data;    // [cap, N, N, N, 1] feature-row index per slot
    const int32_t* child;   // [cap, N, N, N]    relative child offset, 0 = leaf
    int N;
    T offset[3];
    T scaling[3];
    const T* extra;         // [extra_rows, extra_cols] (SG / ASG lobes)
    int extra_rows, extra_cols;
    const T* xform;         // transformation_matrices [M, d, d] or null (rt_kernel.cu:283-291)
    int xform_dim;          // d: 3 or 4
};

struct Counters {           // sums that define algorithmic bytes
    int64_t rays_hit;       // rays that enter the cube
    int64_t steps;          // S   = number of leaf crossings
    int64_t levels;         // sum of L (child words read) over all steps
    int64_t valid;          // steps whose leaf holds a valid feature index
    int64_t active;         // steps whose sample is composited
};

// SH constants: declared `const float` in the reference, i.e. rounded to
// float before use (svox_t/csrc/rt_kernel.cu:54-84).
const float kC0 = 0.28209479177387814;
const float kC1 = 0.4886025119029199;
const float kC2[5] = {1.0925484305920792, -1.0925484305920792,
                      0.31539156525252005, -1.0925484305920792,
                      0.5462742152960396};
const float kC3[7] = {-0.5900435899266435, 2.890611442640554,
                      -0.4570457994644658, 0.3731763325901154,
                      -0.4570457994644658, 1.445305721320277,
                      -0.5900435899266435};
const float kC4[9] = {2.5033429417967046,  -1.7701307697799304,
                      0.9461746957575601,  -0.6690465435572892,
                      0.10578554691520431, -0.6690465435572892,
                      0.47308734787878004, -1.7701307697799304,
                      0.6258357354491761};

template <typename T> inline T exp_T(T x);
// expf with a fixed sequence of correctly rounded operations; the HIP kernels
// carry the same sequence (svox_t_amd/csrc/svoxt_device.h, pexpf), so the two
// sides agree bit for bit instead of "to within the ulps of two different
// libms".  ~1 ulp accurate (CUDA's expf, which the reference calls, is
// specified to 2 ulp).  `svoxt_oracle_use_libm_exp(1)` switches to glibc's
// expf; tests use that to show the choice of expf moves results only at the
// level of a 1-ulp perturbation.
static bool g_use_libm_exp = false;
inline float portable_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -87.0f) return 0.0f;
    const float n = rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
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
    union { int32_t i; float f; } s1, s2;
    s1.i = (n1 + 127) << 23;
    s2.i = (n2 + 127) << 23;
    y = y * s1.f;
    return y * s2.f;
}
template <> inline float exp_T<float>(float x) { return g_use_libm_exp ? expf(x) : portable_expf(x); }
template <> inline double exp_T<double>(double x) { return exp(x); }

// rt_kernel.cu:88-91 (`sqrtf` even for double in the reference; for the
// double instantiation we keep full precision: it is only used for
// finite-difference checks, never for parity).
template <typename T> inline T norm3(const T* d);
template <> inline float norm3<float>(const float* d) {
    return sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}
template <> inline double norm3<double>(const double* d) {
    return sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}

// rt_kernel.cu:100-105
template <typename T> inline T dot3(const T* u, const T* v) {
    return u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
}

// rt_kernel.cu:110-185.  basis values for one view direction.
template <typename T>
void precalc_basis(int format, int basis_dim, const Tree<T>& tree,
                   const T* dir, T* out) {
    if (format == FORMAT_ASG) {
        // rt_kernel.cu:118-130 (marked UNTESTED upstream)
        for (int i = 0; i < basis_dim; ++i) {
            const T* lobe = tree.extra + (int64_t)i * tree.extra_cols;
            T S = dot3(dir, lobe + 8);
            T dot_x = dot3(dir, lobe + 2);
            T dot_y = dot3(dir, lobe + 5);
            out[i] = S * exp_T<T>(-lobe[0] * dot_x * dot_x - lobe[1] * dot_y * dot_y) / basis_dim;
        }
    } else if (format == FORMAT_SG) {
        // rt_kernel.cu:131-138
        for (int i = 0; i < basis_dim; ++i) {
            const T* lobe = tree.extra + (int64_t)i * tree.extra_cols;
            out[i] = exp_T<T>(lobe[0] * (dot3(dir, lobe + 1) - 1.f)) / basis_dim;
        }
    } else if (format == FORMAT_SH) {
        // rt_kernel.cu:139-178 (the reference falls through 25 -> 16 -> 9 -> 4)
        out[0] = kC0;
        const T x = dir[0], y = dir[1], z = dir[2];
        const T xx = x * x, yy = y * y, zz = z * z;
        const T xy = x * y, yz = y * z, xz = x * z;
        const bool d25 = basis_dim == 25;
        const bool d16 = d25 || basis_dim == 16;
        const bool d9 = d16 || basis_dim == 9;
        const bool d4 = d9 || basis_dim == 4;
        if (d25) {
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
        if (d16) {
            out[9] = kC3[0] * y * (3 * xx - yy);
            out[10] = kC3[1] * xy * z;
            out[11] = kC3[2] * y * (4 * zz - xx - yy);
            out[12] = kC3[3] * z * (2 * zz - 3 * xx - 3 * yy);
            out[13] = kC3[4] * x * (4 * zz - xx - yy);
            out[14] = kC3[5] * z * (xx - yy);
            out[15] = kC3[6] * x * (xx - 3 * yy);
        }
        if (d9) {
            out[4] = kC2[0] * xy;
            out[5] = kC2[1] * yz;
            out[6] = kC2[2] * (2.0 * zz - xx - yy);  // double term, :169
            out[7] = kC2[3] * xz;
            out[8] = kC2[4] * (xx - yy);
        }
        if (d4) {
            out[1] = -kC1 * y;
            out[2] = kC1 * z;
            out[3] = -kC1 * x;
        }
    }
}

// Per-leaf view-direction rotation (rt_kernel.cu:283-291, :387-395): the basis
// is re-evaluated for ray_dir = M[idx] * vdir.
template <typename T>
inline void rotated_basis(const Tree<T>& tree, const RenderOptions& opt, int32_t idx, const T* vdir, T* basis_fn) {
    const int d = tree.xform_dim;
    const T* m = tree.xform + (int64_t)idx * (d * d);
    T ray_dir[3];
    ray_dir[0] = m[0] * vdir[0] + m[1] * vdir[1] + m[2] * vdir[2];
    ray_dir[1] = m[d] * vdir[0] + m[d + 1] * vdir[1] + m[d + 2] * vdir[2];
    ray_dir[2] = m[2 * d] * vdir[0] + m[2 * d + 1] * vdir[1] + m[2 * d + 2] * vdir[2];
    precalc_basis<T>(opt.format, opt.basis_dim, tree, ray_dir, basis_fn);
}

// rt_kernel.cu:188-199
template <typename T>
inline T get_delta_scale(const T* scaling, T* dir) {
    dir[0] *= scaling[0];
    dir[1] *= scaling[1];
    dir[2] *= scaling[2];
    T delta_scale = 1.f / norm3<T>(dir);
    dir[0] *= delta_scale;
    dir[1] *= delta_scale;
    dir[2] *= delta_scale;
    return delta_scale;
}

// rt_kernel.cu:202-218
template <typename T>
inline void dda_unit(const T* cen, const T* invdir, T* tmin, T* tmax) {
    T t1, t2;
    *tmin = 0.0f;
    *tmax = 1e9f;
    for (int i = 0; i < 3; ++i) {
        t1 = -cen[i] * invdir[i];
        t2 = t1 + invdir[i];
        *tmin = std::max(*tmin, std::min(t1, t2));
        *tmax = std::min(*tmax, std::max(t1, t2));
    }
}

// common.cuh:45-51
template <typename T>
inline void transform_coord(T* q, const T* offset, const T* scaling) {
    for (int i = 0; i < 3; ++i) q[i] = offset[i] + scaling[i] * q[i];
}

// common.cuh:38-42 + 63-100.  Returns the flat slot index
// ((node*N+u)*N+v)*N+w of the leaf; xyz becomes leaf-local, *cube_sz = N^k.
// The upper clamp is evaluated in double then rounded to T (`scalar_t(1.0) -
// 1e-6` is a double; CUDA's mixed min/max overloads return double).
template <typename T>
inline int64_t query_from_root(const Tree<T>& tree, T* xyz, T* cube_sz,
                               int* levels_out) {
    const T N = tree.N;
    for (int i = 0; i < 3; ++i) {
        xyz[i] = (T)std::max(0.0, std::min((double)T(1.0) - 1e-6, (double)xyz[i]));
    }
    int32_t node_id = 0;
    int levels = 0;
    *cube_sz = N;
    while (true) {
        xyz[0] *= N;
        xyz[1] *= N;
        xyz[2] *= N;
        int32_t u = (int32_t)std::floor(xyz[0]);
        int32_t v = (int32_t)std::floor(xyz[1]);
        int32_t w = (int32_t)std::floor(xyz[2]);
        xyz[0] -= u;
        xyz[1] -= v;
        xyz[2] -= w;
        // The reference would index out of bounds if rounding produced N;
        // keep the access in range (differs only where the reference is UB).
        u = std::min(std::max(u, 0), tree.N - 1);
        v = std::min(std::max(v, 0), tree.N - 1);
        w = std::min(std::max(w, 0), tree.N - 1);
        const int64_t slot = (((int64_t)node_id * tree.N + u) * tree.N + v) * tree.N + w;
        const int32_t skip = tree.child[slot];
        ++levels;
        if (skip == 0) {
            if (levels_out) *levels_out = levels;
            return slot;
        }
        *cube_sz *= N;
        node_id += skip;
    }
}

// Per-ray preamble shared by every trace variant (rt_kernel.cu:227-247,
// :339-356, :505-522, :787-803 and the kernel wrappers :661-670).
template <typename T>
struct RaySetup {
    T origin[3], dir[3], vdir[3], invdir[3];
    T delta_scale, tmin, tmax;
    bool hit;
};

template <typename T>
inline RaySetup<T> setup_ray(const Tree<T>& tree, const T* o, const T* d, const T* vd) {
    RaySetup<T> r;
    for (int i = 0; i < 3; ++i) { r.origin[i] = o[i]; r.dir[i] = d[i]; r.vdir[i] = vd[i]; }
    transform_coord<T>(r.origin, tree.offset, tree.scaling);   // :664
    r.delta_scale = get_delta_scale<T>(tree.scaling, r.dir);   // :227
    for (int i = 0; i < 3; ++i) r.invdir[i] = 1.0 / (r.dir[i] + 1e-9);  // :237 double
    dda_unit<T>(r.origin, r.invdir, &r.tmin, &r.tmax);         // :239
    r.hit = !(r.tmax < 0 || r.tmin > r.tmax);                  // :241
    return r;
}

// One leaf crossing (rt_kernel.cu:261-277).  Returns the feature row or null.
template <typename T>
struct Step {
    const T* row;   // null when the leaf is empty
    int32_t idx;
    T delta_t;
    T sigma;
    int levels;
    int64_t slot;
};

template <typename T>
inline Step<T> march_step(const Tree<T>& tree, const RaySetup<T>& r,
                          const RenderOptions& opt, T t) {
    Step<T> s;
    T pos[3];
    for (int j = 0; j < 3; ++j) pos[j] = r.origin[j] + t * r.dir[j];
    T cube_sz;
    s.slot = query_from_root<T>(tree, pos, &cube_sz, &s.levels);
    s.idx = tree.data[s.slot];
    // :269 compares int32 with int64 size(0): a negative index would pass and
    // read out of bounds in the reference; it is treated as empty here.
    s.row = (s.idx < 0 || (int64_t)s.idx >= tree.M) ? nullptr : tree.features + (int64_t)s.idx * tree.K;
    T sub_tmin, sub_tmax;
    dda_unit<T>(pos, r.invdir, &sub_tmin, &sub_tmax);
    const T t_subcube = (sub_tmax - sub_tmin) / cube_sz;
    s.delta_t = t_subcube + opt.step_size;
    s.sigma = s.row != nullptr ? s.row[tree.K - 1] : 0.0;
    return s;
}

// rt_kernel.cu:222-328
// weight_accum (nullable, [n_internal * N^3] doubles): per leaf slot, the sum of the compositing
// weights of the samples taken there -- `tree.weight_accum[node_id] += weight` with node_id the
// packed leaf id query_single_from_root returns (rt_kernel.cu:266-267, :309-311; common.cuh:90-93).
// The reference adds without atomics (it races); the sum is what it means.  Kept in
// double so that it does not depend on the order of the rays.
template <typename T>
void trace_ray(const Tree<T>& tree, const RaySetup<T>& r, const RenderOptions& opt,
               T* out, int out_data_dim, Counters* cnt, double* weight_accum = nullptr) {
    if (!r.hit) {
        for (int j = 0; j < out_data_dim; ++j) out[j] = opt.background_brightness;
        out[out_data_dim] = 0;
        return;
    }
    if (cnt) cnt->rays_hit++;
    for (int j = 0; j < out_data_dim; ++j) out[j] = 0.f;
    T basis_fn[25];
    precalc_basis<T>(opt.format, opt.basis_dim, tree, r.vdir, basis_fn);

    T light_intensity = 1.f;
    T t = r.tmin;
    while (t < r.tmax) {
        const Step<T> s = march_step<T>(tree, r, opt, t);
        if (cnt) { cnt->steps++; cnt->levels += s.levels; cnt->valid += s.row != nullptr; }
        if (s.sigma > opt.sigma_thresh) {
            if (cnt) cnt->active++;
            const T att = exp_T<T>(-s.delta_t * r.delta_scale * s.sigma);
            const T weight = light_intensity * (1.f - att);
            if (tree.xform != nullptr) rotated_basis<T>(tree, opt, s.idx, r.vdir, basis_fn);
            if (opt.format != FORMAT_RGBA) {
                for (int c = 0; c < out_data_dim; ++c) {
                    const int off = c * opt.basis_dim;
                    T tmp = 0.0;
                    for (int i = opt.min_comp; i <= opt.max_comp; ++i)
                        tmp += basis_fn[i] * s.row[off + i];
                    out[c] += weight / (1.0 + exp_T<T>(-tmp));
                }
            } else {
                for (int j = 0; j < out_data_dim; ++j)
                    out[j] += weight / (1.0 + exp_T<T>(-s.row[j]));
            }
            light_intensity *= att;
            if (weight_accum != nullptr) {             // :309-311, after the transmittance update
#pragma omp atomic
                weight_accum[s.slot] += (double)weight;
            }
            if (light_intensity <= opt.stop_thresh) {
                T scale = 1.0 / (1.0 - light_intensity);
                for (int j = 0; j != out_data_dim; ++j) out[j] *= scale;
                out[out_data_dim] = 1 - light_intensity;
                return;
            }
        }
        t += s.delta_t;
    }
    for (int j = 0; j < out_data_dim; ++j)
        out[j] += light_intensity * opt.background_brightness;
    out[out_data_dim] = 1 - light_intensity;
}

// rt_kernel.cu:331-496.  `grad` is accumulated in double so that the oracle's
// sums do not depend on ray order (the reference uses float atomicAdd, whose
// order is undefined); each contribution is still computed in T exactly as
// the reference computes `toadd`.  `abs_sum` (nullable) receives, per entry,
// the summed magnitude of the terms entering each contribution (|toadd| for
// colour entries; for sigma entries the size of the operands of the
// difference): the scale against which tests bound float-atomic reordering
// and accumulation-order error.
template <typename T>
void trace_ray_backward(const Tree<T>& tree, const RaySetup<T>& r, const RenderOptions& opt,
                        const T* grad_output, int out_data_dim,
                        double* grad, double* abs_sum, bool atomic, double* abs_sum_tight = nullptr) {
    if (!r.hit) return;
    T basis_fn[25];
    precalc_basis<T>(opt.format, opt.basis_dim, tree, r.vdir, basis_fn);
    const int K = tree.K;

    // mag: magnitude of the terms that enter v before any cancellation
    // abs_sum_tight: the same with the size of `accum` taken as the sum of |w_j * total_color_j| and
    // |T * bg * sum_c g_c| -- the quantities the reference's own sequential pass 1 adds
    // (rt_kernel.cu:428-436) -- i.e. what bounds a backward that forms accum the way the reference
    // does (two walks: SVOXT_BWD_EXACT).  abs_sum takes the sum of |w_j * s_jc * g_c| and
    // |T * bg * g_c| instead: what bounds ANY order of adding those products, e.g. accum formed as
    // sum_c g_c * out_c from the forward's output (the single-march backward).
    auto add = [&](int64_t e, T v, double mag, double mag_tight) {
        if (atomic) {
#pragma omp atomic
            grad[e] += (double)v;
            if (abs_sum) {
#pragma omp atomic
                abs_sum[e] += mag;
            }
            if (abs_sum_tight) {
#pragma omp atomic
                abs_sum_tight[e] += mag_tight;
            }
        } else {
            grad[e] += (double)v;
            if (abs_sum) abs_sum[e] += mag;
            if (abs_sum_tight) abs_sum_tight[e] += mag_tight;
        }
    };

    T accum = 0.0;
    // sum of the magnitudes of every elementary product that enters accum
    // (w_j * s_jc * g_c and T * bg * g_c): the size its rounding error scales with,
    // whichever order the products are added in
    double accum_scale = 0.0;
    double accum_scale_tight = 0.0;
    T light_intensity_ray = 0.0;
    {   // PASS 1 (:365-437)
        T light_intensity = 1.f, t = r.tmin;
        while (t < r.tmax) {
            const Step<T> s = march_step<T>(tree, r, opt, t);
            if (s.sigma > 0.0) {
                const int64_t base = (int64_t)s.idx * K;
                // pass 1 re-evaluates the rotated basis (:387-395); pass 2 does not and
                // keeps whatever the last sample of pass 1 left in basis_fn
                if (tree.xform != nullptr) rotated_basis<T>(tree, opt, s.idx, r.vdir, basis_fn);
                const T att = exp_T<T>(-s.delta_t * s.sigma * r.delta_scale);
                const T weight = light_intensity * (1.f - att);
                T total_color = 0.f;
                double color_mag = 0.0;
                if (opt.format != FORMAT_RGBA) {
                    for (int c = 0; c < out_data_dim; ++c) {
                        const int off = c * opt.basis_dim;
                        T tmp = 0.0;
                        for (int i = opt.min_comp; i <= opt.max_comp; ++i)
                            tmp += basis_fn[i] * s.row[off + i];
                        const T sigmoid = 1.0 / (1.0 + exp_T<T>(-tmp));
                        const T grad_sigmoid = sigmoid * (1.0 - sigmoid);
                        for (int i = opt.min_comp; i <= opt.max_comp; ++i) {
                            const T toadd = weight * basis_fn[i] * grad_sigmoid * grad_output[c];
                            add(base + off + i, toadd, std::fabs((double)toadd), std::fabs((double)toadd));
                        }
                        total_color += sigmoid * grad_output[c];
                        color_mag += std::fabs((double)sigmoid * (double)grad_output[c]);
                    }
                } else {
                    for (int j = 0; j < out_data_dim; ++j) {
                        const T sigmoid = 1.0 / (1.0 + exp_T<T>(-s.row[j]));
                        const T toadd = weight * sigmoid * (1.f - sigmoid) * grad_output[j];
                        add(base + j, toadd, std::fabs((double)toadd), std::fabs((double)toadd));
                        total_color += sigmoid * grad_output[j];
                        color_mag += std::fabs((double)sigmoid * (double)grad_output[j]);
                    }
                }
                light_intensity *= att;
                accum += weight * total_color;
                accum_scale += std::fabs((double)weight) * color_mag;
                accum_scale_tight += std::fabs((double)weight * (double)total_color);
            }
            t += s.delta_t;
        }
        T total_grad = 0.f;
        for (int j = 0; j < out_data_dim; ++j) total_grad += grad_output[j];
        accum += light_intensity * opt.background_brightness * total_grad;
        double grad_mag = 0.0;
        for (int j = 0; j < out_data_dim; ++j) grad_mag += std::fabs((double)grad_output[j]);
        accum_scale += std::fabs((double)light_intensity * opt.background_brightness) * grad_mag;
        accum_scale_tight += std::fabs((double)light_intensity * opt.background_brightness * (double)total_grad);
        light_intensity_ray = light_intensity;
    }
    {   // PASS 2 (:439-494)
        T light_intensity = 1.f, t = r.tmin;
        while (t < r.tmax) {
            const Step<T> s = march_step<T>(tree, r, opt, t);
            if (s.sigma > 0.0) {
                const int64_t base = (int64_t)s.idx * K;
                const T att = exp_T<T>(-s.delta_t * s.sigma * r.delta_scale);
                const T weight = light_intensity * (1.f - att);
                T total_color = 0.f;
                if (opt.format != FORMAT_RGBA) {
                    for (int c = 0; c < out_data_dim; ++c) {
                        const int off = c * opt.basis_dim;
                        T tmp = 0.0;
                        for (int i = opt.min_comp; i <= opt.max_comp; ++i)
                            tmp += basis_fn[i] * s.row[off + i];
                        total_color += 1.0 / (1.0 + exp_T<T>(-tmp)) * grad_output[c];
                    }
                } else {
                    for (int j = 0; j < out_data_dim; ++j)
                        total_color += 1.0 / (1.0 + exp_T<T>(-s.row[j])) * grad_output[j];
                }
                light_intensity *= att;
                accum -= weight * total_color;
                const T toadd = s.delta_t * r.delta_scale * (total_color * light_intensity - accum)
                              + s.delta_t * r.delta_scale * grad_output[out_data_dim] * light_intensity_ray;
                // the sigma term is delta times a difference of O(1) quantities;
                // `accum` itself is a running sum that was built up and is being
                // taken down again, so its rounding error scales with the sum of
                // its addends (accum_scale), not with its current value
                const double dd = std::fabs((double)s.delta_t * (double)r.delta_scale);
                const double rest = std::fabs((double)total_color * (double)light_intensity)
                                  + std::fabs((double)grad_output[out_data_dim] * (double)light_intensity_ray);
                add(base + K - 1, toadd, dd * (rest + accum_scale), dd * (rest + accum_scale_tight));
            }
            t += s.delta_t;
        }
    }
}

// rt_kernel.cu:500-560
template <typename T>
void opacity_trace_ray(const Tree<T>& tree, const RaySetup<T>& r, const RenderOptions& opt, T* out) {
    out[0] = 0.f;
    if (!r.hit) return;
    T light_intensity = 1.f;
    T t = r.tmin;
    while (t < r.tmax) {
        const Step<T> s = march_step<T>(tree, r, opt, t);
        if (s.sigma > opt.sigma_thresh) {
            const T att = exp_T<T>(-s.delta_t * r.delta_scale * s.sigma);
            light_intensity *= att;
            if (light_intensity <= opt.stop_thresh) {
                out[0] = 1 - light_intensity;
                return;
            }
        }
        t += s.delta_t;
    }
    out[0] = 1 - light_intensity;
}

// rt_kernel.cu:782-834
template <typename T>
void depth_trace_ray(const Tree<T>& tree, const RaySetup<T>& r, const RenderOptions& opt,
                     T* depth_out, Counters* cnt) {
    depth_out[0] = 0.f;
    if (!r.hit) return;
    if (cnt) cnt->rays_hit++;
    T t = r.tmin;
    while (t < r.tmax) {
        const Step<T> s = march_step<T>(tree, r, opt, t);
        if (cnt) { cnt->steps++; cnt->levels += s.levels; cnt->valid += s.row != nullptr; }
        if (s.sigma > opt.sigma_thresh) {
            depth_out[0] = r.delta_scale * t;
            return;
        }
        t += s.delta_t;
    }
}

// ---------------------------------------------------------------------------
// Motion variants
// ---------------------------------------------------------------------------

// One leaf crossing that also hands back the leaf-local coordinates the
// reference's in/out `pos` holds after query_single_from_root.
template <typename T>
inline Step<T> march_step_local(const Tree<T>& tree, const RaySetup<T>& r, const RenderOptions& opt, T t, T* pos) {
    Step<T> s;
    for (int j = 0; j < 3; ++j) pos[j] = r.origin[j] + t * r.dir[j];
    T cube_sz;
    s.slot = query_from_root<T>(tree, pos, &cube_sz, &s.levels);
    s.idx = tree.data[s.slot];
    s.row = (s.idx < 0 || (int64_t)s.idx >= tree.M) ? nullptr : tree.features + (int64_t)s.idx * tree.K;
    T sub_tmin, sub_tmax;
    dda_unit<T>(pos, r.invdir, &sub_tmin, &sub_tmax);
    s.delta_t = (sub_tmax - sub_tmin) / cube_sz + opt.step_size;
    s.sigma = s.row != nullptr ? s.row[tree.K - 1] : 0.0;
    return s;
}

// motion_trace_ray (rt_kernel.cu:698-778): first sample with sigma > sigma_thresh ->
// distances from the "hit point" to the J joint positions in extra_data[:, 0:3], the
// depth, the hit point, the feature row index.  The hit point is
// transform_coord_world (common.cuh:54-60) of `pos`, which at that line holds the
// LEAF-LOCAL coordinates (query_single_from_root rewrote it, :745) -- reproduced as is.
// All outputs are zero when nothing is hit (torch::zeros, :1489-1492).
template <typename T>
void motion_trace_ray(const Tree<T>& tree, const RaySetup<T>& r, const RenderOptions& opt, int J,
                      T* out, T* depth_out, T* hit_point_out, int64_t* data_idx_out) {
    for (int j = 0; j < J; ++j) out[j] = 0.f;
    depth_out[0] = 0.f;
    hit_point_out[0] = hit_point_out[1] = hit_point_out[2] = 0.f;
    data_idx_out[0] = 0;
    if (!r.hit) return;
    T t = r.tmin;
    while (t < r.tmax) {
        T pos[3];
        const Step<T> s = march_step_local<T>(tree, r, opt, t, pos);
        if (s.sigma > opt.sigma_thresh) {
            for (int i = 0; i < 3; ++i) pos[i] = (pos[i] - tree.offset[i]) / tree.scaling[i];
            for (int i = 0; i < 3; ++i) hit_point_out[i] = pos[i];
            depth_out[0] = t * r.delta_scale;
            for (int i = 0; i < J; ++i) {
                T dis[3];
                for (int k = 0; k < 3; ++k) dis[k] = pos[k] - tree.extra[(int64_t)i * tree.extra_cols + k];
                out[i] = norm3<T>(dis);
            }
            data_idx_out[0] = s.idx;
            return;
        }
        t += s.delta_t;
    }
}

template <typename T>
struct Motion {
    const T* joint_features;       // [n_joints, F]
    int n_joints, F;
    const T* skinning_weights;     // [M, B]
    const int32_t* joint_index;    // [M, B]
    int B;
};

// pos_joint_feature (rt_kernel.cu:946-952)
template <typename T>
inline void blend_joint_features(const Motion<T>& mo, int32_t idx, T* pjf) {
    for (int k = 0; k < mo.F; ++k) pjf[k] = 0.f;
    const T* sw = mo.skinning_weights + (int64_t)idx * mo.B;
    const int32_t* ji = mo.joint_index + (int64_t)idx * mo.B;
    for (int j = 0; j < mo.B; ++j)
        if (sw[j] > 0)
            for (int k = 0; k < mo.F; ++k) pjf[k] += sw[j] * mo.joint_features[(int64_t)ji[j] * mo.F + k];
}

// motion_feature_trace_ray (rt_kernel.cu:886-981)
template <typename T>
void motion_feature_trace_ray(const Tree<T>& tree, const Motion<T>& mo, const RaySetup<T>& r,
                              const RenderOptions& opt, T* out) {
    const int F = mo.F;
    for (int j = 0; j < F; ++j) out[j] = 0.f;
    if (!r.hit) return;                                   // :913-919: zeros, not the background
    T light_intensity = 1.f;
    T t = r.tmin;
    T pjf[32];
    while (t < r.tmax) {
        const Step<T> s = march_step<T>(tree, r, opt, t);
        if (s.sigma > opt.sigma_thresh) {
            const T att = exp_T<T>(-s.delta_t * r.delta_scale * s.sigma);
            const T weight = light_intensity * (1.f - att);
            blend_joint_features<T>(mo, s.idx, pjf);
            for (int j = 0; j < F; ++j) out[j] += weight / (1.0 + exp_T<T>(-pjf[j]));
            light_intensity *= att;
            if (light_intensity <= opt.stop_thresh) {
                T scale = 1.0 / (1.0 - light_intensity);
                for (int j = 0; j != F; ++j) out[j] *= scale;
                return;
            }
        }
        t += s.delta_t;
    }
    for (int j = 0; j < F; ++j) out[j] += light_intensity * opt.background_brightness;
}

// Gradient of the above wrt joint_features, as motion_feature_trace_ray_backward
// (rt_kernel.cu:983-1061) sets out to compute: per sample with sigma > 0,
//   toadd_k = weight * sigmoid_k * (1 - sigmoid_k) * grad_output_k,
//   grad[joint_index_j][k] += skinning_weight_j * toadd_k.
// The reference's own loop adds into an uninitialised local and indexes it by bone
// instead of channel (:1043, :1048), so its result is undefined;
// this is the derivative of the forward.  Exponent associated as the reference's
// backward does (:1031); thresholds ignored as there.  `grad` / `abs_sum` double.
template <typename T>
void motion_feature_trace_ray_backward(const Tree<T>& tree, const Motion<T>& mo, const RaySetup<T>& r,
                                       const RenderOptions& opt, const T* grad_output,
                                       double* grad, double* abs_sum, bool atomic) {
    if (!r.hit) return;
    const int F = mo.F;
    T light_intensity = 1.f, t = r.tmin;
    T pjf[32];
    while (t < r.tmax) {
        const Step<T> s = march_step<T>(tree, r, opt, t);
        if (s.sigma > 0.0) {
            const T att = exp_T<T>(-s.delta_t * s.sigma * r.delta_scale);
            const T weight = light_intensity * (1.f - att);
            blend_joint_features<T>(mo, s.idx, pjf);
            const T* sw = mo.skinning_weights + (int64_t)s.idx * mo.B;
            const int32_t* ji = mo.joint_index + (int64_t)s.idx * mo.B;
            for (int k = 0; k < F; ++k) {
                const T sigmoid = 1.0 / (1.0 + exp_T<T>(-pjf[k]));
                const T toadd = weight * sigmoid * (1.f - sigmoid) * grad_output[k];
                for (int j = 0; j < mo.B; ++j) {
                    if (!(sw[j] > 0)) continue;
                    const T v = sw[j] * toadd;
                    double* g = grad + (int64_t)ji[j] * F + k;
                    if (atomic) {
#pragma omp atomic
                        *g += (double)v;
                    } else {
                        *g += (double)v;
                    }
                    if (abs_sum) {
                        double* a = abs_sum + (int64_t)ji[j] * F + k;
                        const double av = std::fabs((double)v);
                        if (atomic) {
#pragma omp atomic
                            *a += av;
                        } else {
                            *a += av;
                        }
                    }
                }
            }
            light_intensity *= att;
        }
        t += s.delta_t;
    }
}

// transformation_matrices for the next f32 render / backward calls (test hook)
static const float* g_xform_f32 = nullptr;
static int g_xform_dim = 3;
template <typename T> inline const T* current_xform() { return nullptr; }
template <> inline const float* current_xform<float>() { return g_xform_f32; }

template <typename T>
Tree<T> make_tree(const T* features, int64_t M, int K, const int32_t* data,
                  const int32_t* child, int N, const T* offset, const T* scaling,
                  const T* extra, int extra_rows, int extra_cols) {
    Tree<T> t;
    t.features = features; t.M = M; t.K = K; t.data = data; t.child = child; t.N = N;
    for (int i = 0; i < 3; ++i) { t.offset[i] = offset[i]; t.scaling[i] = scaling[i]; }
    t.extra = extra; t.extra_rows = extra_rows; t.extra_cols = extra_cols;
    t.xform = current_xform<T>();
    t.xform_dim = g_xform_dim;
    return t;
}

// rt_kernel.cu:1352-1358
inline int get_out_data_dim(int format, int basis_dim, int in_data_dim) {
    if (format != FORMAT_RGBA) return (in_data_dim - 1) / basis_dim + 1;
    return in_data_dim;
}

template <typename T>
void render_impl(const T* features, int64_t M, int K, const int32_t* data, const int32_t* child,
                 int N, const T* offset, const T* scaling, const T* extra, int er, int ec,
                 const T* origins, const T* dirs, const T* vdirs, int64_t Q,
                 const RenderOptions* opt, T* out, int64_t* counters5, double* weight_accum = nullptr) {
    const Tree<T> tree = make_tree<T>(features, M, K, data, child, N, offset, scaling, extra, er, ec);
    const int od = get_out_data_dim(opt->format, opt->basis_dim, K);   // = C + 1
    int64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : c0, c1, c2, c3, c4)
    for (int64_t q = 0; q < Q; ++q) {
        Counters cnt = {0, 0, 0, 0, 0};
        const RaySetup<T> r = setup_ray<T>(tree, origins + 3 * q, dirs + 3 * q, vdirs + 3 * q);
        trace_ray<T>(tree, r, *opt, out + q * od, od - 1, counters5 ? &cnt : nullptr, weight_accum);
        c0 += cnt.rays_hit; c1 += cnt.steps; c2 += cnt.levels; c3 += cnt.valid; c4 += cnt.active;
    }
    if (counters5) { counters5[0] = c0; counters5[1] = c1; counters5[2] = c2; counters5[3] = c3; counters5[4] = c4; }
}

template <typename T>
void render_backward_impl(const T* features, int64_t M, int K, const int32_t* data, const int32_t* child,
                          int N, const T* offset, const T* scaling, const T* extra, int er, int ec,
                          const T* origins, const T* dirs, const T* vdirs, int64_t Q,
                          const RenderOptions* opt, const T* grad_output, int grad_cols,
                          double* grad, double* abs_sum, double* abs_sum_tight = nullptr) {
    const Tree<T> tree = make_tree<T>(features, M, K, data, child, N, offset, scaling, extra, er, ec);
    std::memset(grad, 0, sizeof(double) * (size_t)M * K);
    if (abs_sum) std::memset(abs_sum, 0, sizeof(double) * (size_t)M * K);
    if (abs_sum_tight) std::memset(abs_sum_tight, 0, sizeof(double) * (size_t)M * K);
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    const bool atomic = nthreads > 1;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t q = 0; q < Q; ++q) {
        const RaySetup<T> r = setup_ray<T>(tree, origins + 3 * q, dirs + 3 * q, vdirs + 3 * q);
        trace_ray_backward<T>(tree, r, *opt, grad_output + q * grad_cols, grad_cols - 1,
                              grad, abs_sum, atomic, abs_sum_tight);
    }
}

}  // namespace

extern "C" {

int svoxt_oracle_sizeof_options(void) { return (int)sizeof(RenderOptions); }

void svoxt_oracle_use_libm_exp(int on) { g_use_libm_exp = on != 0; }

// [M, dim, dim] float (dim 3 or 4) per-leaf view rotation used by the f32 volume_render /
// volume_render_backward entry points until reset with NULL.
void svoxt_oracle_set_transformation_matrices(const float* xform, int dim) {
    g_xform_f32 = xform;
    g_xform_dim = dim == 4 ? 4 : 3;
}

// exp for a batch (pins portable_expf against libm and against the GPU's pexpf)
void svoxt_oracle_expf(const float* x, int64_t n, float* y) {
    for (int64_t i = 0; i < n; ++i) y[i] = exp_T<float>(x[i]);
}

int svoxt_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void svoxt_oracle_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

// cam2world_ray (rt_kernel.cu:1153-1166) + maybe_world2ndc (:1170-1190) as
// render_image_kernel (:1193-1211) applies them, for every pixel of a W x H image,
// row-major.  scalar_t = float; the double sub-expressions are the reference's
// (`0.5 * cam.width`, `+ 1.0`).  c2w: row-major, 4 floats per row, first 3 rows used.
// vdirs = the directions before the NDC warp (:1203).
void svoxt_oracle_camera_rays(const float* c2w, float fx, float fy, int W, int H,
                              int ndc_width, int ndc_height, float ndc_focal,
                              float* origins, float* dirs, float* vdirs) {
    for (int iy = 0; iy < H; ++iy)
        for (int ix = 0; ix < W; ++ix) {
            float x = (float)((ix - 0.5 * W) / fx);
            float y = (float)(-(iy - 0.5 * H) / fy);
            float z = sqrtf((float)(x * x + y * y + 1.0));
            x /= z; y /= z; z = -1.0f / z;
            float dir[3], cen[3];
            for (int i = 0; i < 3; ++i) {
                dir[i] = c2w[4 * i + 0] * x + c2w[4 * i + 1] * y + c2w[4 * i + 2] * z;
                cen[i] = c2w[4 * i + 3];
            }
            float* vo = vdirs + 3 * ((int64_t)iy * W + ix);
            vo[0] = dir[0]; vo[1] = dir[1]; vo[2] = dir[2];
            if (ndc_width >= 0) {                       // `if (opt.ndc_width < 0) return;`
                const float near = 1.f;
                const float t = -(near + cen[2]) / dir[2];
                for (int i = 0; i < 3; ++i) cen[i] = cen[i] + t * dir[i];
                dir[0] = -((2 * ndc_focal) / ndc_width) * (dir[0] / dir[2] - cen[0] / cen[2]);
                dir[1] = -((2 * ndc_focal) / ndc_height) * (dir[1] / dir[2] - cen[1] / cen[2]);
                dir[2] = -2 * near / cen[2];
                cen[0] = -((2 * ndc_focal) / ndc_width) * (cen[0] / cen[2]);
                cen[1] = -((2 * ndc_focal) / ndc_height) * (cen[1] / cen[2]);
                cen[2] = 1 + 2 * near / cen[2];
                const float norm = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
                dir[0] /= norm; dir[1] /= norm; dir[2] /= norm;
            }
            float* oo = origins + 3 * ((int64_t)iy * W + ix);
            float* od = dirs + 3 * ((int64_t)iy * W + ix);
            for (int i = 0; i < 3; ++i) { oo[i] = cen[i]; od[i] = dir[i]; }
        }
}

// motion_render (rt_kernel.cu:1480-1504).  out [Q, J], depth [Q], hit_point [Q, 3], data_idx [Q].
void svoxt_oracle_motion_render_f32(
    const float* features, int64_t M, int K, const int32_t* data, const int32_t* child, int N,
    const float* offset, const float* scaling, const float* extra, int er, int ec,
    const float* origins, const float* dirs, const float* vdirs, int64_t Q,
    const RenderOptions* opt, float* out, float* depth, float* hit_point, int64_t* data_idx) {
    const Tree<float> tree = make_tree<float>(features, M, K, data, child, N, offset, scaling, extra, er, ec);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t q = 0; q < Q; ++q) {
        const RaySetup<float> r = setup_ray<float>(tree, origins + 3 * q, dirs + 3 * q, vdirs + 3 * q);
        motion_trace_ray<float>(tree, r, *opt, er, out + q * er, depth + q, hit_point + 3 * q, data_idx + q);
    }
}

#define SVOXT_ORACLE_MOTION_FEATURE(SUFFIX, T)                                                                  \
    void svoxt_oracle_motion_feature_render_##SUFFIX(                                                           \
        const T* features, int64_t M, int K, const int32_t* data, const int32_t* child, int N,                  \
        const T* offset, const T* scaling, const T* joint_features, int n_joints, int F,                        \
        const T* skinning_weights, const int32_t* joint_index, int B,                                           \
        const T* origins, const T* dirs, const T* vdirs, int64_t Q, const RenderOptions* opt, T* out) {         \
        const Tree<T> tree = make_tree<T>(features, M, K, data, child, N, offset, scaling, nullptr, 0, 0);      \
        const Motion<T> mo = {joint_features, n_joints, F, skinning_weights, joint_index, B};                   \
        _Pragma("omp parallel for schedule(dynamic, 256)")                                                      \
        for (int64_t q = 0; q < Q; ++q) {                                                                       \
            const RaySetup<T> r = setup_ray<T>(tree, origins + 3 * q, dirs + 3 * q, vdirs + 3 * q);             \
            motion_feature_trace_ray<T>(tree, mo, r, *opt, out + q * F);                                        \
        }                                                                                                       \
    }                                                                                                           \
    void svoxt_oracle_motion_feature_render_backward_##SUFFIX(                                                  \
        const T* features, int64_t M, int K, const int32_t* data, const int32_t* child, int N,                  \
        const T* offset, const T* scaling, const T* joint_features, int n_joints, int F,                        \
        const T* skinning_weights, const int32_t* joint_index, int B,                                           \
        const T* origins, const T* dirs, const T* vdirs, int64_t Q, const RenderOptions* opt,                   \
        const T* grad_output, double* grad, double* abs_sum) {                                                  \
        const Tree<T> tree = make_tree<T>(features, M, K, data, child, N, offset, scaling, nullptr, 0, 0);      \
        const Motion<T> mo = {joint_features, n_joints, F, skinning_weights, joint_index, B};                   \
        std::memset(grad, 0, sizeof(double) * (size_t)n_joints * F);                                            \
        if (abs_sum) std::memset(abs_sum, 0, sizeof(double) * (size_t)n_joints * F);                            \
        _Pragma("omp parallel for schedule(dynamic, 256)")                                                      \
        for (int64_t q = 0; q < Q; ++q) {                                                                       \
            const RaySetup<T> r = setup_ray<T>(tree, origins + 3 * q, dirs + 3 * q, vdirs + 3 * q);             \
            motion_feature_trace_ray_backward<T>(tree, mo, r, *opt, grad_output + q * F, grad, abs_sum, true);  \
        }                                                                                                       \
    }
SVOXT_ORACLE_MOTION_FEATURE(f32, float)
SVOXT_ORACLE_MOTION_FEATURE(f64, double)

// Per-ray leaf crossings and composited samples of volume_render's march (analysis
// aid for scheduling studies; the sums are what counters5 reports).
void svoxt_oracle_ray_steps_f32(
    const float* features, int64_t M, int K, const int32_t* data, const int32_t* child, int N,
    const float* offset, const float* scaling,
    const float* origins, const float* dirs, const float* vdirs, int64_t Q,
    const RenderOptions* opt, int32_t* steps, int32_t* active) {
    const Tree<float> tree = make_tree<float>(features, M, K, data, child, N, offset, scaling, nullptr, 0, 0);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t q = 0; q < Q; ++q) {
        const RaySetup<float> r = setup_ray<float>(tree, origins + 3 * q, dirs + 3 * q, vdirs + 3 * q);
        int32_t ns = 0, na = 0;
        if (r.hit) {
            float light = 1.f, t = r.tmin;
            while (t < r.tmax) {
                const Step<float> s = march_step<float>(tree, r, *opt, t);
                ++ns;
                if (s.sigma > opt->sigma_thresh) {
                    ++na;
                    light *= exp_T<float>(-s.delta_t * r.delta_scale * s.sigma);
                    if (light <= opt->stop_thresh) break;
                }
                t += s.delta_t;
            }
        }
        steps[q] = ns;
        active[q] = na;
    }
}

// volume_render (rt_kernel.cu:1362-1379).  out is [Q, C+1].
void svoxt_oracle_volume_render_f32(
    const float* features, int64_t M, int K, const int32_t* data, const int32_t* child, int N,
    const float* offset, const float* scaling, const float* extra, int er, int ec,
    const float* origins, const float* dirs, const float* vdirs, int64_t Q,
    const RenderOptions* opt, float* out, int64_t* counters5) {
    render_impl<float>(features, M, K, data, child, N, offset, scaling, extra, er, ec,
                       origins, dirs, vdirs, Q, opt, out, counters5);
}

// ... with tree._weight_accum set (svox.py:948-969): weight_accum [n_internal * N^3], zeroed by the caller
void svoxt_oracle_volume_render_weights_f32(
    const float* features, int64_t M, int K, const int32_t* data, const int32_t* child, int N,
    const float* offset, const float* scaling, const float* extra, int er, int ec,
    const float* origins, const float* dirs, const float* vdirs, int64_t Q,
    const RenderOptions* opt, float* out, double* weight_accum) {
    render_impl<float>(features, M, K, data, child, N, offset, scaling, extra, er, ec,
                       origins, dirs, vdirs, Q, opt, out, nullptr, weight_accum);
}

void svoxt_oracle_volume_render_f64(
    const double* features, int64_t M, int K, const int32_t* data, const int32_t* child, int N,
    const double* offset, const double* scaling, const double* extra, int er, int ec,
    const double* origins, const double* dirs, const double* vdirs, int64_t Q,
    const RenderOptions* opt, double* out, int64_t* counters5) {
    render_impl<double>(features, M, K, data, child, N, offset, scaling, extra, er, ec,
                        origins, dirs, vdirs, Q, opt, out, counters5);
}

// volume_render_backward (rt_kernel.cu:1402-1426) and, with grad_cols == 1,
// opacity_render_backward (:1593-1616, which launches the same kernel).
void svoxt_oracle_volume_render_backward_f32(
    const float* features, int64_t M, int K, const int32_t* data, const int32_t* child, int N,
    const float* offset, const float* scaling, const float* extra, int er, int ec,
    const float* origins, const float* dirs, const float* vdirs, int64_t Q,
    const RenderOptions* opt, const float* grad_output, int grad_cols,
    double* grad, double* abs_sum) {
    render_backward_impl<float>(features, M, K, data, child, N, offset, scaling, extra, er, ec,
                                origins, dirs, vdirs, Q, opt, grad_output, grad_cols, grad, abs_sum);
}

void svoxt_oracle_volume_render_backward_f64(
    const double* features, int64_t M, int K, const int32_t* data, const int32_t* child, int N,
    const double* offset, const double* scaling, const double* extra, int er, int ec,
    const double* origins, const double* dirs, const double* vdirs, int64_t Q,
    const RenderOptions* opt, const double* grad_output, int grad_cols,
    double* grad, double* abs_sum) {
    render_backward_impl<double>(features, M, K, data, child, N, offset, scaling, extra, er, ec,
                                 origins, dirs, vdirs, Q, opt, grad_output, grad_cols, grad, abs_sum);
}

// ... with both error scales (see trace_ray_backward): abs_sum and abs_sum_tight, [M, K] each
void svoxt_oracle_volume_render_backward_scales_f32(
    const float* features, int64_t M, int K, const int32_t* data, const int32_t* child, int N,
    const float* offset, const float* scaling, const float* extra, int er, int ec,
    const float* origins, const float* dirs, const float* vdirs, int64_t Q,
    const RenderOptions* opt, const float* grad_output, int grad_cols,
    double* grad, double* abs_sum, double* abs_sum_tight) {
    render_backward_impl<float>(features, M, K, data, child, N, offset, scaling, extra, er, ec,
                                origins, dirs, vdirs, Q, opt, grad_output, grad_cols, grad, abs_sum, abs_sum_tight);
}

// opacity_render (rt_kernel.cu:1574-1591).  out is [Q, 1].
void svoxt_oracle_opacity_render_f32(
    const float* features, int64_t M, int K, const int32_t* data, const int32_t* child, int N,
    const float* offset, const float* scaling,
    const float* origins, const float* dirs, const float* vdirs, int64_t Q,
    const RenderOptions* opt, float* out) {
    const Tree<float> tree = make_tree<float>(features, M, K, data, child, N, offset, scaling, nullptr, 0, 0);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t q = 0; q < Q; ++q) {
        const RaySetup<float> r = setup_ray<float>(tree, origins + 3 * q, dirs + 3 * q, vdirs + 3 * q);
        opacity_trace_ray<float>(tree, r, *opt, out + q);
    }
}

// render_depth (rt_kernel.cu:1506-1523).  out is [Q, 1].
void svoxt_oracle_render_depth_f32(
    const float* features, int64_t M, int K, const int32_t* data, const int32_t* child, int N,
    const float* offset, const float* scaling,
    const float* origins, const float* dirs, const float* vdirs, int64_t Q,
    const RenderOptions* opt, float* out, int64_t* counters5) {
    const Tree<float> tree = make_tree<float>(features, M, K, data, child, N, offset, scaling, nullptr, 0, 0);
    int64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : c0, c1, c2, c3)
    for (int64_t q = 0; q < Q; ++q) {
        Counters cnt = {0, 0, 0, 0, 0};
        const RaySetup<float> r = setup_ray<float>(tree, origins + 3 * q, dirs + 3 * q, vdirs + 3 * q);
        depth_trace_ray<float>(tree, r, *opt, out + q, counters5 ? &cnt : nullptr);
        c0 += cnt.rays_hit; c1 += cnt.steps; c2 += cnt.levels; c3 += cnt.valid;
    }
    if (counters5) { counters5[0] = c0; counters5[1] = c1; counters5[2] = c2; counters5[3] = c3; counters5[4] = 0; }
}

// query_vertical's per-point part (svox_kernel.cu:45-81): nearest-leaf lookup.
// values rows of empty leaves are left untouched by the reference
// (torch::empty, :282); here they are written as zeros and data_ids as -1.
// node_ids is the packed leaf id node*N^3 + u*N^2 + v*N + w (common.cuh:90-93).
void svoxt_oracle_query_f32(
    const float* features, int64_t M, int K, const int32_t* data, const int32_t* child, int N,
    const float* offset, const float* scaling,
    const float* points, int64_t Q,
    float* values, int64_t* node_ids, int64_t* data_ids) {
    const Tree<float> tree = make_tree<float>(features, M, K, data, child, N, offset, scaling, nullptr, 0, 0);
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < Q; ++q) {
        float xyz[3] = {points[3 * q], points[3 * q + 1], points[3 * q + 2]};
        transform_coord<float>(xyz, tree.offset, tree.scaling);
        float cube_sz;
        const int64_t slot = query_from_root<float>(tree, xyz, &cube_sz, nullptr);
        node_ids[q] = slot;
        const int32_t idx = data[slot];
        if (idx < 0 || (int64_t)idx >= M) {
            data_ids[q] = -1;
            for (int i = 0; i < K; ++i) values[q * K + i] = 0.f;
        } else {
            data_ids[q] = idx;
            for (int i = 0; i < K; ++i) values[q * K + i] = features[(int64_t)idx * K + i];
        }
    }
}

// query_vertical_backward (svox_kernel.cu:84-94, :380-402): scatter-add rows.
void svoxt_oracle_query_backward_f32(
    int64_t M, int K, const int32_t* data, const int32_t* child, int N,
    const float* offset, const float* scaling,
    const float* points, int64_t Q, const float* grad_output, double* grad) {
    const Tree<float> tree = make_tree<float>(nullptr, M, K, data, child, N, offset, scaling, nullptr, 0, 0);
    std::memset(grad, 0, sizeof(double) * (size_t)M * K);
    for (int64_t q = 0; q < Q; ++q) {
        float xyz[3] = {points[3 * q], points[3 * q + 1], points[3 * q + 2]};
        transform_coord<float>(xyz, tree.offset, tree.scaling);
        float cube_sz;
        const int64_t slot = query_from_root<float>(tree, xyz, &cube_sz, nullptr);
        const int32_t idx = data[slot];
        if (idx < 0 || (int64_t)idx >= M) continue;
        for (int i = 0; i < K; ++i) grad[(int64_t)idx * K + i] += (double)grad_output[q * K + i];
    }
}

// Per-ray leaf-crossing counts (forward march, thresholds from opt): used to
// study load balance, not for parity.
void svoxt_oracle_steps_per_ray_f32(
    const float* features, int64_t M, int K, const int32_t* data, const int32_t* child, int N,
    const float* offset, const float* scaling,
    const float* origins, const float* dirs, const float* vdirs, int64_t Q,
    const RenderOptions* opt, int32_t* steps, int32_t* active) {
    const Tree<float> tree = make_tree<float>(features, M, K, data, child, N, offset, scaling, nullptr, 0, 0);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t q = 0; q < Q; ++q) {
        const RaySetup<float> r = setup_ray<float>(tree, origins + 3 * q, dirs + 3 * q, vdirs + 3 * q);
        int32_t ns = 0, na = 0;
        if (r.hit) {
            float t = r.tmin;
            while (t < r.tmax) {
                const Step<float> s = march_step<float>(tree, r, *opt, t);
                ++ns;
                na += s.sigma > 0.f;
                t += s.delta_t;
            }
        }
        steps[q] = ns;
        if (active) active[q] = na;
    }
}

// Feature rows of the first S composited samples of every ray (rec[q*S + k], -1
// padded): to study how many gradient rows a wavefront can merge.
void svoxt_oracle_sample_rows_f32(
    const float* features, int64_t M, int K, const int32_t* data, const int32_t* child, int N,
    const float* offset, const float* scaling,
    const float* origins, const float* dirs, const float* vdirs, int64_t Q,
    const RenderOptions* opt, int32_t S, int32_t* rec, float* rec_t) {
    const Tree<float> tree = make_tree<float>(features, M, K, data, child, N, offset, scaling, nullptr, 0, 0);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t q = 0; q < Q; ++q) {
        for (int k = 0; k < S; ++k) rec[q * S + k] = -1;
        const RaySetup<float> r = setup_ray<float>(tree, origins + 3 * q, dirs + 3 * q, vdirs + 3 * q);
        if (!r.hit) continue;
        int n = 0;
        float t = r.tmin;
        while (t < r.tmax) {
            const Step<float> s = march_step<float>(tree, r, *opt, t);
            if (s.sigma > 0.f && n < S) { if (rec_t) rec_t[q * S + n] = t; rec[q * S + n++] = s.idx; }
            t += s.delta_t;
        }
    }
}

// maybe_precalc_basis for a batch of directions (rt_kernel.cu:110-185); used
// to pin the SH polynomials against the reference's own svox_t/sh.py.
void svoxt_oracle_basis_f32(int format, int basis_dim, const float* extra, int er, int ec,
                            const float* dirs, int64_t n, float* out) {
    Tree<float> tree;
    std::memset(&tree, 0, sizeof(tree));
    tree.xform = nullptr;
    tree.extra = extra; tree.extra_rows = er; tree.extra_cols = ec;
    for (int64_t q = 0; q < n; ++q) {
        float b[25] = {0};
        precalc_basis<float>(format, basis_dim, tree, dirs + 3 * q, b);
        for (int i = 0; i < basis_dim; ++i) out[q * basis_dim + i] = b[i];
    }
}

}  // extern "C"
