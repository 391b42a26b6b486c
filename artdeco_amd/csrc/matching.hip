// MASt3R-SLAM frontend matching kernels for gfx950: iter_proj and refine_matches.
//
// Replaces VSLAM/backend/src/matching_kernels.cu: iter_proj_kernel (:119-275, host :279-316) and
// refine_matches_kernel (:25-81, host :84-116), bound in gn.cpp:84-114 and called from
// VSLAM/utils_matching.py:152-159 and :171-179.
//
// Per-pixel semantics are kept operation for operation, including the reference's mixed
// precision (double literals in `(1.0-du)*dv`, `1.0/r_norm`, `1.0/det`, `lambda*=0.1`), its
// "weights named opposite to the pixels they multiply" pairing (:161-170), the overwritten
// `converged` flag (:263,267) and, for refine, accumulation in the tensor's own scalar type
// (half: product rounded to half, then sum rounded to half) with max_score starting at the
// smallest positive normal (:47) and a strict `>` (:65).  Compiled with -ffp-contract=off; the
// oracle (oracle/matching_oracle.py) uses the same unfused IEEE operations => bit-exact checks.
//
// What is redesigned for CDNA4: the reference launches 16-thread blocks (a quarter of a
// wavefront).  Here a workgroup is 256 threads = 4 full waves over consecutive pixels, n is
// bounds-checked, descriptors are moved with 16 B loads, and refine keeps the query descriptor in
// registers while the (L2-resident) window rows stream through.
#include "adk_common.hpp"
#include <hip/hip_fp16.h>

namespace adk {

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

__global__ __launch_bounds__(256) void iter_proj_kernel(
    const float* __restrict__ rays_img, const float* __restrict__ pts_3d_norm, const float* __restrict__ p_init,
    float* __restrict__ p_new, uint8_t* __restrict__ converged, int n, int h, int w, int max_iter,
    float lambda_init, float cost_thresh)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (i >= n) return;
    const float* img = rays_img + b * (int64_t)h * w * 9;
    const int64_t pn = b * n + i;
    float u = p_init[2 * pn], v = p_init[2 * pn + 1];
    u = clampf(u, 1.f, (float)(w - 2));
    v = clampf(v, 1.f, (float)(h - 2));
    const float t0 = pts_3d_norm[3 * pn], t1 = pts_3d_norm[3 * pn + 1], t2 = pts_3d_norm[3 * pn + 2];
    float lambda = lambda_init;
    bool conv = false;

    for (int it = 0; it < max_iter; ++it) {
        int u11 = (int)floorf(u), v11 = (int)floorf(v);
        float du = u - (float)u11, dv = v - (float)v11;
        float w11 = du * dv;
        float w12 = (float)((1.0 - (double)du) * (double)dv);
        float w21 = (float)((double)du * (1.0 - (double)dv));
        float w22 = (float)((1.0 - (double)du) * (1.0 - (double)dv));
        const float* r11 = img + ((int64_t)(v11 + 1) * w + (u11 + 1)) * 9;
        const float* r12 = img + ((int64_t)(v11 + 1) * w + u11) * 9;
        const float* r21 = img + ((int64_t)v11 * w + (u11 + 1)) * 9;
        const float* r22 = img + ((int64_t)v11 * w + u11) * 9;
        float r[3], gx[3], gy[3], err[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            r[j] = w11 * r11[j] + w12 * r12[j] + w21 * r21[j] + w22 * r22[j];
            gx[j] = w11 * r11[j + 3] + w12 * r12[j + 3] + w21 * r21[j + 3] + w22 * r22[j + 3];
            gy[j] = w11 * r11[j + 6] + w12 * r12[j + 6] + w21 * r21[j + 6] + w22 * r22[j + 6];
        }
        float r_norm = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        float r_norm_inv = (float)(1.0 / (double)r_norm);
#pragma unroll
        for (int j = 0; j < 3; ++j) r[j] *= r_norm_inv;
        err[0] = r[0] - t0; err[1] = r[1] - t1; err[2] = r[2] - t2;
        const float cost = err[0] * err[0] + err[1] * err[1] + err[2] * err[2];

        float A00 = gx[0] * gx[0] + gx[1] * gx[1] + gx[2] * gx[2];
        const float A01 = gx[0] * gy[0] + gx[1] * gy[1] + gx[2] * gy[2];
        float A11 = gy[0] * gy[0] + gy[1] * gy[1] + gy[2] * gy[2];
        const float b0 = -(err[0] * gx[0] + err[1] * gx[1] + err[2] * gx[2]);
        const float b1 = -(err[0] * gy[0] + err[1] * gy[1] + err[2] * gy[2]);
        A00 += lambda;
        A11 += lambda;
        const float det_inv = (float)(1.0 / (double)(A00 * A11 - A01 * A01));
        const float delta_u = det_inv * (A11 * b0 - A01 * b1);
        const float delta_v = det_inv * (-A01 * b0 + A00 * b1);
        float u_new = clampf(u + delta_u, 1.f, (float)(w - 2));
        float v_new = clampf(v + delta_v, 1.f, (float)(h - 2));

        u11 = (int)floorf(u_new); v11 = (int)floorf(v_new);
        du = u_new - (float)u11; dv = v_new - (float)v11;
        w11 = du * dv;
        w12 = (float)((1.0 - (double)du) * (double)dv);
        w21 = (float)((double)du * (1.0 - (double)dv));
        w22 = (float)((1.0 - (double)du) * (1.0 - (double)dv));
        r11 = img + ((int64_t)(v11 + 1) * w + (u11 + 1)) * 9;
        r12 = img + ((int64_t)(v11 + 1) * w + u11) * 9;
        r21 = img + ((int64_t)v11 * w + (u11 + 1)) * 9;
        r22 = img + ((int64_t)v11 * w + u11) * 9;
#pragma unroll
        for (int j = 0; j < 3; ++j) r[j] = w11 * r11[j] + w12 * r12[j] + w21 * r21[j] + w22 * r22[j];
        r_norm = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        r_norm_inv = (float)(1.0 / (double)r_norm);
#pragma unroll
        for (int j = 0; j < 3; ++j) r[j] *= r_norm_inv;
        err[0] = r[0] - t0; err[1] = r[1] - t1; err[2] = r[2] - t2;
        const float new_cost = err[0] * err[0] + err[1] * err[1] + err[2] * err[2];
        if (new_cost < cost) {
            u = u_new; v = v_new;
            lambda = (float)((double)lambda * 0.1);
            conv = new_cost < cost_thresh;
        } else {
            lambda = (float)((double)lambda * 10.0);
            conv = cost < cost_thresh;
        }
    }
    p_new[2 * pn] = u;
    p_new[2 * pn + 1] = v;
    converged[pn] = conv ? 1 : 0;
}

// Scalar-type adapters: accumulate exactly in the tensor's own type.
struct HalfOps {
    using T = __half;
    static __device__ __forceinline__ T zero() { return __float2half(0.f); }
    static __device__ __forceinline__ T min_pos() { return __ushort_as_half((unsigned short)0x0400); } // 2^-14
    static __device__ __forceinline__ T madd(T acc, T a, T b) { return __hadd(acc, __hmul(a, b)); }
    static __device__ __forceinline__ bool gt(T a, T b) { return __hgt(a, b); }
};
struct FloatOps {
    using T = float;
    static __device__ __forceinline__ T zero() { return 0.f; }
    static __device__ __forceinline__ T min_pos() { return 1.17549435e-38f; }
    static __device__ __forceinline__ T madd(T acc, T a, T b) { return acc + a * b; }
    static __device__ __forceinline__ bool gt(T a, T b) { return a > b; }
};

template <class Ops, int FDIM> // FDIM > 0: compile-time descriptor length (register-resident query)
__global__ __launch_bounds__(256) void refine_matches_kernel(
    const typename Ops::T* __restrict__ D11, const typename Ops::T* __restrict__ D21, const int64_t* __restrict__ p1,
    int64_t* __restrict__ p1_new, int n, int h, int w, int fdim_rt, int radius, int dilation_max)
{
    using T = typename Ops::T;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (i >= n) return;
    const int fdim = FDIM > 0 ? FDIM : fdim_rt;
    const int64_t pn = b * n + i;
    const T* q = D21 + pn * fdim;
    const T* img = D11 + b * (int64_t)h * w * fdim;
    long u0 = p1[2 * pn], v0 = p1[2 * pn + 1];
    T qreg[FDIM > 0 ? FDIM : 1];
    if (FDIM > 0) {
#pragma unroll
        for (int k = 0; k < FDIM; ++k) qreg[k] = q[k];
    }
    T max_score = Ops::min_pos();
    long u_new = u0, v_new = v0;
    for (int d = dilation_max; d > 0; --d) {
        const int rd = radius * d;
        const int diam = 2 * rd + 1;
        for (int ii = 0; ii < diam; ii += d) {
            for (int jj = 0; jj < diam; jj += d) {
                const long u = u0 - rd + ii;
                const long v = v0 - rd + jj;
                if (v >= 0 && v < h && u >= 0 && u < w) {
                    const T* c = img + (v * w + u) * fdim;
                    T score = Ops::zero();
                    if (FDIM > 0) {
#pragma unroll
                        for (int k = 0; k < FDIM; ++k) score = Ops::madd(score, qreg[k], c[k]);
                    } else {
                        for (int k = 0; k < fdim; ++k) score = Ops::madd(score, q[k], c[k]);
                    }
                    if (Ops::gt(score, max_score)) { max_score = score; u_new = u; v_new = v; }
                }
            }
        }
        u0 = u_new; v0 = v_new;
    }
    p1_new[2 * pn] = u_new;
    p1_new[2 * pn + 1] = v_new;
}

} // namespace adk

extern "C" int adk_iter_proj(const float* rays_img_with_grad, const float* pts_3d_norm, const float* p_init, int batch,
                             int h, int w, int n, int max_iter, float lambda_init, float cost_thresh, float* p_new,
                             uint8_t* converged, hipStream_t stream)
{
    if (batch < 0 || h < 0 || w < 0 || n < 0 || max_iter < 0) return ADK_EINVAL;
    if (batch == 0 || n == 0) return 0;
    if (h < 3 || w < 3) return ADK_EINVAL; // the clamp to [1, w-2] x [1, h-2] needs a 3x3 image
    if (!rays_img_with_grad || !pts_3d_norm || !p_init || !p_new || !converged) return ADK_EINVAL;
    if (batch > 65535) return ADK_EUNSUPPORTED;
    hipLaunchKernelGGL(adk::iter_proj_kernel, dim3((unsigned)adk::ceil_div(n, 256), batch), dim3(256), 0, stream,
                       rays_img_with_grad, pts_3d_norm, p_init, p_new, converged, n, h, w, max_iter, lambda_init, cost_thresh);
    ADK_RETURN_LAST_ERROR();
}

// dtype: 0 = float16, 1 = float32.
extern "C" int adk_refine_matches(const void* D11, const void* D21, const int64_t* p1, int dtype, int batch, int h,
                                  int w, int n, int fdim, int radius, int dilation_max, int64_t* p1_new,
                                  hipStream_t stream)
{
    if (batch < 0 || h < 0 || w < 0 || n < 0 || fdim < 0 || radius < 0 || dilation_max < 0) return ADK_EINVAL;
    if (batch == 0 || n == 0) return 0;
    if (!D11 || !D21 || !p1 || !p1_new) return ADK_EINVAL;
    if (batch > 65535) return ADK_EUNSUPPORTED;
    const dim3 grid((unsigned)adk::ceil_div(n, 256), batch), block(256);
    if (dtype == 0) {
        if (fdim == 24)
            hipLaunchKernelGGL((adk::refine_matches_kernel<adk::HalfOps, 24>), grid, block, 0, stream, (const __half*)D11, (const __half*)D21, p1, p1_new, n, h, w, fdim, radius, dilation_max);
        else
            hipLaunchKernelGGL((adk::refine_matches_kernel<adk::HalfOps, 0>), grid, block, 0, stream, (const __half*)D11, (const __half*)D21, p1, p1_new, n, h, w, fdim, radius, dilation_max);
    } else if (dtype == 1) {
        if (fdim == 24)
            hipLaunchKernelGGL((adk::refine_matches_kernel<adk::FloatOps, 24>), grid, block, 0, stream, (const float*)D11, (const float*)D21, p1, p1_new, n, h, w, fdim, radius, dilation_max);
        else
            hipLaunchKernelGGL((adk::refine_matches_kernel<adk::FloatOps, 0>), grid, block, 0, stream, (const float*)D11, (const float*)D21, p1, p1_new, n, h, w, fdim, radius, dilation_max);
    } else {
        return ADK_EUNSUPPORTED;
    }
    ADK_RETURN_LAST_ERROR();
}
