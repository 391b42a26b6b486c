// Per-keyframe exposure correction + clamp, forward and backward, for gfx950.
//
// Replaces the torch ops of SceneModel.render_from_id, Reconstruct/scene/scene_models/h3dgsv3.py:611-614:
//     render = (exposure[:3,:3] @ render.view(3,-1)) + exposure[:3,3,None] ; render = render.clamp(0,1)
// whose forward/backward are [3,3]x[3,P] and [3,P]x[P,3] GEMMs; at P = 2 M pixels hipBLASLt spends
// 3.7 ms per step on the K = P reduction.  Here both directions are one streaming pass over the
// pixels (24 B/px forward, 36 B/px backward) with the 12 exposure gradients reduced DPP-wise per
// wave, in LDS per workgroup, and finished with 12 atomics per workgroup.
#include "adk_common.hpp"

namespace adk {

__global__ __launch_bounds__(256) void exposure_fwd_kernel(const float* __restrict__ E, const float* __restrict__ img,
                                                           int64_t P, float* __restrict__ out)
{
    float e[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) e[i] = E[i];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        const float r = img[p], g = img[P + p], b = img[2 * P + p];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float v = e[4 * i] * r + e[4 * i + 1] * g + e[4 * i + 2] * b + e[4 * i + 3];
            out[i * P + p] = fminf(fmaxf(v, 0.f), 1.f);
        }
    }
}

__global__ __launch_bounds__(256) void exposure_bwd_kernel(const float* __restrict__ E, const float* __restrict__ img,
                                                           const float* __restrict__ v_out, int64_t P,
                                                           float* __restrict__ v_img, float* __restrict__ v_E /*[12], zeroed*/)
{
    __shared__ float red[4][12];
    float e[12], acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { e[i] = E[i]; acc[i] = 0.f; }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        const float c[3] = {img[p], img[P + p], img[2 * P + p]};
        float gi[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float v = e[4 * i] * c[0] + e[4 * i + 1] * c[1] + e[4 * i + 2] * c[2] + e[4 * i + 3];
            gi[i] = (v >= 0.f && v <= 1.f) ? v_out[i * P + p] : 0.f; // clamp passes the gradient on [0,1]
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) v_img[j * P + p] = e[j] * gi[0] + e[4 + j] * gi[1] + e[8 + j] * gi[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            acc[4 * i] += gi[i] * c[0]; acc[4 * i + 1] += gi[i] * c[1]; acc[4 * i + 2] += gi[i] * c[2]; acc[4 * i + 3] += gi[i];
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float s = wave_sum_to_lane63(acc[i]);
        if (lane == 63) red[wv][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (t != 0.f) unsafeAtomicAdd(v_E + threadIdx.x, t);
    }
}

} // namespace adk

// out[3,P] = clamp(E[:3,:3] img + E[:3,3], 0, 1); E [3,4] row-major on the device.
extern "C" int adk_exposure_fwd(const float* E, const float* img, int64_t P, float* out, hipStream_t stream)
{
    if (P < 0) return ADK_EINVAL;
    if (P == 0) return 0;
    if (!E || !img || !out) return ADK_EINVAL;
    hipLaunchKernelGGL(adk::exposure_fwd_kernel, dim3(adk::stream_grid(P, 256)), dim3(256), 0, stream, E, img, P, out);
    ADK_RETURN_LAST_ERROR();
}

// v_E [12] must be zero-filled by the caller.
extern "C" int adk_exposure_bwd(const float* E, const float* img, const float* v_out, int64_t P, float* v_img,
                                float* v_E, hipStream_t stream)
{
    if (P < 0) return ADK_EINVAL;
    if (P == 0) return 0;
    if (!E || !img || !v_out || !v_img || !v_E) return ADK_EINVAL;
    hipLaunchKernelGGL(adk::exposure_bwd_kernel, dim3(adk::stream_grid(P, 256)), dim3(256), 0, stream, E, img, v_out, P, v_img, v_E);
    ADK_RETURN_LAST_ERROR();
}
