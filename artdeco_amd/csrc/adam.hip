// Sparse (visibility-gated) and dense Adam updates for gfx950.
//
// Replaces diff_gaussian_rasterization.adamUpdate / adamUpdateBasic
// [UPSTREAM on-the-fly-nvs fork, not vendored]; semantics taken from the call
// sites Reconstruct/scene/optimizers.py:41-57 (dense, python-float lr) and
// :106-161 (row-gated, lr = 0-dim / per-row / per-element device tensor):
//
//     m = b1*m + (1-b1)*g ;  v = b2*v + (1-b2)*g*g ;  p += -lr * m / (sqrt(v) + eps)
//
// with NO bias correction (Taming-3DGS sparse-Adam convention) and rows whose
// `visible` flag is false left completely untouched (param, m and v).
//
// Pure streaming kernel: 16 B read + 12 B write per updated element
// (28 B/element) + 1 B/row of mask.  Each lane owns one float4
// of the flat [N*M] array so a wave moves 1 KiB per instruction per array; a
// float4 whose (up to 4) rows are all invisible issues no loads at all, which
// is where the "sparse" saving comes from.  Compiled with -ffp-contract=off so
// the result is bit-identical to the IEEE fp32 oracle (oracle/adam_oracle.py).
#include "adk_common.hpp"

namespace adk {

enum { LR_SCALAR_PTR = 0, LR_PER_ROW = 1, LR_PER_ELEM = 2, LR_VALUE = 3 };

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float lr, float b1, float b2,
                                          float omb1, float omb2, float eps)
{
    m = b1 * m + omb1 * g;
    v = b2 * v + omb2 * g * g;
    const float step = -lr * m / (sqrtf(v) + eps);
    p += step;
}

template <int LRMODE, bool GATED>
__global__ __launch_bounds__(256) void adam_vec4_kernel(
    float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg,
    float* __restrict__ exp_avg_sq, const uint8_t* __restrict__ visible, const float* __restrict__ lr_ptr,
    float lr_val, float b1, float b2, float eps, int64_t total, int M)
{
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    const int64_t nvec = total >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float lr_s = lr_val;
    if (LRMODE == LR_SCALAR_PTR) lr_s = lr_ptr[0];

    for (int64_t vi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; vi < nvec; vi += stride) {
        const int64_t e0 = vi << 2;
        int64_t row = 0;
        int col = 0;
        bool vis[4] = {true, true, true, true};
        int64_t rows[4] = {0, 0, 0, 0};
        if (GATED || LRMODE == LR_PER_ROW) {
            row = e0 / M;
            col = (int)(e0 - row * M);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rows[j] = row;
                if (GATED) vis[j] = visible[row] != 0;
                if (++col == M) { col = 0; ++row; }
            }
        }
        if (GATED && !(vis[0] | vis[1] | vis[2] | vis[3])) continue;

        float4 p4 = reinterpret_cast<float4*>(param)[vi];
        const float4 g4 = reinterpret_cast<const float4*>(grad)[vi];
        float4 m4 = reinterpret_cast<float4*>(exp_avg)[vi];
        float4 v4 = reinterpret_cast<float4*>(exp_avg_sq)[vi];
        float lr4[4] = {lr_s, lr_s, lr_s, lr_s};
        if (LRMODE == LR_PER_ELEM) {
            const float4 l = reinterpret_cast<const float4*>(lr_ptr)[vi];
            lr4[0] = l.x; lr4[1] = l.y; lr4[2] = l.z; lr4[3] = l.w;
        } else if (LRMODE == LR_PER_ROW) {
#pragma unroll
            for (int j = 0; j < 4; ++j) lr4[j] = lr_ptr[rows[j]];
        }
        if (vis[0]) adam_elem(p4.x, g4.x, m4.x, v4.x, lr4[0], b1, b2, omb1, omb2, eps);
        if (vis[1]) adam_elem(p4.y, g4.y, m4.y, v4.y, lr4[1], b1, b2, omb1, omb2, eps);
        if (vis[2]) adam_elem(p4.z, g4.z, m4.z, v4.z, lr4[2], b1, b2, omb1, omb2, eps);
        if (vis[3]) adam_elem(p4.w, g4.w, m4.w, v4.w, lr4[3], b1, b2, omb1, omb2, eps);
        reinterpret_cast<float4*>(param)[vi] = p4;
        reinterpret_cast<float4*>(exp_avg)[vi] = m4;
        reinterpret_cast<float4*>(exp_avg_sq)[vi] = v4;
    }

    // scalar tail (total % 4 elements) handled by the first few threads of block 0
    if (blockIdx.x == 0 && threadIdx.x < (total & 3)) {
        const int64_t e = (nvec << 2) + threadIdx.x;
        const int64_t r = (GATED || LRMODE == LR_PER_ROW) ? e / M : 0;
        if (!GATED || visible[r]) {
            float lr = lr_s;
            if (LRMODE == LR_PER_ELEM) lr = lr_ptr[e];
            else if (LRMODE == LR_PER_ROW) lr = lr_ptr[r];
            float p = param[e], m = exp_avg[e], v = exp_avg_sq[e];
            adam_elem(p, grad[e], m, v, lr, b1, b2, omb1, omb2, eps);
            param[e] = p; exp_avg[e] = m; exp_avg_sq[e] = v;
        }
    }
}

// Fallback for pointers that are not 16-byte aligned (views into larger tensors).
template <int LRMODE, bool GATED>
__global__ __launch_bounds__(256) void adam_scalar_kernel(
    float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg,
    float* __restrict__ exp_avg_sq, const uint8_t* __restrict__ visible, const float* __restrict__ lr_ptr,
    float lr_val, float b1, float b2, float eps, int64_t total, int M)
{
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float lr_s = lr_val;
    if (LRMODE == LR_SCALAR_PTR) lr_s = lr_ptr[0];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t r = (GATED || LRMODE == LR_PER_ROW) ? e / M : 0;
        if (GATED && !visible[r]) continue;
        float lr = lr_s;
        if (LRMODE == LR_PER_ELEM) lr = lr_ptr[e];
        else if (LRMODE == LR_PER_ROW) lr = lr_ptr[r];
        float p = param[e], m = exp_avg[e], v = exp_avg_sq[e];
        adam_elem(p, grad[e], m, v, lr, b1, b2, omb1, omb2, eps);
        param[e] = p; exp_avg[e] = m; exp_avg_sq[e] = v;
    }
}

template <int LRMODE, bool GATED>
static int launch_adam(float* param, const float* grad, float* m, float* v, const uint8_t* visible,
                       const float* lr_ptr, float lr_val, float b1, float b2, float eps, int64_t total, int M,
                       hipStream_t stream)
{
    const uintptr_t a = (uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v |
                        (LRMODE == LR_PER_ELEM ? (uintptr_t)lr_ptr : 0);
    if ((a & 15) == 0) {
        const int grid = stream_grid(ceil_div(total, 4), 256);
        hipLaunchKernelGGL((adam_vec4_kernel<LRMODE, GATED>), dim3(grid), dim3(256), 0, stream, param, grad, m, v,
                           visible, lr_ptr, lr_val, b1, b2, eps, total, M);
    } else {
        const int grid = stream_grid(total, 256);
        hipLaunchKernelGGL((adam_scalar_kernel<LRMODE, GATED>), dim3(grid), dim3(256), 0, stream, param, grad, m, v,
                           visible, lr_ptr, lr_val, b1, b2, eps, total, M);
    }
    ADK_RETURN_LAST_ERROR();
}

} // namespace adk

// lr_numel selects the broadcast rule: 1 -> lr[0]; N -> lr[row]; N*M -> lr[row*M+col].
extern "C" int adk_adam_update(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                               const uint8_t* visible, const float* lr, int64_t lr_numel, float b1, float b2,
                               float eps, int64_t N, int64_t M, hipStream_t stream)
{
    if (N < 0 || M < 0) return ADK_EINVAL;
    const int64_t total = N * M;
    if (total == 0) return 0;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !visible || !lr) return ADK_EINVAL;
    if (M > 0x7fffffff) return ADK_EUNSUPPORTED;
    using namespace adk;
    if (lr_numel == 1)
        return launch_adam<LR_SCALAR_PTR, true>(param, grad, exp_avg, exp_avg_sq, visible, lr, 0.f, b1, b2, eps, total, (int)M, stream);
    if (lr_numel == total)
        return launch_adam<LR_PER_ELEM, true>(param, grad, exp_avg, exp_avg_sq, visible, lr, 0.f, b1, b2, eps, total, (int)M, stream);
    if (lr_numel == N)
        return launch_adam<LR_PER_ROW, true>(param, grad, exp_avg, exp_avg_sq, visible, lr, 0.f, b1, b2, eps, total, (int)M, stream);
    return ADK_EINVAL;
}

extern "C" int adk_adam_update_basic(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr,
                                     float b1, float b2, float eps, int64_t numel, hipStream_t stream)
{
    if (numel < 0) return ADK_EINVAL;
    if (numel == 0) return 0;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return ADK_EINVAL;
    using namespace adk;
    return launch_adam<LR_VALUE, false>(param, grad, exp_avg, exp_avg_sq, nullptr, nullptr, lr, b1, b2, eps, numel, 1, stream);
}
