// Residual add + LayerNorm + operand cast in one pass, for the MASt3R blocks in the fp16-operand ("TF32-class") mode.
//
// croco/models/blocks.py:88-95 / :176-191: every sub-layer is `x = x + f(norm(x))`.  With an fp32 residual stream and fp16
// GEMM operands that is, per sub-layer, an fp32 + fp16 add, a LayerNorm and a cast of its output -- three launches of
// 3-5 us each around GEMMs of 10-20 us at 768 tokens (LayerNorm + casts + adds were 14 % of the frontend's GPU time,
// profiles/r02_frontend_kernel_stats.csv).  Here: x_out = x + delta (fp32), y = LN(x_out) * gamma + beta rounded once to
// fp16 (or kept fp32), one launch, one wave per row, the row held in registers between the statistics and the output
// (two-pass mean / variance, biased variance like torch.nn.functional.layer_norm).
#include "adk_common.hpp"

namespace adk {

typedef _Float16 ln_f16x4 __attribute__((ext_vector_type(4)));

// Full 64-lane sum in EVERY lane: 4 DPP adds inside the 16-lane rows + 2 permlane swaps across them, all in the VALU (the
// __shfl_xor form is six ds_bpermute round trips through the LDS crossbar with an lgkmcnt wait each).
__device__ __forceinline__ float wave_allsum_dpp(float v) {
    v = row16_allreduce_sum(v);
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(sw.x) + __uint_as_float(sw.y);
    sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(sw.x) + __uint_as_float(sw.y);
}

template <int CHUNKS, bool HAS_DELTA, bool Y_F16>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float* __restrict__ x_in, const _Float16* __restrict__ delta,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, int rows, int C, float* __restrict__ x_out,
                                                            void* __restrict__ y_out)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int n4 = C >> 2; // float4 chunks per row; chunk j of this lane is lane + 64 j
    const float4* xr = reinterpret_cast<const float4*>(x_in + (int64_t)row * C);
    float4 v[CHUNKS];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j) {
        const int c4 = lane + 64 * j;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < n4) {
            v[j] = xr[c4];
            if (HAS_DELTA) {
                const ln_f16x4 d = reinterpret_cast<const ln_f16x4*>(delta + (int64_t)row * C)[c4];
                v[j].x += (float)d[0]; v[j].y += (float)d[1]; v[j].z += (float)d[2]; v[j].w += (float)d[3];
                reinterpret_cast<float4*>(x_out + (int64_t)row * C)[c4] = v[j];
            }
            sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
    }
    const float mean = wave_allsum_dpp(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j) {
        if (lane + 64 * j < n4) {
            const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
            sq += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = rsqrtf(wave_allsum_dpp(sq) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j) {
        const int c4 = lane + 64 * j;
        if (c4 < n4) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[c4], bt = reinterpret_cast<const float4*>(beta)[c4];
            const float y0 = (v[j].x - mean) * rstd * g.x + bt.x, y1 = (v[j].y - mean) * rstd * g.y + bt.y;
            const float y2 = (v[j].z - mean) * rstd * g.z + bt.z, y3 = (v[j].w - mean) * rstd * g.w + bt.w;
            if (Y_F16) {
                ln_f16x4 o; o[0] = (_Float16)y0; o[1] = (_Float16)y1; o[2] = (_Float16)y2; o[3] = (_Float16)y3;
                reinterpret_cast<ln_f16x4*>(static_cast<_Float16*>(y_out) + (int64_t)row * C)[c4] = o;
            } else {
                reinterpret_cast<float4*>(static_cast<float*>(y_out) + (int64_t)row * C)[c4] = make_float4(y0, y1, y2, y3);
            }
        }
    }
}

template <int CHUNKS>
static int launch_add_layernorm(const float* x_in, const void* delta, const float* gamma, const float* beta, float eps, int rows,
                                int C, float* x_out, void* y_out, int y_f16, hipStream_t stream)
{
    const dim3 grid((rows + 3) / 4), block(256);
    const _Float16* d = static_cast<const _Float16*>(delta);
#define ADK_LN_LAUNCH(HD, YF) hipLaunchKernelGGL((add_layernorm_kernel<CHUNKS, HD, YF>), grid, block, 0, stream, x_in, d, gamma, beta, eps, rows, C, x_out, y_out)
    if (delta) { if (y_f16) ADK_LN_LAUNCH(true, true); else ADK_LN_LAUNCH(true, false); }
    else { if (y_f16) ADK_LN_LAUNCH(false, true); else ADK_LN_LAUNCH(false, false); }
#undef ADK_LN_LAUNCH
    return (int)hipGetLastError();
}

} // namespace adk

// x_in [rows, C] float32; delta [rows, C] float16 or NULL; gamma / beta [C] float32.  With delta: x_out [rows, C] float32
// receives x_in + delta (x_out may be x_in itself: every element is read and written by the same lane) and the LayerNorm is
// taken of that sum; without: x_out is ignored.  y_out [rows, C]: float16 if y_f16 else float32.  C a multiple of 4, <= 2048.
extern "C" int adk_add_layernorm(const float* x_in, const void* delta, const float* gamma, const float* beta, float eps,
                                 int rows, int C, float* x_out, void* y_out, int y_f16, hipStream_t stream)
{
    if (!x_in || !gamma || !beta || !y_out || rows < 0 || C <= 0 || (C & 3) || C > 2048) return ADK_EINVAL;
    if (delta && !x_out) return ADK_EINVAL;
    if (((uintptr_t)x_in | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y_out | (uintptr_t)x_out) & 15) return ADK_EINVAL;
    if ((uintptr_t)delta & 7) return ADK_EINVAL;
    if (rows == 0) return 0;
    const int chunks = (C / 4 + 63) / 64;
    if (chunks <= 3) return adk::launch_add_layernorm<3>(x_in, delta, gamma, beta, eps, rows, C, x_out, y_out, y_f16, stream);
    if (chunks <= 4) return adk::launch_add_layernorm<4>(x_in, delta, gamma, beta, eps, rows, C, x_out, y_out, y_f16, stream);
    return adk::launch_add_layernorm<8>(x_in, delta, gamma, beta, eps, rows, C, x_out, y_out, y_f16, stream);
}
