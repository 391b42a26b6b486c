// In-place 2-D rotary position embedding for gfx950.
//
// Replaces rope_2d(tokens, positions, base, fwd) of the `curope` extension --
// VSLAM/thirdparty/mast3r/dust3r/croco/models/curope/curope.cpp:49-65, kernels.cu:17-108
// (wrapper curope2d.py:12-39): tokens [B,N,H,D] are rotated in place, the first D/2 channels by
// the token's y position and the last D/2 by its x position; within a half, channel m (< D/4)
// pairs with m + D/4 and turns by angle pos * fwd / base^(m / (D/4)).
//
// The reference uses one block per token with D (=64) threads and an LDS round trip per head.
// Here a thread owns 4 consecutive rotation pairs of one token (two 16 B accesses per head),
// computes its 4 (cos, sin) once and streams over all heads: no LDS, fully coalesced 16 B
// traffic, 8 B of HBM traffic per element (read + write) -- a pure streaming kernel.
#include "adk_common.hpp"
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>

namespace adk {

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    float v[4];
    __device__ __forceinline__ void load(const float* p) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec4<__half> {
    float v[4];
    __device__ __forceinline__ void load(const __half* p) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        const __half2 a = *reinterpret_cast<const __half2*>(&t.x), b = *reinterpret_cast<const __half2*>(&t.y);
        v[0] = __low2float(a); v[1] = __high2float(a); v[2] = __low2float(b); v[3] = __high2float(b);
    }
    __device__ __forceinline__ void store(__half* p) const {
        const __half2 a = __floats2half2_rn(v[0], v[1]), b = __floats2half2_rn(v[2], v[3]);
        uint2 t; t.x = *reinterpret_cast<const unsigned*>(&a); t.y = *reinterpret_cast<const unsigned*>(&b);
        *reinterpret_cast<uint2*>(p) = t;
    }
};

template <> struct Vec4<__hip_bfloat16> {
    float v[4];
    __device__ __forceinline__ void load(const __hip_bfloat16* p) {
        const uint2 t = *reinterpret_cast<const uint2*>(p); // bf16 -> f32 is a 16-bit shift
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
    }
    __device__ __forceinline__ void store(__hip_bfloat16* p) const {
        auto rn = [](float f) { // round-to-nearest-even to bf16 bits
            const unsigned u = __float_as_uint(f);
            return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
        };
        uint2 t; t.x = rn(v[0]) | (rn(v[1]) << 16); t.y = rn(v[2]) | (rn(v[3]) << 16);
        *reinterpret_cast<uint2*>(p) = t;
    }
};

// VEC path: D % 16 == 0.  One thread = (token, slice of heads, half X, group of 4 pair indices).  The heads are
// split over `hsplit` threads so that small inputs (MASt3R: 768 tokens x 16 heads) still fill the chip: with one
// thread per (token, group) the launch had 24 workgroups and took 14.8 us; split over the heads it is bandwidth-bound.
template <typename T>
__global__ __launch_bounds__(256) void rope2d_vec_kernel(T* __restrict__ tokens, const int64_t* __restrict__ pos,
                                                         int64_t n_tokens, int N, int64_t stride_b, int64_t stride_n, int H, int D, float base, float fwd,
                                                         int hsplit)
{
    const int Q = D >> 2, groups = Q >> 2, per_token = 2 * groups;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n_tokens * per_token * hsplit) return;
    const int64_t tok = tid / (per_token * hsplit);
    const int rr = (int)(tid - tok * (per_token * hsplit));
    const int hs = rr / per_token, r = rr - hs * per_token;
    const int X = r / groups, m0 = (r - X * groups) * 4;
    const float p = (float)pos[tok * 2 + X];
    float c[4], s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float inv_freq = fwd / powf(base, (float)(m0 + j) / (float)Q);
        const float f = p * inv_freq;
        c[j] = cosf(f); s[j] = sinf(f);
    }
    const int hper = (H + hsplit - 1) / hsplit, h0 = hs * hper, h1 = min(H, h0 + hper);
    T* tp = tokens + (tok / N) * stride_b + (tok % N) * stride_n + (int64_t)h0 * D + X * (D >> 1) + m0;
    for (int h = h0; h < h1; ++h, tp += D) {
        Vec4<T> u, v, ou, ov;
        u.load(tp); v.load(tp + Q);
#pragma unroll
        for (int j = 0; j < 4; ++j) { ou.v[j] = u.v[j] * c[j] - v.v[j] * s[j]; ov.v[j] = v.v[j] * c[j] + u.v[j] * s[j]; }
        ou.store(tp); ov.store(tp + Q);
    }
}

// The same rotation with the (cos, sin) pairs read from a table.  Every block of the model rotates q and k by the SAME
// positions (24 encoder + 36 decoder rotations per frame), and powf / cosf / sinf with full range reduction are ~95 % of the
// instructions of the kernel above (7.2 us per call at 768 tokens x 32 q/k heads, 2.7 us as a pure streaming kernel):
// rope2d_table_kernel evaluates them once per (positions, D, base) -- the identical fp32 expressions, so results are bit-identical --
// and rope2d_apply_kernel streams.  table [n_tokens][2 (y, x)][D/4] float2 (cos, sin).
__global__ __launch_bounds__(256) void rope2d_table_kernel(const int64_t* __restrict__ pos, int64_t n_tokens, int Q, float base, float fwd,
                                                           float2* __restrict__ table)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n_tokens * 2 * Q) return;
    const int64_t tok = tid / (2 * Q);
    const int r = (int)(tid - tok * 2 * Q), X = r / Q, m = r - X * Q;
    const float inv_freq = fwd / powf(base, (float)m / (float)Q);
    const float f = (float)pos[tok * 2 + X] * inv_freq;
    table[tid] = make_float2(cosf(f), sinf(f));
}

template <typename T>
__global__ __launch_bounds__(256) void rope2d_apply_kernel(T* __restrict__ tokens, const float2* __restrict__ table, int64_t n_tokens, int N,
                                                           int64_t stride_b, int64_t stride_n, int H, int D, int hsplit)
{
    const int Q = D >> 2, groups = Q >> 2, per_token = 2 * groups;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n_tokens * per_token * hsplit) return;
    const int64_t tok = tid / (per_token * hsplit);
    const int rr = (int)(tid - tok * (per_token * hsplit));
    const int hs = rr / per_token, r = rr - hs * per_token;
    const int X = r / groups, m0 = (r - X * groups) * 4;
    const float4* tb = reinterpret_cast<const float4*>(table + (tok * 2 + X) * Q + m0); // 4 (cos, sin) pairs = 32 B
    const float4 t0 = tb[0], t1 = tb[1];
    const float c[4] = {t0.x, t0.z, t1.x, t1.z}, s[4] = {t0.y, t0.w, t1.y, t1.w};
    const int hper = (H + hsplit - 1) / hsplit, h0 = hs * hper, h1 = min(H, h0 + hper);
    T* tp = tokens + (tok / N) * stride_b + (tok % N) * stride_n + (int64_t)h0 * D + X * (D >> 1) + m0;
    for (int h = h0; h < h1; ++h, tp += D) {
        Vec4<T> u, v, ou, ov;
        u.load(tp); v.load(tp + Q);
#pragma unroll
        for (int j = 0; j < 4; ++j) { ou.v[j] = u.v[j] * c[j] - v.v[j] * s[j]; ov.v[j] = v.v[j] * c[j] + u.v[j] * s[j]; }
        ou.store(tp); ov.store(tp + Q);
    }
}

template <typename T>
static int launch_rope_apply(T* tokens, const float2* table, int64_t n_tokens, int N, int64_t stride_b, int64_t stride_n, int H, int D, hipStream_t stream)
{
    if ((D % 16) || ((uintptr_t)tokens & (4 * sizeof(T) - 1)) || (stride_b % 4) || (stride_n % 4) || ((uintptr_t)table & 15)) return ADK_EINVAL;
    const int64_t base_work = n_tokens * 2 * (D / 16);
    int hsplit = 1;
    while (hsplit < H && base_work * hsplit < 256 * 1024) hsplit *= 2;
    if (hsplit > H) hsplit = H;
    hipLaunchKernelGGL((rope2d_apply_kernel<T>), dim3((unsigned)ceil_div(base_work * hsplit, 256)), dim3(256), 0, stream, tokens, table, n_tokens, N,
                       stride_b, stride_n, H, D, hsplit);
    ADK_RETURN_LAST_ERROR();
}

template <typename T> __device__ __forceinline__ float to_f(T x);
template <> __device__ __forceinline__ float to_f<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f<__half>(__half x) { return __half2float(x); }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ __half from_f<__half>(float x) { return __float2half(x); }
template <> __device__ __forceinline__ float to_f<__hip_bfloat16>(__hip_bfloat16 x) { return __bfloat162float(x); }
template <> __device__ __forceinline__ __hip_bfloat16 from_f<__hip_bfloat16>(float x) { return __float2bfloat16(x); }

// Generic path (D % 4 == 0): one thread = (token, half, pair).
template <typename T>
__global__ __launch_bounds__(256) void rope2d_scalar_kernel(T* __restrict__ tokens, const int64_t* __restrict__ pos,
                                                            int64_t n_tokens, int N, int64_t stride_b, int64_t stride_n, int H, int D, float base, float fwd)
{
    const int Q = D >> 2, per_token = 2 * Q;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n_tokens * per_token) return;
    const int64_t tok = tid / per_token;
    const int r = (int)(tid - tok * per_token);
    const int X = r / Q, m = r - X * Q;
    const float f = (float)pos[tok * 2 + X] * (fwd / powf(base, (float)m / (float)Q));
    const float c = cosf(f), s = sinf(f);
    T* tp = tokens + (tok / N) * stride_b + (tok % N) * stride_n + X * (D >> 1) + m;
    for (int h = 0; h < H; ++h, tp += D) {
        const float u = to_f<T>(tp[0]), v = to_f<T>(tp[Q]);
        tp[0] = from_f<T>(u * c - v * s);
        tp[Q] = from_f<T>(v * c + u * s);
    }
}

template <typename T>
static int launch_rope(T* tokens, const int64_t* pos, int64_t n_tokens, int N, int64_t stride_b, int64_t stride_n, int H, int D, float base, float fwd, hipStream_t stream)
{
    const int epv = 4; // elements per 4-wide access
    const bool vec = (D % 16 == 0) && (((uintptr_t)tokens & (4 * sizeof(T) - 1)) == 0) && (stride_b % epv == 0) && (stride_n % epv == 0);
    if (vec) {
        const int64_t base_work = n_tokens * 2 * (D / 16);
        int hsplit = 1; // split the heads until there are >= ~4 waves per SIMD worth of threads
        while (hsplit < H && base_work * hsplit < 256 * 1024) hsplit *= 2;
        if (hsplit > H) hsplit = H;
        const int64_t work = base_work * hsplit;
        hipLaunchKernelGGL((rope2d_vec_kernel<T>), dim3((unsigned)ceil_div(work, 256)), dim3(256), 0, stream, tokens, pos, n_tokens, N, stride_b, stride_n, H, D, base, fwd, hsplit);
    } else {
        const int64_t work = n_tokens * 2 * (D / 4);
        hipLaunchKernelGGL((rope2d_scalar_kernel<T>), dim3((unsigned)ceil_div(work, 256)), dim3(256), 0, stream, tokens, pos, n_tokens, N, stride_b, stride_n, H, D, base, fwd);
    }
    ADK_RETURN_LAST_ERROR();
}

} // namespace adk

// dtype: 0 = float16, 1 = float32, 2 = bfloat16.  tokens [B,N,H,D] with the last two dims dense (stride D, 1) and
// arbitrary element strides for batch / token (the reference passes a transposed qkv view),
// positions [B,N,2] int64 (y, x) contiguous.
extern "C" int adk_rope_2d(void* tokens, const int64_t* positions, int dtype, int B, int N, int64_t stride_b,
                           int64_t stride_n, int H, int D, float base, float fwd, hipStream_t stream)
{
    if (B < 0 || N < 0 || H < 0 || D < 0 || (D & 3)) return ADK_EINVAL;
    const int64_t n_tokens = (int64_t)B * N;
    if (n_tokens == 0 || H == 0 || D == 0) return 0;
    if (!tokens || !positions) return ADK_EINVAL;
    if (dtype == 1) return adk::launch_rope<float>((float*)tokens, positions, n_tokens, N, stride_b, stride_n, H, D, base, fwd, stream);
    if (dtype == 0) return adk::launch_rope<__half>((__half*)tokens, positions, n_tokens, N, stride_b, stride_n, H, D, base, fwd, stream);
    if (dtype == 2) return adk::launch_rope<__hip_bfloat16>((__hip_bfloat16*)tokens, positions, n_tokens, N, stride_b, stride_n, H, D, base, fwd, stream);
    return ADK_EUNSUPPORTED;
}

// table [B*N][2][D/4] float2 (cos, sin) of positions [B,N,2] int64 -- what adk_rope_2d evaluates per call, evaluated once.
extern "C" int adk_rope_2d_table(const int64_t* positions, int64_t n_tokens, int D, float base, float fwd, float* table, hipStream_t stream)
{
    if (n_tokens < 0 || D <= 0 || (D & 3)) return ADK_EINVAL;
    if (n_tokens == 0) return 0;
    if (!positions || !table || ((uintptr_t)table & 15)) return ADK_EINVAL;
    const int64_t work = n_tokens * 2 * (D / 4);
    hipLaunchKernelGGL(adk::rope2d_table_kernel, dim3((unsigned)adk::ceil_div(work, 256)), dim3(256), 0, stream, positions, n_tokens, D / 4, base, fwd,
                       reinterpret_cast<float2*>(table));
    ADK_RETURN_LAST_ERROR();
}

// adk_rope_2d with the trigonometry taken from a table built by adk_rope_2d_table for the same positions / D / base / fwd: bit-identical
// results.  D % 16 == 0, token / batch strides multiples of 4 elements, tokens aligned to 4 elements.
extern "C" int adk_rope_2d_apply(void* tokens, const float* table, int dtype, int B, int N, int64_t stride_b, int64_t stride_n, int H, int D,
                                 hipStream_t stream)
{
    if (B < 0 || N < 0 || H < 0 || D < 0 || (D & 3)) return ADK_EINVAL;
    const int64_t n_tokens = (int64_t)B * N;
    if (n_tokens == 0 || H == 0 || D == 0) return 0;
    if (!tokens || !table) return ADK_EINVAL;
    const float2* tb = reinterpret_cast<const float2*>(table);
    if (dtype == 1) return adk::launch_rope_apply<float>((float*)tokens, tb, n_tokens, N, stride_b, stride_n, H, D, stream);
    if (dtype == 0) return adk::launch_rope_apply<__half>((__half*)tokens, tb, n_tokens, N, stride_b, stride_n, H, D, stream);
    if (dtype == 2) return adk::launch_rope_apply<__hip_bfloat16>((__hip_bfloat16*)tokens, tb, n_tokens, N, stride_b, stride_n, H, D, stream);
    return ADK_EUNSUPPORTED;
}
