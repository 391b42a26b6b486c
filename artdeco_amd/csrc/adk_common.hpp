// Shared helpers for the artdeco_amd HIP kernels (gfx950 / CDNA4 only).
//
// Conventions (see include/artdeco_hip.h):
//   * every entry point is extern "C", takes raw device pointers + sizes + an
//     explicit hipStream_t, never allocates, never synchronises, returns int
//     (0 = ok, otherwise a hipError_t value or a negative ADK_E* code);
//   * wavefront width is 64, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ADK_WAVE 64

#define ADK_EINVAL (-1)     // bad argument (null pointer, negative size, ...)
#define ADK_EWORKSPACE (-2) // caller-provided workspace too small
#define ADK_EUNSUPPORTED (-3)

#define ADK_RETURN_LAST_ERROR()              \
    do {                                     \
        hipError_t e__ = hipGetLastError();  \
        return (int)e__;                     \
    } while (0)

namespace adk {

__host__ __device__ constexpr inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid size for grid-stride streaming kernels: enough waves to cover the chip
// (256 CUs x 8 blocks of 256 threads) without paying for a huge launch.
static inline int stream_grid(int64_t work_items, int block) {
    int64_t g = ceil_div(work_items, block);
    const int64_t cap = 256 * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// XCD-aware remap of a linear block id: the dispatcher places block b on XCD
// b % 8 (observed, speed only).  Give each XCD a contiguous range of logical
// ids so that neighbouring tiles share an L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = b & 7, k = b >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { int t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
}

// ---- DPP cross-lane adds (VALU only: no LDS crossbar, no s_waitcnt) --------------------------------
// dpp_ctrl encodings (GFX9): quad_perm 0x00-0xFF, row_mirror 0x140, row_half_mirror 0x141,
// row_bcast:15 0x142, row_bcast:31 0x143.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(t);
}
// After this every lane of each 16-lane DPP row holds that row's sum (4 VALU instructions).
__device__ __forceinline__ float row16_allreduce_sum(float v) {
    v = dpp_add<0xB1>(v);  // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);  // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v); // row_half_mirror
    v = dpp_add<0x140>(v); // row_mirror
    return v;
}
// Full 64-lane sum, valid in lane 63 only (6 VALU instructions).
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = row16_allreduce_sum(v);
    v = dpp_add<0x142, 0xa>(v); // row_bcast:15 -> rows 1,3
    v = dpp_add<0x143, 0xc>(v); // row_bcast:31 -> rows 2,3
    return v;
}

} // namespace adk
