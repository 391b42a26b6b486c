// ABI version + a reference streaming-copy kernel used by bench.py to measure
// the achievable HBM bandwidth of the box it runs on (the denominator-sanity
// check BASELINE.md asks for; the roofline `peak` itself stays the 8 TB/s spec).
#include "adk_common.hpp"
#include "artdeco_hip.h"

extern "C" int adk_abi_version(void) { return ADK_ABI_VERSION; }

namespace adk {
// 4 float4 per thread, all loads before the first store, nontemporal both ways, the grid covers the data: the fastest of the shapes
// tried on MI355X (tools/lab/copy_lab.py, 1 GiB): 6.2 TB/s, one float4 per thread 6.1, a grid-stride loop over 2048 blocks (this
// kernel's first form, and the shape adam_multi_kernel had) 4.8.
typedef float copy_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_copy_kernel(copy_f4* __restrict__ dst, const copy_f4* __restrict__ src, int64_t n4)
{
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    copy_f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int64_t i = base + 256 * u; if (i < n4) v[u] = __builtin_nontemporal_load(&src[i]); }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int64_t i = base + 256 * u; if (i < n4) __builtin_nontemporal_store(v[u], &dst[i]); }
}
} // namespace adk

// Copies nbytes (multiple of 16, both pointers 16 B aligned) with float4 loads/stores.
extern "C" int adk_stream_copy(void* dst, const void* src, int64_t nbytes, hipStream_t stream)
{
    if (nbytes < 0 || (nbytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) return ADK_EINVAL;
    if (nbytes == 0) return 0;
    const int64_t n4 = nbytes >> 4;
    if (n4 > ((int64_t)1 << 40)) return ADK_EUNSUPPORTED;
    hipLaunchKernelGGL(adk::stream_copy_kernel, dim3((unsigned)adk::ceil_div(n4, 1024)), dim3(256), 0, stream,
                       (adk::copy_f4*)dst, (const adk::copy_f4*)src, n4);
    ADK_RETURN_LAST_ERROR();
}

// ---- 4x4 inverse (round 5) ----------------------------------------------------------------------------------------------------------------
// run_system.py:194-227 re-reads every mapper keyframe's pose on a SLAM keyframe and inverts THREE 4x4 matrices per keyframe
// (`view_matrix.inverse()`, two `torch.linalg.inv`): on the GPU each is torch's batched LU + solve + a read-back of `info` to raise on a
// singular input -- ~190 us and one host synchronisation per call, 27.6 ms per SLAM keyframe at 48 keyframes (DESIGN finding 48).
// One thread per matrix: Gauss-Jordan with partial pivoting (first largest |entry| of the column, LAPACK's idamax rule) carried in fp64 and
// rounded once, any element strides (the script inverts a transposed view).  A singular matrix (an exactly zero pivot) gives a NaN-filled
// result and info = 1 + the column it failed at instead of an exception: there is no read-back.
namespace adk {
__global__ __launch_bounds__(64) void inv4x4_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, int64_t sb, int64_t sr,
                                                    int64_t sc, int32_t* __restrict__ info, int32_t* __restrict__ singular_count)
{
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    double a[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { a[r][c] = (double)in[i * sb + r * sr + c * sc]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
    int bad = 0;
#pragma unroll
    for (int col = 0; col < 4; ++col) {
        // bring the first largest |a[r][col]|, r >= col, into row col (static indices only: the rows stay in registers)
#pragma unroll
        for (int r = col + 1; r < 4; ++r) {
            const bool sw = fabs(a[r][col]) > fabs(a[col][col]);
#pragma unroll
            for (int c = 0; c < 8; ++c) { const double x = a[col][c], y = a[r][c]; a[col][c] = sw ? y : x; a[r][c] = sw ? x : y; }
        }
        const double piv = a[col][col];
        if (piv == 0.0 || piv != piv) { if (bad == 0) bad = col + 1; continue; }
        const double ip = 1.0 / piv;
#pragma unroll
        for (int c = 0; c < 8; ++c) a[col][c] *= ip;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = a[r][col];
#pragma unroll
            for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
        }
    }
    const float nan = __builtin_nanf("");
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) out[16 * i + 4 * r + c] = bad ? nan : (float)a[r][4 + c];
    if (info) info[i] = bad;
    // the caller's sticky counter may be host-mapped pinned memory (system scope): written only in the singular case, read by the host at its
    // next wait -- the error surfaces there instead of costing every inversion a read-back
    if (bad && singular_count) __hip_atomic_fetch_add(singular_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
} // namespace adk

// in: n matrices, element (r, c) of matrix i at in[i * batch_stride + r * row_stride + c * col_stride] (strides in elements); out [n,4,4] contiguous;
// info [n] or NULL; singular_count (device or host-mapped pinned int32, or NULL) += 1 per singular matrix.
extern "C" int adk_inv4x4(const float* in, float* out, int64_t n, int64_t batch_stride, int64_t row_stride, int64_t col_stride, int32_t* info,
                          int32_t* singular_count, hipStream_t stream)
{
    if (n < 0) return ADK_EINVAL;
    if (n == 0) return 0;
    if (!in || !out) return ADK_EINVAL;
    hipLaunchKernelGGL(adk::inv4x4_kernel, dim3((unsigned)adk::ceil_div(n, (int64_t)64)), dim3(64), 0, stream, in, out, n, batch_stride, row_stride,
                       col_stride, info, singular_count);
    ADK_RETURN_LAST_ERROR();
}
