// ABI version + a reference streaming-copy kernel used by bench.py to measure
// the achievable HBM bandwidth of the box it runs on (the denominator-sanity
// check BASELINE.md asks for; the roofline `peak` itself stays the 8 TB/s spec).
#include "adk_common.hpp"
#include "artdeco_hip.h"

extern "C" int adk_abi_version(void) { return ADK_ABI_VERSION; }

namespace adk {
__global__ __launch_bounds__(256) void stream_copy_kernel(float4* __restrict__ dst, const float4* __restrict__ src, int64_t n4)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}
} // namespace adk

// Copies nbytes (multiple of 16, both pointers 16 B aligned) with float4 loads/stores.
extern "C" int adk_stream_copy(void* dst, const void* src, int64_t nbytes, hipStream_t stream)
{
    if (nbytes < 0 || (nbytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) return ADK_EINVAL;
    if (nbytes == 0) return 0;
    const int64_t n4 = nbytes >> 4;
    hipLaunchKernelGGL(adk::stream_copy_kernel, dim3(adk::stream_grid(n4, 256)), dim3(256), 0, stream,
                       (float4*)dst, (const float4*)src, n4);
    ADK_RETURN_LAST_ERROR();
}
