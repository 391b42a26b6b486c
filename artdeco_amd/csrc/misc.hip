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
