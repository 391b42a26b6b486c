// DEV TOOL: which streaming-copy shape reaches the HBM ceiling of this box?  (the guide quotes 6.29 TB/s for a float4 copy; the
// product's adk_stream_copy, a 2048-block grid-stride loop, measures 4.9-5.0.)  Variants differ in grid size, loads in flight per
// thread and cache policy; tools/lab/copy_lab.py times them on 1 GiB.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_gridstride(f4* __restrict__ d, const f4* __restrict__ s, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = s[i];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_unrolled(f4* __restrict__ d, const f4* __restrict__ s, int64_t n) {
    // block b owns a contiguous span of 256 * U float4; thread t touches t, t + 256, ... (coalesced), all loads before all stores
    const int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int64_t i = base + 256 * u; if (i < n) v[u] = NT ? __builtin_nontemporal_load(&s[i]) : s[i]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { const int64_t i = base + 256 * u; if (i < n) { if (NT) __builtin_nontemporal_store(v[u], &d[i]); else d[i] = v[u]; } }
}
template <int U>
__global__ __launch_bounds__(256) void copy_gridstride_unrolled(f4* __restrict__ d, const f4* __restrict__ s, int64_t n) {
    const int64_t span = (int64_t)gridDim.x * 256 * U;
    for (int64_t b0 = (int64_t)blockIdx.x * 256 * U; b0 < n; b0 += span) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t i = b0 + threadIdx.x + 256 * u; if (i < n) v[u] = s[i]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t i = b0 + threadIdx.x + 256 * u; if (i < n) d[i] = v[u]; }
    }
}
__global__ __launch_bounds__(256) void read_only(const f4* __restrict__ s, int64_t n, float* out) {
    const int64_t base = (int64_t)blockIdx.x * 256 * 4 + threadIdx.x;
    f4 a = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int64_t i = base + 256 * u; if (i < n) a += s[i]; }
    if (a.x + a.y + a.z + a.w == 123456.789f) out[0] = a.x;
}
__global__ __launch_bounds__(256) void write_only(f4* __restrict__ d, int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * 256 * 4 + threadIdx.x;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int64_t i = base + 256 * u; if (i < n) d[i] = v; }
}
static unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }
extern "C" int copy_lab(int variant, void* d, const void* s, int64_t nbytes, float* scratch, hipStream_t st) {
    const int64_t n = nbytes / 16;
    f4* D = (f4*)d; const f4* S = (const f4*)s;
    switch (variant) {
    case 0: hipLaunchKernelGGL(copy_gridstride, dim3(2048), dim3(256), 0, st, D, S, n); break;
    case 1: hipLaunchKernelGGL(copy_gridstride, dim3(blocks_for(n, 256)), dim3(256), 0, st, D, S, n); break;
    case 2: hipLaunchKernelGGL((copy_unrolled<4, false>), dim3(blocks_for(n, 1024)), dim3(256), 0, st, D, S, n); break;
    case 3: hipLaunchKernelGGL((copy_unrolled<8, false>), dim3(blocks_for(n, 2048)), dim3(256), 0, st, D, S, n); break;
    case 4: hipLaunchKernelGGL((copy_unrolled<4, true>), dim3(blocks_for(n, 1024)), dim3(256), 0, st, D, S, n); break;
    case 5: hipLaunchKernelGGL((copy_unrolled<8, true>), dim3(blocks_for(n, 2048)), dim3(256), 0, st, D, S, n); break;
    case 6: hipLaunchKernelGGL((copy_gridstride_unrolled<4>), dim3(2048), dim3(256), 0, st, D, S, n); break;
    case 7: hipLaunchKernelGGL((copy_gridstride_unrolled<4>), dim3(8192), dim3(256), 0, st, D, S, n); break;
    case 8: hipLaunchKernelGGL(read_only, dim3(blocks_for(n, 1024)), dim3(256), 0, st, S, n, scratch); break;
    case 9: hipLaunchKernelGGL(write_only, dim3(blocks_for(n, 1024)), dim3(256), 0, st, D, n); break;
    case 10: hipLaunchKernelGGL((copy_unrolled<2, false>), dim3(blocks_for(n, 512)), dim3(256), 0, st, D, S, n); break;
    case 11: hipLaunchKernelGGL((copy_unrolled<16, false>), dim3(blocks_for(n, 4096)), dim3(256), 0, st, D, S, n); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}
