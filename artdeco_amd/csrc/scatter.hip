// scatter_max / scatter_min with argument, for gfx950.
//
// Replaces torch_scatter.scatter_max [UPSTREAM pytorch_scatter, CUDA-only wheel, not vendored], imported at module
// level by the hot scene model (Reconstruct/scene/scene_models/h3dgsv3.py:35) and called by update_voxel
// (:289, majority class per voxel: int64 counts grouped by int64 voxel index).  Semantics restated from the
// published behaviour: out[j] = max_{i: index[i]==j} src[i]; groups that receive nothing get 0 and arg = n;
// arg[j] is an i attaining the maximum.  Upstream's CUDA kernel leaves the choice among ties to a store race;
// its CPU path keeps the FIRST one (strict comparison in a sequential loop) -- this kernel always returns the
// first, so it is deterministic and equal to upstream's CPU result.
//
// Three streaming passes (HBM-bound, 8-24 B per element): atomic max into an order-preserving key space,
// atomic min of the positions whose key equals the winner, decode.
#include "adk_common.hpp"

namespace adk {

// order-preserving map of fp32 onto u32 (NaNs sort above +inf; not expected here)
__device__ __forceinline__ uint32_t f32_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float f32_unkey(uint32_t k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

template <typename T> struct Key;
template <> struct Key<float> {
    typedef uint32_t K;
    static __device__ K enc(float v, bool is_min) { const K k = f32_key(v); return is_min ? ~k : k; }
    static __device__ float dec(K k, bool is_min) { return f32_unkey(is_min ? ~k : k); }
};
template <> struct Key<int32_t> {
    typedef uint32_t K;
    static __device__ K enc(int32_t v, bool is_min) { const K k = (uint32_t)v ^ 0x80000000u; return is_min ? ~k : k; }
    static __device__ int32_t dec(K k, bool is_min) { return (int32_t)((is_min ? ~k : k) ^ 0x80000000u); }
};
template <> struct Key<int64_t> {
    typedef unsigned long long K;
    static __device__ K enc(int64_t v, bool is_min) { const K k = (unsigned long long)v ^ 0x8000000000000000ull; return is_min ? ~k : k; }
    static __device__ int64_t dec(K k, bool is_min) { return (int64_t)((is_min ? ~k : k) ^ 0x8000000000000000ull); }
};

template <typename T>
__global__ __launch_bounds__(256) void scatter_init_kernel(int64_t dim_size, int64_t n, typename Key<T>::K* __restrict__ out, int64_t* __restrict__ arg) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < dim_size) { out[j] = 0; arg[j] = n; } // key 0 is below every encoded value except enc(lowest): handled by `touched` via arg
}

template <typename T>
__global__ __launch_bounds__(256) void scatter_max_kernel(int64_t n, const T* __restrict__ src, const int64_t* __restrict__ index,
                                                          int64_t dim_size, bool is_min, typename Key<T>::K* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t j = index[i];
    if (j < 0 || j >= dim_size) return; // out-of-range indices are ignored (upstream: undefined behaviour)
    atomicMax(out + j, Key<T>::enc(src[i], is_min));
}

template <typename T>
__global__ __launch_bounds__(256) void scatter_arg_kernel(int64_t n, const T* __restrict__ src, const int64_t* __restrict__ index,
                                                          int64_t dim_size, bool is_min, const typename Key<T>::K* __restrict__ out,
                                                          int64_t* __restrict__ arg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t j = index[i];
    if (j < 0 || j >= dim_size) return;
    if (Key<T>::enc(src[i], is_min) == out[j]) atomicMin((unsigned long long*)(arg + j), (unsigned long long)i);
}

template <typename T>
__global__ __launch_bounds__(256) void scatter_decode_kernel(int64_t dim_size, int64_t n, bool is_min, typename Key<T>::K* __restrict__ out,
                                                             const int64_t* __restrict__ arg) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= dim_size) return;
    const T v = (arg[j] == n) ? (T)0 : Key<T>::dec(out[j], is_min); // empty group -> 0
    reinterpret_cast<T*>(out)[j] = v;
}

template <typename T>
static int run_scatter(int64_t n, const void* src, const int64_t* index, int64_t dim_size, int is_min, void* out, int64_t* arg, hipStream_t stream) {
    typedef typename Key<T>::K K;
    const unsigned gd = (unsigned)ceil_div(dim_size, (int64_t)256), gn = (unsigned)ceil_div(n, (int64_t)256);
    hipLaunchKernelGGL(scatter_init_kernel<T>, dim3(gd), dim3(256), 0, stream, dim_size, n, (K*)out, arg);
    if (n > 0) {
        hipLaunchKernelGGL(scatter_max_kernel<T>, dim3(gn), dim3(256), 0, stream, n, (const T*)src, index, dim_size, is_min != 0, (K*)out);
        hipLaunchKernelGGL(scatter_arg_kernel<T>, dim3(gn), dim3(256), 0, stream, n, (const T*)src, index, dim_size, is_min != 0, (const K*)out, arg);
    }
    hipLaunchKernelGGL(scatter_decode_kernel<T>, dim3(gd), dim3(256), 0, stream, dim_size, n, is_min != 0, (K*)out, arg);
    ADK_RETURN_LAST_ERROR();
}

} // namespace adk

// 1-D scatter with argument.  dtype: 0 = float32, 1 = int32, 2 = int64 (src and out).  out [dim_size] and
// arg [dim_size] (int64) are fully written: empty groups get out = 0 and arg = n.
extern "C" int adk_scatter_argmax(int64_t n, const void* src, int dtype, const int64_t* index, int64_t dim_size,
                                  int is_min, void* out, int64_t* arg, hipStream_t stream)
{
    if (n < 0 || dim_size < 0) return ADK_EINVAL;
    if (dim_size == 0) return 0;
    if (!out || !arg || (n > 0 && (!src || !index))) return ADK_EINVAL;
    switch (dtype) {
    case 0: return adk::run_scatter<float>(n, src, index, dim_size, is_min, out, arg, stream);
    case 1: return adk::run_scatter<int32_t>(n, src, index, dim_size, is_min, out, arg, stream);
    case 2: return adk::run_scatter<int64_t>(n, src, index, dim_size, is_min, out, arg, stream);
    default: return ADK_EUNSUPPORTED;
    }
}
