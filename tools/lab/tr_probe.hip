// DEV TOOL: what does gfx950's ds_read_b64_tr_b16 return?  LDS holds u16 = its own index; lane l reads at the byte address the
// host passes for it; the 4 results per lane are written out.  tools/lab/tr_probe.py prints the (source lane, element) map.
#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void tr_probe_kernel(const int* addr_bytes, unsigned short* out) {
    __shared__ unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int a = addr_bytes[threadIdx.x];
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)((__attribute__((address_space(3))) char*)lds + a));
    out[threadIdx.x * 4 + 0] = r.x; out[threadIdx.x * 4 + 1] = r.y; out[threadIdx.x * 4 + 2] = r.z; out[threadIdx.x * 4 + 3] = r.w;
}
extern "C" int tr_probe(const int* addr_bytes, unsigned short* out, hipStream_t s) {
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, s, addr_bytes, out);
    return (int)hipGetLastError();
}
