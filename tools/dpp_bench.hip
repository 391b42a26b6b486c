#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
    return v + __int_as_float(t);
}
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
// gfx950 only: v_permlane32_swap / v_permlane16_swap exchange half-waves / odd-even rows between TWO registers
template <int W> __device__ __forceinline__ void swap_add(float& x, float& y) {
    u2 r = W == 32 ? __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false)
                   : __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r.x); y = __uint_as_float(r.y);
}
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[10];
    for (int i = 0; i < 10; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            if (MODE == 0) { a[i] = a[i] * 1.0001f + 0.5f; a[i] = a[i] * 0.9999f + 0.25f; a[i] = a[i] * 1.0001f + 0.5f; a[i] = a[i] * 0.9999f + 0.25f; }
            if (MODE == 1) { a[i] = dpp_add<0xB1>(a[i]); a[i] = dpp_add<0x4E>(a[i]); a[i] = dpp_add<0x141>(a[i]); a[i] = dpp_add<0x140>(a[i]); }
            if (MODE == 2) { a[i] = dpp_add<0xB1>(a[i]); a[i] = dpp_add<0x4E>(a[i]); a[i] = dpp_add<0xB1>(a[i]); a[i] = dpp_add<0x4E>(a[i]); }
            if (MODE == 3) { a[i] = dpp_add<0x141>(a[i]); a[i] = dpp_add<0x140>(a[i]); a[i] = dpp_add<0x141>(a[i]); a[i] = dpp_add<0x140>(a[i]); }
            if (MODE == 4) { a[i] += __shfl_xor(a[i], 1); a[i] += __shfl_xor(a[i], 2); a[i] += __shfl_xor(a[i], 4); a[i] += __shfl_xor(a[i], 8); }
            if (MODE == 6 && i < 5) { swap_add<32>(a[i], a[i + 5]); swap_add<32>(a[i], a[i + 5]); swap_add<32>(a[i], a[i + 5]); swap_add<32>(a[i], a[i + 5]);
                                      swap_add<32>(a[i], a[i + 5]); swap_add<32>(a[i], a[i + 5]); swap_add<32>(a[i], a[i + 5]); swap_add<32>(a[i], a[i + 5]); }
            if (MODE == 7 && i < 5) { swap_add<16>(a[i], a[i + 5]); swap_add<16>(a[i], a[i + 5]); swap_add<16>(a[i], a[i + 5]); swap_add<16>(a[i], a[i + 5]);
                                      swap_add<16>(a[i], a[i + 5]); swap_add<16>(a[i], a[i + 5]); swap_add<16>(a[i], a[i + 5]); swap_add<16>(a[i], a[i + 5]); }
            if (MODE == 8 && i < 5) { // what a butterfly stage does: swap, add, swap, add ...
#pragma unroll
                for (int r = 0; r < 4; ++r) { swap_add<32>(a[i], a[i + 5]); a[i] += a[i + 5]; swap_add<16>(a[i], a[i + 5]); a[i + 5] += a[i]; } }
            if (MODE == 5) { a[i] = dpp_add<0x128>(a[i]); a[i] = dpp_add<0x124>(a[i]); a[i] = dpp_add<0x122>(a[i]); a[i] = dpp_add<0x121>(a[i]); } // row_ror 8,4,2,1
        }
    }
    float s = 0; for (int i = 0; i < 10; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name) {
    float* d; hipMalloc(&d, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, d, 2000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winst = 2048.0 * 4 * 2000 * 40; // wave-instructions of interest
    printf("%-28s %.3f ms  -> %.2f cycles per wave-instr per SIMD (at 2.1 GHz)\n", name, ms, ms * 1e-3 * 2.1e9 * 1024 / winst);
    hipFree(d);
}
int main() { run<0>("plain v_fma"); run<1>("dpp quad,quad,halfmirror,mirror"); run<2>("dpp quad_perm only"); run<3>("dpp row mirrors only"); run<4>("shfl_xor (bpermute)"); run<5>("dpp row_ror 8,4,2,1");
    run<6>("v_permlane32_swap"); run<7>("v_permlane16_swap"); run<8>("swap32,add,swap16,add (x2 = instrs)"); return 0; }
