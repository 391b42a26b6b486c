// DEV TOOL: the per-(splat, tile) reduction of raster_bwd_kernel in isolation -- 10 per-lane floats summed over the 64 lanes of a wave.
//   variant 0: adk::wave_reduce10 as shipped (in-row stages first: 15 bank-masked DPP adds, 3 DPP moves, selects, then two
//              v_permlane*_swap for the row stages)
//   variant 1: ROW STAGES FIRST.  v_permlane32_swap(a_i, a_j) + one add folds the 32-lane stage of TWO values at once (10 registers
//              -> 5), v_permlane16_swap the 16-lane stage (5 -> 3), then the four in-row stages on 3 / 2 / 1 / 1 registers with
//              DPP adds: 8 swaps + 8 plain adds + 7 DPP adds.
//              Result: lanes of (row r, quad g) hold  g = 0: {S0, S2, S1, S3}[r]   g = 1, 3: {S8, S8, S9, S9}[r]   g = 2: {S4, S6, S5, S7}[r]
// Each thread block is one wave; it generates `iters` sets of inputs with a couple of cheap VALU ops, reduces each, and accumulates
// what its lane ended up with, so that the reductions cannot be hoisted or dropped.  out[variant][lane] = accumulated value.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../artdeco_amd/csrc/adk_common.hpp"

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float swap32_add(float x, float y) {
    const u32x2 s = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(s.x) + __uint_as_float(s.y);     // lanes 0-31: x.lo + x.hi, lanes 32-63: y.lo + y.hi
}
__device__ __forceinline__ float swap16_add(float x, float y) {
    const u32x2 s = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(s.x) + __uint_as_float(s.y);     // rows: x.r0 + x.r1, y.r0 + y.r1, x.r2 + x.r3, y.r2 + y.r3
}
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_move(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, BANK, false));
}

__device__ __forceinline__ float reduce10_rows_first(const float (&a)[10]) {
    const float p01 = swap32_add(a[0], a[1]), p23 = swap32_add(a[2], a[3]), p45 = swap32_add(a[4], a[5]);
    const float p67 = swap32_add(a[6], a[7]), p89 = swap32_add(a[8], a[9]);
    const float q0 = swap16_add(p01, p23);      // rows: v0 v2 v1 v3
    const float q1 = swap16_add(p45, p67);      // rows: v4 v6 v5 v7
    const float q2 = swap16_add(p89, p89);      // rows: v8 v8 v9 v9
    // 8-lane stage: row_mirror (0x140).  banks 0,1 keep q0, banks 2,3 keep q1: partner values come through a DPP move into a
    // register preloaded with the other operand's partner, i.e. t = lanes(b01) ? q0.mirror : q1.mirror
    float t = dpp_move<0x140, 0x3>(0.f, q0);
    t = dpp_move<0x140, 0xc>(t, q1);
    const bool hi8 = (threadIdx.x & 8) != 0;
    const float r0 = (hi8 ? q1 : q0) + t;       // lanes 0-7: q0 pairs, lanes 8-15: q1 pairs
    const float r1 = q2 + dpp_move<0x140, 0xf>(0.f, q2);
    // 4-lane stage: row_half_mirror (0x141).  banks 0,2 keep r0, banks 1,3 keep r1
    float u = dpp_move<0x141, 0x5>(0.f, r0);
    u = dpp_move<0x141, 0xa>(u, r1);
    const bool hi4 = (threadIdx.x & 4) != 0;
    float f = (hi4 ? r1 : r0) + u;              // quads: q0 | q2 | q1 | q2
    f += dpp_move<0x4E, 0xf>(0.f, f);           // quad_perm [2,3,0,1]
    f += dpp_move<0xB1, 0xf>(0.f, f);           // quad_perm [1,0,3,2]
    return f;
}

template <int VARIANT>
__global__ __launch_bounds__(64) void reduce_lab_kernel(int iters, float* __restrict__ out)
{
    const int lane = threadIdx.x;
    float acc = 0.f;
    float seed = (float)(lane + 1) * 0.001f + (float)blockIdx.x;
    for (int it = 0; it < iters; ++it) {
        float a[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) a[k] = seed * (float)(k + 1) + (float)it * 0.5f;     // 10 fmas: stands for the accumulation
        seed += 0.25f;
        if (VARIANT == 0) {
            const adk::Reduce10 r = adk::wave_reduce10(a, lane);
            acc += r.is_owner ? r.value * (float)(r.slot + 1) : 0.f;
        } else if (VARIANT == 3) {   // round 5: the same pair, rows first (adk::wave_reduce20_rows_first)
            float b[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) b[k] = seed * (float)(k + 2) - (float)it * 0.25f;
            const adk::Reduce20 r = adk::wave_reduce20_rows_first(a, b);
            acc += r.z0 + 0.5f * r.z1;
        } else if (VARIANT == 2) {   // round 4: two splats' 20 sums in one butterfly (adk::wave_reduce20); one iteration = TWO (10 fma + reduction)
            float b[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) b[k] = seed * (float)(k + 2) - (float)it * 0.25f;
            const adk::Reduce20 r = adk::wave_reduce20(a, b);
            acc += r.z0 + 0.5f * r.z1;
        } else {
            const float f = reduce10_rows_first(a);
            acc += f;
        }
    }
    out[(size_t)blockIdx.x * 64 + lane] = acc;
}

// one set of inputs, both variants, raw results per lane: for the correctness check
__global__ __launch_bounds__(64) void reduce_check_kernel(const float* __restrict__ in /* [10][64] */, float* __restrict__ out0, int* __restrict__ slot0,
                                                          int* __restrict__ owner0, float* __restrict__ out1)
{
    const int lane = threadIdx.x;
    float a[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) a[k] = in[k * 64 + lane];
    const adk::Reduce10 r = adk::wave_reduce10(a, lane);
    out0[lane] = r.value; slot0[lane] = r.slot; owner0[lane] = r.is_owner ? 1 : 0;
    float b[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) b[k] = in[k * 64 + lane];
    out1[lane] = reduce10_rows_first(b);
}

// round 4: wave_reduce20 raw results per lane
__global__ __launch_bounds__(64) void reduce20_check_kernel(const float* __restrict__ in /* [20][64] */, float* __restrict__ z0, float* __restrict__ z1)
{
    const int lane = threadIdx.x;
    float a[10], b[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) { a[k] = in[k * 64 + lane]; b[k] = in[(10 + k) * 64 + lane]; }
    const adk::Reduce20 r = adk::wave_reduce20(a, b);
    z0[lane] = r.z0; z1[lane] = r.z1;
}
extern "C" int reduce20_lab_check(const float* in, float* z0, float* z1, hipStream_t st)
{
    hipLaunchKernelGGL(reduce20_check_kernel, dim3(1), dim3(64), 0, st, in, z0, z1);
    return (int)hipGetLastError();
}
__global__ __launch_bounds__(64) void reduce20s_check_kernel(const float* __restrict__ in /* [20][64] */, float* __restrict__ z0, float* __restrict__ z1)
{
    const int lane = threadIdx.x;
    float a[10], b[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) { a[k] = in[k * 64 + lane]; b[k] = in[(10 + k) * 64 + lane]; }
    const adk::Reduce20 r = adk::wave_reduce20_rows_first(a, b);
    z0[lane] = r.z0; z1[lane] = r.z1;
}
extern "C" int reduce20s_lab_check(const float* in, float* z0, float* z1, hipStream_t st)
{
    hipLaunchKernelGGL(reduce20s_check_kernel, dim3(1), dim3(64), 0, st, in, z0, z1);
    return (int)hipGetLastError();
}

extern "C" int reduce_lab_time(int variant, int blocks, int iters, float* out, hipStream_t st)
{
    if (variant == 3) { hipLaunchKernelGGL(reduce_lab_kernel<3>, dim3(blocks), dim3(64), 0, st, iters, out); return (int)hipGetLastError(); }
    if (variant == 2) { hipLaunchKernelGGL(reduce_lab_kernel<2>, dim3(blocks), dim3(64), 0, st, iters, out); return (int)hipGetLastError(); }
    if (variant == 0) hipLaunchKernelGGL(reduce_lab_kernel<0>, dim3(blocks), dim3(64), 0, st, iters, out);
    else hipLaunchKernelGGL(reduce_lab_kernel<1>, dim3(blocks), dim3(64), 0, st, iters, out);
    return (int)hipGetLastError();
}
extern "C" int reduce_lab_check(const float* in, float* out0, int* slot0, int* owner0, float* out1, hipStream_t st)
{
    hipLaunchKernelGGL(reduce_check_kernel, dim3(1), dim3(64), 0, st, in, out0, slot0, owner0, out1);
    return (int)hipGetLastError();
}
