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

// Zero-fill as a KERNEL.  A hipMemsetAsync node captured into a hipGraph is not ordered before the kernel nodes that
// follow it on this stack (measured while building the tracker: replay 0 correct, later replays read half-cleared state),
// so nothing that may end up inside a captured step uses hipMemsetAsync.  Any size / alignment (tail bytes one by one).
static __global__ __launch_bounds__(256) void clear_bytes_kernel(unsigned char* __restrict__ p, int64_t nbytes) {
    const int64_t n4 = nbytes >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool aligned = (((uintptr_t)p) & 3) == 0;
    if (aligned) {
        uint32_t* q = reinterpret_cast<uint32_t*>(p);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) q[i] = 0u;
        for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += stride) p[i] = 0;
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += stride) p[i] = 0;
    }
}
static inline int clear_bytes(void* p, int64_t nbytes, hipStream_t stream) {
    if (nbytes <= 0) return 0;
    hipLaunchKernelGGL(clear_bytes_kernel, dim3(stream_grid(ceil_div(nbytes, 4), 256)), dim3(256), 0, stream,
                       static_cast<unsigned char*>(p), nbytes);
    return (int)hipGetLastError();
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

// Sum 10 per-lane values across the 64 lanes with as few cross-lane operations as possible.
// DPP-modified VALU ops issue at ~1/3.3 of the plain rate on gfx950 (measured, tools/dpp_bench.hip), so
// instead of 10 independent 4-step row reductions (40 DPP ops) this is a TRANSPOSING butterfly inside each
// 16-lane row: at every stage a lane keeps one half of its values and ships the other half to its partner
// (row_mirror, row_half_mirror, quad mirror, quad xor-1): 10 + 5 bank-masked DPP adds for the two bank-level stages,
// 2 + 1 DPP adds + 6 selects for the two in-quad stages.
// Afterwards lane `owner lanes` {0,1,2,4,6} (+8 for values 5..9) of each row hold the row sums of values
// {0,1,2,3,4} (+5); two permlane-swap adds finish the 4 rows.
// Returns the total of value `slot` (valid in every row's owner lanes); is_owner is true for the 10 lanes
// of row 0 that should publish it.
// MEASURED AND REJECTED (round 2, same box A/B): the whole reduction on the otherwise idle matrix cores -- ten chained
// v_mfma_f32_16x16x4_f32 with the value as A and a one-hot column as B add the 4 lane rows of quantity q into column q of D
// (exact), 3 adds + 2 permlane swaps finish -- is bit-compatible and 45 % SLOWER (raster bwd 0.60 -> 0.87 ms): the 8-pass
// MFMAs of the 5 resident waves serialise on the SIMD's one matrix pipe instead of hiding under the other waves' VALU work.
struct Reduce10 { float value; int slot; bool is_owner; };

__device__ __forceinline__ Reduce10 wave_reduce10(const float (&a)[10], int lane) {
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
    // Stages 1 and 2 exchange whole BANKS (4-lane groups), so the keep / send choice is the DPP bank_mask itself: the lanes
    // of banks 0,1 (0,2) execute `own + partner` on the registers they keep, the lanes of banks 2,3 (1,3) on the others;
    // 15 DPP adds and no selects instead of 8 DPP adds + 16 selects.  One asm block: hipcc's hazard recogniser does not
    // look into inline asm, so the 2 wait states a DPP read needs after a VALU write of the same register are the s_nop 1
    // in front (the inputs come straight out of the accumulation fmas), the instruction order inside (every read is >= 3
    // instructions after the write it depends on) and the s_nop 1 behind (the compiler's own DPP ops follow).
    float r0, r1, r2, r3, r4, u0, u1, u2;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %8 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %13, %13 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %9, %9 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %14, %14 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %10, %10 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %15, %15 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %11, %11 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %16, %16 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %12, %12 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %17, %17 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %5, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %6, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %7, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %5, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %6, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(u0), "=&v"(u1), "=&v"(u2)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]));
    // stage 3: quad mirror [3,2,1,0] (0x1B), class bit 1
    const float k30 = b2 ? (b1 ? u1 : u0) : (b1 ? u2 : u0);
    const float s30 = b2 ? (b1 ? u0 : u1) : (b1 ? u0 : u2);
    const float w0 = k30 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s30), 0x1B, 0xf, 0xf, false));
    const float w1 = u1 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u1), 0x1B, 0xf, 0xf, false)); // (b2,b1)=(0,0) only
    // stage 4: quad xor 1 [1,0,3,2] (0xB1), class bit 0
    const bool two = !b2 && !b1;
    const float k4 = (two && b0) ? w1 : w0;
    const float s4 = two ? (b0 ? w0 : w1) : w0;
    float f = k4 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s4), 0xB1, 0xf, 0xf, false));
    // the 4 rows, in registers: swap(x = f, y = f) leaves (lo, lo) / (hi, hi) resp. (even rows) / (odd rows) in x / y, so
    // x + y is the pairwise sum in every lane (gfx950 v_permlane32_swap / v_permlane16_swap: 7.2 cycles each against ~21
    // for a ds_bpermute + its lgkmcnt wait, tools/dpp_bench.hip)
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(f), __float_as_uint(f), false, false);
    f = __uint_as_float(sw.x) + __uint_as_float(sw.y);
    sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(f), __float_as_uint(f), false, false);
    f = __uint_as_float(sw.x) + __uint_as_float(sw.y);
    Reduce10 o;
    o.slot = (b3 ? 5 : 0) + (b2 ? (b1 ? 4 : 3) : (b1 ? 2 : (b0 ? 1 : 0)));
    o.is_owner = (lane < 16) && (two || !b0);
    o.value = f;
    return o;
}

// Two splats' 10 + 10 per-lane values summed across the 64 lanes in ONE transposing butterfly (round 4).  Every stage halves both the
// lane extent and the number of values a lane carries, and the stages that cost two instructions per surviving value run while there
// are many values:
//   stage 1  row_mirror, bank-masked adds      20 -> 10   banks 0,1 keep splat A's ten, banks 2,3 splat B's                (20 DPP adds)
//   stage 2  row_half_mirror, bank-masked      10 -> 5    banks 0,2 keep sums 0..4, banks 1,3 sums 5..9                    (10 DPP adds)
//   stage 3  v_permlane32_swap + add            5 -> 3    rows 0,1 keep the even register of a pair, rows 2,3 the odd one  (3 swaps + 3 adds)
//   stage 4  v_permlane16_swap + add            3 -> 2                                                                     (2 swaps + 2 adds)
//   stage 5,6  in-quad all-reduce of the 2 survivors (quad_perm xor 1, xor 2)                                             (4 DPP adds)
// = 46 cross-lane / add instructions for two splats against 2 x 32 for two wave_reduce10 (whose in-quad stages need selects and whose row
// stages carry one value each).  Result, lane = 16 r + 4 b + l: every lane of bank b holds, for splat (b >> 1),
//   z0 = the total of value (b & 1) * 5 + {0, 2, 1, 3}[r],   z1 = the total of value (b & 1) * 5 + 4.
// Checked on the GPU against a float64 sum by tools/lab/reduce_lab.py (variant 2).  EXEC must be all ones.
struct Reduce20 { float z0, z1; };

// MEASURED AND REJECTED (round 4, same box A/B): the same butterfly IN PLACE (stage 1 writes its sums over A's registers, stage 2 over A[0..4]:
// 15 fewer live registers, 96 instead of 99 VGPRs in the whole-tile kernel) is 3 % SLOWER in every form (raster_bwd 0.525 -> 0.541 ms in the
// two-halves form, 0.533 -> 0.545 whole-tile at 5 waves): a bank-masked DPP add whose destination is also its source serialises on that register.
__device__ __forceinline__ Reduce20 wave_reduce20(const float (&A)[10], const float (&B)[10]) {
    float r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, u0, u1, u2, u3, u4;
    // inline asm: see wave_reduce10 for the wait-state reasoning (s_nop 1 in front: the inputs come straight out of the accumulation fmas;
    // every DPP read below is >= 3 instructions after the write it depends on).  30-operand limit => three blocks.
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %5, %5 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %6, %6 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %7, %7 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %8, %8 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %9, %9 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %10, %10 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %11, %11 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %12, %12 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %13, %13 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %14, %14 row_mirror row_mask:0xf bank_mask:0xc"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4)
        : "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(B[0]), "v"(B[1]), "v"(B[2]), "v"(B[3]), "v"(B[4]));
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %5, %5 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %6, %6 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %7, %7 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %8, %8 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %9, %9 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %10, %10 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %11, %11 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %12, %12 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %13, %13 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %14, %14 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 1"
        : "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8), "=&v"(r9)
        : "v"(A[5]), "v"(A[6]), "v"(A[7]), "v"(A[8]), "v"(A[9]), "v"(B[5]), "v"(B[6]), "v"(B[7]), "v"(B[8]), "v"(B[9]));
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %5, %5 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %6, %6 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %7, %7 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %3, %8, %8 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %4, %9, %9 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %11, %11 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %12, %12 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %13, %13 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %4, %14, %14 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1"
        : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4)
        : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(r8), "v"(r9));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    // swap32(x, y): lanes 0-31 get x.lo + x.hi, lanes 32-63 y.lo + y.hi;  swap16(x, y): rows get x.r0 + x.r1 | y.r0 + y.r1 | x.r2 + x.r3 | y.r2 + y.r3
    u32x2 s = __builtin_amdgcn_permlane32_swap(__float_as_uint(u0), __float_as_uint(u1), false, false);
    const float w0 = __uint_as_float(s.x) + __uint_as_float(s.y);
    s = __builtin_amdgcn_permlane32_swap(__float_as_uint(u2), __float_as_uint(u3), false, false);
    const float w1 = __uint_as_float(s.x) + __uint_as_float(s.y);
    s = __builtin_amdgcn_permlane32_swap(__float_as_uint(u4), __float_as_uint(u4), false, false);
    const float w2 = __uint_as_float(s.x) + __uint_as_float(s.y);
    s = __builtin_amdgcn_permlane16_swap(__float_as_uint(w0), __float_as_uint(w1), false, false);
    float z0 = __uint_as_float(s.x) + __uint_as_float(s.y);
    s = __builtin_amdgcn_permlane16_swap(__float_as_uint(w2), __float_as_uint(w2), false, false);
    float z1 = __uint_as_float(s.x) + __uint_as_float(s.y);
    z0 = dpp_add<0xB1>(z0); z1 = dpp_add<0xB1>(z1);   // quad_perm [1,0,3,2]
    z0 = dpp_add<0x4E>(z0); z1 = dpp_add<0x4E>(z1);   // quad_perm [2,3,0,1]
    Reduce20 o;
    o.z0 = z0; o.z1 = z1;
    return o;
}

// ROWS FIRST (round 5): the same 20 -> 2 transposing butterfly with the stages in the opposite order.  v_permlane32_swap / v_permlane16_swap
// exchange between lane halves / row pairs AND keep their own half in one instruction, so a swap + one plain add folds TWO values; the
// bank-masked DPP stages cost two cross-lane instructions per surviving value.  wave_reduce20 spends its 30 bank-masked DPP adds while there
// are 20 and 10 values and its swaps on the last 5; here the swaps run on 20 and 10 values and the DPP stages on 5 -> 3 -> 2:
//   stage 1  v_permlane32_swap(A[k], B[k]) + add      20 -> 10   lanes 0-31 keep splat A's value k, lanes 32-63 splat B's       (10 swaps + 10 adds)
//   stage 2  v_permlane16_swap(R[k], R[k + 5]) + add  10 -> 5    rows 0 / 1 / 2 / 3 keep A_k / A_(k+5) / B_k / B_(k+5)          (5 swaps + 5 adds)
//   stage 3  row_mirror, bank-masked adds              5 -> 3                                                                    (5 DPP adds)
//   stage 4  row_half_mirror, bank-masked              3 -> 2                                                                    (3 DPP adds)
//   stage 5,6  in-quad all-reduce of the 2 survivors                                                                             (4 DPP adds)
// = 27 cross-lane instructions + 15 plain adds per PAIR against 39 + 7.  Result, lane = 16 r + 4 b + l: splat (r >> 1),
//   z0 = the total of value (r & 1) * 5 + {0, 2, 1, 3}[b],   z1 = the total of value (r & 1) * 5 + 4   (in every lane of the bank).
// Checked against a float64 sum by tools/lab/reduce_lab.py (variant 3).  EXEC must be all ones.
__device__ __forceinline__ Reduce20 wave_reduce20_rows_first(const float (&A)[10], const float (&B)[10]) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    float R[10], Wv[5];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const u32x2 s = __builtin_amdgcn_permlane32_swap(__float_as_uint(A[k]), __float_as_uint(B[k]), false, false);
        R[k] = __uint_as_float(s.x) + __uint_as_float(s.y);      // lanes 0-31: A[k] over both halves, lanes 32-63: B[k]
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const u32x2 s = __builtin_amdgcn_permlane16_swap(__float_as_uint(R[k]), __float_as_uint(R[k + 5]), false, false);
        Wv[k] = __uint_as_float(s.x) + __uint_as_float(s.y);     // rows: A_k | A_(k+5) | B_k | B_(k+5), 16 per-lane partials each
    }
    float x0, x1, x2, y0, y1;
    // one asm block per DPP stage (hipcc's hazard recogniser does not look into inline asm: a DPP read needs 2 wait states after a VALU write
    // of the same register -- the s_nop 1 in front of each block; inside a block no instruction reads what the block wrote)
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %3, %3 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %5, %5 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %4 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %6, %6 row_mirror row_mask:0xf bank_mask:0xc"
        : "=&v"(x0), "=&v"(x1), "=&v"(x2)
        : "v"(Wv[0]), "v"(Wv[1]), "v"(Wv[2]), "v"(Wv[3]), "v"(Wv[4]));
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1"
        : "=&v"(y0), "=&v"(y1)
        : "v"(x0), "v"(x1), "v"(x2));
    // y0: bank 0 -> W0, bank 1 -> W2, bank 2 -> W1, bank 3 -> W3 (4 partials each); y1: every bank -> W4
    y0 = dpp_add<0xB1>(y0); y1 = dpp_add<0xB1>(y1);   // quad_perm [1,0,3,2]
    y0 = dpp_add<0x4E>(y0); y1 = dpp_add<0x4E>(y1);   // quad_perm [2,3,0,1]
    Reduce20 o;
    o.z0 = y0; o.z1 = y1;
    return o;
}

} // namespace adk
