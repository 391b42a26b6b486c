// DEV TOOL (round 6): second, wider pass over the issue cost of wave64 VALU instruction FORMS on gfx950 -- round 2's table
// (profiles/r02_valu_cost.txt) showed two cost classes (v_mul / v_add / v_mov / v_fma with a separate destination at ~2.5 "cycles",
// everything else -- v_fmac, accumulating v_fma, min / max / compares / selects / DPP -- at ~4) without saying what puts an instruction
// in the cheap class.  This tool varies ONE property at a time (destination == a source or not, VOP2 / VOP3 encoding, SGPR / literal
// operands, integer vs float, DPP controls, packed forms) and also times MIXES of the two classes, at 8 / 4 / 2 / 1 waves per SIMD.
// Same method as tools/valu_cost_bench.hip: REPS x 16 independent instances in inline asm, best of 3 launches.
//     hipcc --offload-arch=gfx950 -O3 tools/valu_cost_bench2.hip -o tools/valu_cost_bench2 && tools/valu_cost_bench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REPS 12000
#define B16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(64) void k(float* out, float sarg)
{
    float a[16], e[16];
    const float b = threadIdx.x * 0.001f + 1.0f;
    float c = 0.999f;
    asm volatile("v_mov_b32 %0, %0" : "+v"(c)); // keep c in a VGPR of its own
    unsigned long long m[4] = {0x5555555555555555ull, 0x3333333333333333ull, 0x0f0f0f0f0f0f0f0full, 0x00ff00ff00ff00ffull};
    float s = sarg;
    asm volatile("s_mov_b32 %0, %0" : "+s"(s));
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = b + i; e[i] = b - i; }
    f2* pa = reinterpret_cast<f2*>(a);
    f2* pe = reinterpret_cast<f2*>(e);
    f2 bb = {b, c};
    for (int r = 0; r < REPS; ++r) {
        if (MODE == 0) {
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            B16(S)
#undef S
        } else if (MODE == 1) {
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b), "v"(c), "v"(a[(i + 1) & 15]));
            B16(S)
#undef S
        } else if (MODE == 2) { // separate destination array, sources never written
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e[i]) : "v"(b), "v"(c), "v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 3) { // ping-pong accumulate: e = fma(.., a); a = fma(.., e)   (32 instructions per iteration)
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(e[i]) : "v"(b), "v"(c), "v"(a[i]));
            B16(S)
#undef S
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(a[i]) : "v"(b), "v"(c), "v"(e[i]));
            B16(S)
#undef S
        } else if (MODE == 4) {
#define S(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            B16(S)
#undef S
        } else if (MODE == 5) {
#define S(i) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
            B16(S)
#undef S
        } else if (MODE == 6) {
#define S(i) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(e[i]) : "v"(c), "v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 7) {
#define S(i) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
            B16(S)
#undef S
        } else if (MODE == 8) {
#define S(i) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
            B16(S)
#undef S
        } else if (MODE == 9) { // fma, destination == src0 (a multiplicand), addend separate
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            B16(S)
#undef S
        } else if (MODE == 10) { // accumulating fma with an SGPR multiplicand
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(s), "v"(c));
            B16(S)
#undef S
        } else if (MODE == 11) { // separate-destination fma with an SGPR multiplicand
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e[i]) : "s"(s), "v"(c), "v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 12) {
#define S(i) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
            B16(S)
#undef S
        } else if (MODE == 13) {
#define S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(m[i & 3]));
            B16(S)
#undef S
        } else if (MODE == 14) {
            asm volatile("s_mov_b64 vcc, %0" : : "s"(m[0]) : "vcc");
#define S(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            B16(S)
#undef S
        } else if (MODE == 15) {
#define S(i) asm volatile("v_min_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
            B16(S)
#undef S
        } else if (MODE == 16) { // min, separate destination
#define S(i) asm volatile("v_min_f32_e32 %0, %1, %2" : "=v"(e[i]) : "v"(c), "v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 17) {
#define S(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            B16(S)
#undef S
        } else if (MODE == 18) {
#define S(i) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m[i & 3]) : "v"(a[i]), "v"(b));
            B16(S)
#undef S
        } else if (MODE == 19) {
#define S(i) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            B16(S)
#undef S
        } else if (MODE == 20) {
#define S(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 21) {
#define S(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 22) { // DPP add, separate destination
#define S(i) asm volatile("v_add_f32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(e[i]) : "v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 23) {
#define S(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(e[i]) : "v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 24) { // DPP add under a partial bank mask (what the butterfly's bank stages use)
#define S(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0x5" : "+v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 25) {
#define S(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 26) {
#define S(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 27) {
#define S(i) asm volatile("v_exp_f32 %0, %1" : "=v"(e[i]) : "v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 28) {
#define S(i) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
            B16(S)
#undef S
        } else if (MODE == 29) {
#define S(i) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
            B16(S)
#undef S
        } else if (MODE == 30) {
#define S(i) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 31) {
#define S(i) asm volatile("v_sub_u32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
            B16(S)
#undef S
        } else if (MODE == 32) {
#define S(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[i & 7]) : "v"(bb));
            B16(S)
#undef S
        } else if (MODE == 33) {
#define S(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa[i & 7]) : "v"(bb));
            B16(S)
#undef S
        } else if (MODE == 34) { // packed fma with a separate destination
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %2" : "=v"(pe[i & 7]) : "v"(pa[i & 7]), "v"(bb));
            B16(S)
#undef S
        } else if (MODE == 35) { // packed fma, accumulate form
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(pa[i & 7]) : "v"(bb));
            B16(S)
#undef S
        } else if (MODE == 36) {
            int sg;
#define S(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg) : "v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 37) {
#define S(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(e[i]));
            B16(S)
#undef S
        } else if (MODE == 38) {
#define S(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(e[i]));
            B16(S)
#undef S
        } else if (MODE == 39) { // MIX: v_mul (cheap class) and v_cndmask (dear class), alternating
#define S(i) asm volatile("v_mul_f32_e32 %0, %2, %0\n\tv_cndmask_b32_e64 %1, %1, %2, %3" : "+v"(a[i]), "+v"(e[i]) : "v"(c), "s"(m[i & 3]));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#undef S
        } else if (MODE == 40) { // MIX: v_mul and v_exp alternating
#define S(i) asm volatile("v_mul_f32_e32 %0, %2, %0\n\tv_exp_f32 %1, %1" : "+v"(a[i]), "+v"(e[i]) : "v"(c));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#undef S
        } else if (MODE == 41) { // MIX: v_mul and a DPP add alternating
#define S(i) asm volatile("v_mul_f32_e32 %0, %2, %0\n\tv_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]), "+v"(e[i]) : "v"(c));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#undef S
        } else if (MODE == 42) { // MIX: 3 v_mul per v_exp
#define S(i) asm volatile("v_mul_f32_e32 %0, %2, %0\n\tv_mul_f32_e32 %1, %2, %1\n\tv_mul_f32_e32 %0, %2, %0\n\tv_exp_f32 %1, %1" : "+v"(a[i]), "+v"(e[i]) : "v"(c));
            S(0) S(1) S(2) S(3)
#undef S
        } else if (MODE == 43) { // VOP3-encoded multiply (64-bit encoding, same operation)
#define S(i) asm volatile("v_mul_f32_e64 %0, %1, %0" : "+v"(a[i]) : "v"(c));
            B16(S)
#undef S
        } else if (MODE == 44) { // add with a 32-bit literal (64-bit encoding)
#define S(i) asm volatile("v_add_f32_e32 %0, 0x3f7fbe77, %0" : "+v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 45) { // fma with a literal addend: d = a * b + K
#define S(i) asm volatile("v_fmaak_f32 %0, %1, %2, 0x3f7fbe77" : "=v"(e[i]) : "v"(c), "v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 46) {
#define S(i) asm volatile("v_cvt_f32_i32_e32 %0, %0" : "+v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 47) {
#define S(i) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(e[i]) : "v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 48) { // multiply with a source modifier (forces VOP3)
#define S(i) asm volatile("v_mul_f32_e64 %0, -%1, %0" : "+v"(a[i]) : "v"(c));
            B16(S)
#undef S
        } else if (MODE == 49) { // max3 separate destination
#define S(i) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(e[i]) : "v"(b), "v"(c), "v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 50) { // fmac with an SGPR multiplicand (VOP2)
#define S(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(c));
            B16(S)
#undef S
        } else if (MODE == 51) { // accumulate as mul + add into the same register (2 instructions)
#define S(i) asm volatile("v_mul_f32_e32 %1, %2, %3\n\tv_add_f32_e32 %0, %1, %0" : "+v"(a[i]), "=&v"(e[i]) : "v"(b), "v"(c));
            S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#undef S
        } else if (MODE == 52) { // accumulating fma whose addend/destination differs per instance, multiplicands too (bank spread)
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(e[i]), "v"(e[(i + 5) & 15]));
            B16(S)
#undef S
        } else if (MODE == 53) { // separate-destination fma, all three sources varying
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(a[i]) : "v"(e[i]), "v"(e[(i + 5) & 15]), "v"(e[(i + 9) & 15]));
            B16(S)
#undef S
        } else if (MODE == 54) { // compare against an inline constant, VOPC
#define S(i) asm volatile("v_cmp_lt_f32_e32 vcc, 0, %0" : : "v"(a[i]) : "vcc");
            B16(S)
#undef S
        } else if (MODE == 55) { // integer compare
#define S(i) asm volatile("v_cmp_le_i32_e64 %0, %1, %2" : "=s"(m[i & 3]) : "v"(a[i]), "v"(b));
            B16(S)
#undef S
        } else if (MODE == 56) { // v_mad_u32_u24-free integer multiply-add: v_mad_i32_i24
#define S(i) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            B16(S)
#undef S
        } else if (MODE == 57) { // 16 s_nop: the loop's own overhead
#define S(i) asm volatile("s_nop 0");
            B16(S)
#undef S
        } else if (MODE == 58) { // v_sub with the REV form and an SGPR (dx = mean - pixel with the mean in an SGPR)
#define S(i) asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(e[i]) : "s"(s), "v"(a[i]));
            B16(S)
#undef S
        } else if (MODE == 59) { // v_mul_legacy / v_mul with omod (x2): VOP3 output modifier
#define S(i) asm volatile("v_mul_f32_e64 %0, %1, %0 mul:2" : "+v"(a[i]) : "v"(c));
            B16(S)
#undef S
        }
    }
    float t = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += a[i] + e[i];
    t += (float)(m[0] ^ m[1] ^ m[2] ^ m[3]);
    out[blockIdx.x * 64 + threadIdx.x] = t;
}

template <int MODE> void run(const char* name, int per_iter = 16, int waves_per_simd = 8)
{
    const int blocks = 256 * 4 * waves_per_simd;
    float* d;
    hipMalloc(&d, blocks * 64 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 1.0001f);
    float ms = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 1.0001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1);
        if (t < ms) ms = t;
    }
    const double insts = (double)REPS * per_iter;
    const double ns = ms * 1e6 / (waves_per_simd * insts);
    printf("[%2d] %-72s w/SIMD %d  %6.3f ns per wave-instruction per SIMD = %5.2f cyc @2.4GHz  (kernel %.3f ms)\n", MODE, name, waves_per_simd, ns, ns * 2.4, ms);
    fflush(stdout);
    hipFree(d);
}

int main()
{
    run<57>("16 x s_nop (loop overhead)");
    run<5>("v_mul_f32_e32 d = c * d");
    run<6>("v_mul_f32_e32 separate destination");
    run<43>("v_mul_f32_e64 (VOP3 encoding)");
    run<48>("v_mul_f32_e64 with a neg modifier");
    run<59>("v_mul_f32_e64 with omod mul:2");
    run<12>("v_mul_f32_e32 by an SGPR");
    run<7>("v_add_f32_e32 in place");
    run<8>("v_sub_f32_e32 in place");
    run<58>("v_sub_f32_e32 sgpr - vgpr, separate destination");
    run<44>("v_add_f32_e32 with a 32-bit literal");
    run<47>("v_mov_b32_e32 separate destination");
    run<0>("v_fma_f32 d = b * c + d (accumulate, constant multiplicands)");
    run<52>("v_fma_f32 d = x * y + d (accumulate, varying multiplicands)");
    run<4>("v_fmac_f32_e32 d += b * c");
    run<50>("v_fmac_f32_e32 d += sgpr * c");
    run<10>("v_fma_f32 d = sgpr * c + d");
    run<9>("v_fma_f32 d = d * b + c (destination = multiplicand)");
    run<1>("v_fma_f32 a[i] = b * c + a[i+1]");
    run<2>("v_fma_f32 e[i] = b * c + a[i] (separate arrays)");
    run<53>("v_fma_f32 a[i] = x * y + z, all varying, early-clobber destination");
    run<11>("v_fma_f32 e[i] = sgpr * c + a[i]");
    run<3>("v_fma_f32 ping-pong accumulate e = f(a); a = f(e)", 32);
    run<51>("accumulate as v_mul + v_add", 16);
    run<45>("v_fmaak_f32 d = a * b + literal");
    run<15>("v_min_f32_e32 in place");
    run<16>("v_min_f32_e32 separate destination");
    run<17>("v_med3_f32");
    run<49>("v_max3_f32 separate destination");
    run<13>("v_cndmask_b32_e64 SGPR-pair mask");
    run<14>("v_cndmask_b32_e32 vcc");
    run<18>("v_cmp_gt_f32_e64 -> SGPR pair");
    run<19>("v_cmp_gt_f32_e32 -> vcc");
    run<54>("v_cmp_lt_f32_e32 vcc, 0, v");
    run<55>("v_cmp_le_i32_e64 -> SGPR pair");
    run<20>("v_add_f32_dpp row_mirror");
    run<21>("v_add_f32_dpp quad_perm");
    run<24>("v_add_f32_dpp quad_perm bank_mask:0x5");
    run<22>("v_add_f32_dpp row_shr:1 separate destination");
    run<23>("v_mov_b32_dpp row_shr:1");
    run<25>("v_exp_f32 in place");
    run<27>("v_exp_f32 separate destination");
    run<26>("v_rcp_f32");
    run<28>("v_and_b32");
    run<29>("v_add_u32");
    run<31>("v_sub_u32");
    run<30>("v_lshlrev_b32 by 1");
    run<56>("v_mad_i32_i24 accumulate");
    run<46>("v_cvt_f32_i32");
    run<32>("v_pk_mul_f32 (2 floats / lane)");
    run<33>("v_pk_add_f32");
    run<35>("v_pk_fma_f32 accumulate");
    run<34>("v_pk_fma_f32 separate destination");
    run<36>("v_readlane_b32");
    run<37>("v_permlane32_swap_b32");
    run<38>("v_permlane16_swap_b32");
    run<39>("MIX 8 x (v_mul + v_cndmask_e64)");
    run<40>("MIX 8 x (v_mul + v_exp)");
    run<41>("MIX 8 x (v_mul + v_add_dpp)");
    run<42>("MIX 4 x (3 v_mul + v_exp)");
    // occupancy: does the per-instruction cost depend on the number of waves sharing the SIMD?
    for (int w : {4, 2, 1}) {
        if (w == 4) { run<5>("v_mul_f32 in place", 16, 4); run<4>("v_fmac_f32", 16, 4); run<2>("v_fma separate dst", 16, 4); run<13>("v_cndmask_e64", 16, 4); run<20>("v_add_dpp", 16, 4); run<25>("v_exp", 16, 4); }
        if (w == 2) { run<5>("v_mul_f32 in place", 16, 2); run<4>("v_fmac_f32", 16, 2); run<2>("v_fma separate dst", 16, 2); run<13>("v_cndmask_e64", 16, 2); run<20>("v_add_dpp", 16, 2); run<25>("v_exp", 16, 2); }
        if (w == 1) { run<5>("v_mul_f32 in place", 16, 1); run<4>("v_fmac_f32", 16, 1); run<2>("v_fma separate dst", 16, 1); run<13>("v_cndmask_e64", 16, 1); run<20>("v_add_dpp", 16, 1); run<25>("v_exp", 16, 1); }
    }
    return 0;
}
