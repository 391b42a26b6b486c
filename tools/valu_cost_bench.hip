// DEV TOOL: issue cost, in shader-clock cycles per wave64 instruction per SIMD, of the instruction classes raster_bwd_kernel is made of
// (profiles/r02_raster_bwd_isa.txt), measured on the GPU it runs on.  Each kernel runs REPS x 16 independent instances of one
// instruction (inline asm, distinct registers: no dependency stalls) in 8 waves per SIMD, best of 3 launches of several ms each; cost =
// wall time / instructions per SIMD, quoted in ns and in cycles of the nominal 2.4 GHz clock (the sustained clock under a pure-VALU
// load is lower; ratios between classes are what matters).
//     hipcc --offload-arch=gfx950 -O3 tools/valu_cost_bench.hip -o tools/valu_cost_bench && tools/valu_cost_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REPS 20000
#define UNROLL 16

#define BODY16(stmt) stmt(0) stmt(1) stmt(2) stmt(3) stmt(4) stmt(5) stmt(6) stmt(7) stmt(8) stmt(9) stmt(10) stmt(11) stmt(12) stmt(13) stmt(14) stmt(15)

template <int MODE> __global__ __launch_bounds__(64) void k(float* out, unsigned long long* cycles)
{
    float a[UNROLL];
    const float b = threadIdx.x * 0.001f + 1.0f, c = 0.999f;
    unsigned long long m[4] = {0x5555555555555555ull, 0x3333333333333333ull, 0x0f0f0f0f0f0f0f0full, 0x00ff00ff00ff00ffull};
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) a[i] = b + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
        if (MODE == 0) {
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            BODY16(S)
#undef S
        } else if (MODE == 1) {
#define S(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
            BODY16(S)
#undef S
        } else if (MODE == 2) {
#define S(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            BODY16(S)
#undef S
        } else if (MODE == 3) {
#define S(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            BODY16(S)
#undef S
        } else if (MODE == 4) { // compare writing an SGPR pair (VOP3), as the validity tests do
#define S(i) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m[i & 3]) : "v"(a[i]), "v"(b));
            BODY16(S)
#undef S
        } else if (MODE == 5) { // compare writing vcc (VOPC)
#define S(i) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            BODY16(S)
#undef S
        } else if (MODE == 6) { // select on an SGPR-pair mask (VOP3)
#define S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(m[i & 3]));
            BODY16(S)
#undef S
        } else if (MODE == 7) { // DPP add inside a row
#define S(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
            BODY16(S)
#undef S
        } else if (MODE == 8) { // packed fp32 fma (two floats per lane): what the SLP vectoriser emits
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2* p = reinterpret_cast<f2*>(a);
            f2 bb = {b, b}, cc = {c, c};
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i & 7]) : "v"(bb), "v"(cc));
            BODY16(S)
#undef S
        } else if (MODE == 9) {
#define S(i) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
            BODY16(S)
#undef S
        } else if (MODE == 10) { // scalar ALU on SGPR pairs (mask logic)
#define S(i) asm volatile("s_and_b64 %0, %0, %1" : "+s"(m[i & 3]) : "s"(m[(i + 1) & 3]));
            BODY16(S)
#undef S
        } else if (MODE == 12) { // VOP2 fused multiply-add (dst is the addend)
#define S(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            BODY16(S)
#undef S
        } else if (MODE == 13) {
#define S(i) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
            BODY16(S)
#undef S
        } else if (MODE == 15) {
#define S(i) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[i]) : "v"(b));
            BODY16(S)
#undef S
        } else if (MODE == 16) {
#define S(i) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
            BODY16(S)
#undef S
        } else if (MODE == 17) { // multiply by an SGPR / by an inline constant
#define S(i) asm volatile("v_mul_f32_e32 %0, 0.5, %0" : "+v"(a[i]));
            BODY16(S)
#undef S
        } else if (MODE == 18) {
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b), "v"(c), "v"(a[(i + 1) & 15]));
            BODY16(S)
#undef S
        } else if (MODE == 19) { // v_max3 (VOP3, three VGPR sources)
#define S(i) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            BODY16(S)
#undef S
        } else if (MODE == 20) {
#define S(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            BODY16(S)
#undef S
        } else if (MODE == 11) { // LDS broadcast read of one float4 (wave-uniform address), as the staged splat records are read
            __shared__ float4 lds[64];
            if (r == 0) lds[threadIdx.x] = make_float4(b, b, b, b);
            float4 v;
#define S(i) asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(0)); a[i] += v.x;
            BODY16(S)
#undef S
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += a[i];
    s += (float)(m[0] ^ m[1] ^ m[2] ^ m[3]);
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, int extra_per_iter = 0)
{
    const int waves_per_simd = 8, blocks = 256 * 4 * waves_per_simd;
    float* d; unsigned long long* cyc;
    hipMalloc(&d, blocks * 64 * sizeof(float)); hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, cyc);
    float ms = 1e30f;
    for (int rep = 0; rep < 3; ++rep) { // best of 3: the first launches run while the clocks are still ramping up
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1);
        if (t < ms) ms = t;
    }
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    const double insts = (double)REPS * (UNROLL + extra_per_iter);
    // all 8 waves of a SIMD are resident from start to end: the SIMD issued 8 x insts instructions in `ms`
    const double ns = ms * 1e6 / (waves_per_simd * insts);
    printf("%-44s %6.3f ns per wave-instruction per SIMD = %5.2f cycles at 2.4 GHz   (kernel %.3f ms; s_memtime ticks per instruction %.3f)\n",
           name, ns, ns * 2.4, ms, mean / (waves_per_simd * insts));
    hipFree(d); hipFree(cyc);
}

int main()
{
    run<0>("v_fma_f32 (plain VALU)");
    run<1>("v_mul_f32");
    run<9>("v_min_f32");
    run<2>("v_exp_f32 (transcendental)");
    run<3>("v_rcp_f32 (transcendental)");
    run<4>("v_cmp_gt_f32_e64 -> SGPR pair");
    run<5>("v_cmp_gt_f32_e32 -> vcc");
    run<6>("v_cndmask_b32_e64, SGPR-pair mask");
    run<7>("v_add_f32_dpp row_mirror");
    run<8>("v_pk_fma_f32 (2 floats / lane)");
    run<12>("v_fmac_f32_e32 (VOP2 fma)");
    run<18>("v_fma_f32 with a separate destination");
    run<13>("v_add_f32_e32");
    run<16>("v_max_f32_e32");
    run<19>("v_max3_f32");
    run<17>("v_mul_f32_e32 by an inline constant");
    run<15>("v_mov_b32_e32");
    run<20>("v_cvt_pk_f16_f32");
    run<10>("s_and_b64 (SALU)");
    run<11>("ds_read_b128 broadcast + wait, + v_add", 16);
    return 0;
}
