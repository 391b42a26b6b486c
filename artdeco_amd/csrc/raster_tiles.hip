// Per-tile alpha compositing, forward and backward, for gfx950 (CDNA4).
//
// Replaces gsplat's rasterize_to_pixels fwd/bwd [UPSTREAM gsplat >= 1.5, not vendored; semantics
// per SURVEY.md App. A items 4-5], reached from gsplat.rendering.rasterization at
// Reconstruct/scene/scene_models/h3dgsv3.py:664-680.  Per-pixel results follow the upstream
// control flow exactly (skip sigma<0 or alpha<1/255, terminate BEFORE adding when
// T(1-alpha) <= 1e-4, last contributing index, back-to-front gradient recurrences).
//
// Wave64 design (not a port of the 32-lane warp tiling):
//   * a workgroup = ONE wavefront = one 16x16 tile; lane l owns pixel l (8x8 raster order,
//     lane -> (lane&7, lane>>3)) of each of the four 8x8 QUADRANTS, the squarest footprint 64
//     lanes can have, so a splat that misses a quadrant costs that quadrant nothing;
//   * splats are staged 64 at a time in LDS as packed 48 B records (three ds_read_b128 per
//     splat, wave-uniform address => broadcast reads, no bank conflicts);
//   * SPLAT-PARALLEL CULLING: before walking a group of 64 staged splats, each LANE tests ONE
//     splat's screen-space box against the wave's quadrant; __ballot gives a 64-bit mask in
//     SGPRs and the wave then walks only the set bits (s_ff1 / s_flbit), in order.  64 cull
//     tests cost what one used to, and culled splats cost nothing in the compositing loop.
//     The test is conservative w.r.t. the alpha>=1/255 rule (the radii bound exactly that).
//   * backward: the same one-wave-per-tile layout; per splat only the quadrants whose cull bit is set are
//     evaluated, their contributions summed in 10 registers, reduced ONCE per (splat, tile) across the 64 lanes
//     (transposing DPP butterfly), parked by the 10 owner lanes in an LDS table and flushed after every group of 64 staged
//     splats with hardware fp32 atomics that cover whole 48 B gradient records.
//   * tiles are mapped to workgroups through an XCD-aware bijection so that neighbouring tiles
//     (which share Gaussians) run on the same XCD and hit the same 4 MiB L2.
#include "adk_common.hpp"
#include <type_traits>

namespace adk {

#define TILE 16
#define BATCH 256
// How the per-quadrant cull results reach the scalar unit: one 64-bit ballot per quadrant whose bits the wave tests per splat
// (round 2's form), or every lane keeps its own splat's quadrant hits as a bit set and the wave reads one lane's set per splat
// (v_readlane).  Same-box A/B at 1 M / 1080p (gpurun_out/r03_ab_cull_style.txt, round 3): forward 0.2715 ms with ballots, 0.2655 with
// the per-lane set; backward 0.567 with ballots, 0.574 with the per-lane set -- so each kernel keeps the form that won.
#ifndef ADK_CULL_BALLOT_FWD
#define ADK_CULL_BALLOT_FWD 0
#endif
#ifndef ADK_CULL_BALLOT_BWD
#define ADK_CULL_BALLOT_BWD 1
#endif
// Round-4 forms of the backward's accumulate / reduce steps.  Each is a lab knob (set to 0 to get the round-3 form back); the defaults are
// what the same-box A/Bs kept (tools/lab/ab_bwd_r04.sh, profiles/r04_ab_bwd*.txt; 1 M Gaussians, ms, whole tile / two halves at 1080p, quadrants at 512x384):
//     round 3                               0.563 / 0.588    0.1055
//     FIRST                                 0.549 / 0.558    0.1045      first quadrant that blends a splat WRITES its 10 products (no zero fill, no fmac onto 0)
//     FIRST + PAIR                          0.551 / 0.532    0.1015      two splats' 20 sums reduced in one butterfly (wave_reduce20): 46 instead of 2 x 32 instructions
//     FIRST + PAIR + LEAN                   0.541 / 0.525    0.0993      validity folded into ov once, E = C0 - bdot, clamp in a not-taken branch
//     FIRST + PAIR + LEAN, >= 5 waves       0.533 / 0.525    0.0998      (the whole-tile form needs 99 VGPRs otherwise: 4 waves)
// Counters (profiles/r04_pmc_bwd_variants.txt, whole tile): SQ_INSTS_VALU 333.6 M -> 289.8 M (-13 %), SQ_ACTIVE_INST_VALU 349.3 M -> 306.7 M
// quad-cycles (-12 %) for -4 % of time at 4 waves per SIMD: with fewer instructions the kernel stops being purely VALU-bound and the lost wave
// shows; at 5 waves (MINWAVES) and in the two-halves form (6 waves) most of the saving arrives.
#ifndef ADK_BWD_FIRST
#define ADK_BWD_FIRST 1
#endif
#ifndef ADK_BWD_PAIR
#define ADK_BWD_PAIR 1
#endif
#ifndef ADK_BWD_LEAN
#define ADK_BWD_LEAN 1
#endif
// Round 5: the paired reduction with its row stages (permlane swaps: one cross-lane instruction per TWO values) in front of the in-row DPP
// stages (two per surviving value) instead of behind them: 27 cross-lane instructions + 15 adds per pair against 39 + 7 (adk_common.hpp:
// wave_reduce20_rows_first).  0 = round 4's order.
#ifndef ADK_BWD_ROWS_FIRST
#define ADK_BWD_ROWS_FIRST 0
#endif
// Round 6 (tools/valu_cost_bench2.hip -> profiles/r06_valu_cost2.txt: v_mul / v_add / v_sub / v_fma / v_fmac / v_mov with VGPR operands issue at ~2.6
// cycles per wave64 instruction per SIMD, v_min / v_max / every v_cmp / v_cndmask / DPP / anything with an SGPR operand at ~4.2, v_exp / v_rcp and
// the permlane swaps at ~8.2): the per-evaluation bodies below are priced in those units and the dear class is what gets removed.
//   ADK_FWD_LEAN    forward: ONE select per evaluated quadrant (alpha zeroed where the pixel does not blend the splat: T (1 - 0) = T and 0 T = 0
//                   need no select of their own) instead of three, the terminating lanes fixed up in the branch that already handles them, and
//                   last_ids written only there (it now holds the list index in front of the splat the pixel STOPPED at, or the tile's last
//                   index if it never stopped -- all the backward needs: a splat between the last contributor and that index fails the same
//                   alpha tests in the backward as it did here)
//   ADK_CLAMP_HOIST min(0.999, alpha) -- and the backward's "clamped alpha passes no gradient" compare -- only for splats whose opacity can
//                   reach 0.999 at all (alpha <= opacity wherever sigma >= 0): one ballot per staged batch, one scalar bit test per splat
#ifndef ADK_FWD_LEAN
#define ADK_FWD_LEAN 1
#endif
#ifndef ADK_CLAMP_HOIST
#define ADK_CLAMP_HOIST 1
#endif
#ifndef ADK_CLAMP_HOIST_BWD
#define ADK_CLAMP_HOIST_BWD 0   // measured (profiles/r06_ab_lean1.txt): forward -2 % alone / -9.1 % with FWD_LEAN; backward +0.7 % (two selects fewer, six scalar instructions more, at 6 waves)
#endif
// A wave-uniform flag tested where it is used: laundered through an empty asm at the use site so that the compare stays next to its branch
// (s_cmp + s_cbranch_scc).  Hoisted into the defining block the i1 crossed basic blocks as a lane mask that hipcc rebuilt through a VGPR
// (v_cndmask 0,1 + v_cmp_ne: two dear VALU instructions per use).
__device__ __forceinline__ bool scalar_flag(unsigned f) { asm("" : "+s"(f)); return f != 0u; }
#define MAX_ALPHA 0.999f
#define CLAMP_OPAC 0.998f   // alpha = exp2(e) <= exp2(log2 opacity) (1 + 2^-21) < 0.999 for every opacity at or below this
#define ALPHA_THR (1.0f / 255.0f)
#define T_EPS 1e-4f

// Exact, conservative test "can this splat reach alpha >= 1/255 at any pixel centre of the rectangle
// [x0,x1] x [y0,y1]?"  alpha = o*exp(-sigma) >= 1/255  <=>  sigma <= ln(255 o); sigma is a positive-
// definite quadratic in d = mean - pixel, so its minimum over the rectangle is 0 if the mean is inside
// and otherwise lies on one of the 4 edges (1-D quadratic, clamped vertex).  Evaluated by ONE LANE PER
// SPLAT (64 splats per wave instruction), so its ~45 instructions cost < 1 instruction per splat.
// The 1e-3 slack on sigma covers the approximate rcp/log/exp used here and in the pixel loop.
__device__ __forceinline__ bool splat_reaches_rect(float mx, float my, float a, float b, float c, float opac,
                                                   float x0, float x1, float y0, float y1)
{
    const float dxl = mx - x1, dxh = mx - x0, dyl = my - y1, dyh = my - y0;
    if (dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f) return true;
    const float tau = __logf(opac * 255.0f) + 1e-3f;
    const float nb_c = -b * __builtin_amdgcn_rcpf(c), nb_a = -b * __builtin_amdgcn_rcpf(a);
    float best;
    {
        const float dy = fminf(fmaxf(nb_c * dxl, dyl), dyh);
        best = 0.5f * (a * dxl * dxl + c * dy * dy) + b * dxl * dy;
    }
    {
        const float dy = fminf(fmaxf(nb_c * dxh, dyl), dyh);
        best = fminf(best, 0.5f * (a * dxh * dxh + c * dy * dy) + b * dxh * dy);
    }
    {
        const float dx = fminf(fmaxf(nb_a * dyl, dxl), dxh);
        best = fminf(best, 0.5f * (a * dx * dx + c * dyl * dyl) + b * dx * dyl);
    }
    {
        const float dx = fminf(fmaxf(nb_a * dyh, dxl), dxh);
        best = fminf(best, 0.5f * (a * dx * dx + c * dyh * dyh) + b * dx * dyh);
    }
    return best <= tau;
}

// What both tile kernels stage in LDS per splat (three float4): the compositing loop evaluates
//     alpha = opacity * exp(-sigma) = exp2(e),   e = dx (A dx + B dy) + C dy^2 + log2(opacity),
// with A = -log2(e)/2 a, B = -log2(e) b, C = -log2(e)/2 c pre-multiplied ONCE per (splat, tile) by the lane that stages the
// splat: v_exp_f32 is a base-2 exponential, so the per-pixel multiply by -log2(e) and the multiply by the opacity both
// disappear (2 of ~25 / ~45 VALU instructions per evaluated pixel quadrant in fwd / bwd).  Forward and backward use the
// SAME explicit fma sequence (splat_exponent) so that they take identical skip / terminate decisions on every pixel.
struct StagedSplat { float4 a, cn, col; }; // a = (mean2d.x, mean2d.y, log2 opacity, 1/opacity), cn = (A, B, C, unused)
__device__ __forceinline__ void stage_splat(float4 (&dst)[3], const float4& r0, const float4& r1, const float4& r2, float spare = 0.f) {
    const float L2E = 1.4426950408889634f;
    dst[0] = make_float4(r0.x, r0.y, __log2f(r0.z), __builtin_amdgcn_rcpf(r0.z)); // .w = 1/opacity (backward: scale of the opacity gradient)
    dst[1] = make_float4(-0.5f * L2E * r1.x, -L2E * r1.y, -0.5f * L2E * r1.z, spare); // .w: free for the caller (backward: the Gaussian's id)
    dst[2] = r2;
}
// log2(opacity * exp(-sigma)) at offset (dx, dy) from the splat centre
__device__ __forceinline__ float splat_exponent(const float4& a, const float4& cn, float dx, float dy) {
    return fmaf(dx, fmaf(cn.x, dx, cn.y * dy), fmaf(cn.z * dy, dy, a.z));
}

// ---------------------------------------------------------------------------------- forward
// One wavefront per 16x16 tile; lane l owns pixel l (8x8 raster order) of EACH of the four 8x8 quadrants,
// exactly like the backward below.  Against the earlier 4-waves-per-tile version this removes every
// workgroup barrier (single-wave __syncthreads is a free s_barrier), shares the scalar work of walking
// the hit mask and the three LDS broadcast reads of a splat between the quadrants it touches, and lets a
// finished quadrant drop out (its hit masks are no longer computed) while the rest of the tile goes on.
// The next group of 64 records is fetched from global memory while the current one is composited.
struct PixFwd {
    float T, o0, o1, o2, o3;
    int cur_idx;
    float best_vis; int best_idx; // MAIN_ID only
};

// QX x QY quadrants of 8x8 pixels per wave: 2 x 2 = gsplat's 16x16 tile, 4 x 2 = the WIDE internal tile (32x16, round 3).  A wide
// tile's list (raster_bin.hip, tile_range_wide) holds every Gaussian gsplat lists for one of its two 16x16 halves; a Gaussian that
// gsplat does not list for a pixel's own 16x16 tile cannot reach alpha >= 1/255 there (its radius box bounds exactly that region),
// and the per-quadrant test below rejects it, so every pixel composites the same splats in the same order as with 16x16 tiles:
// the forward is bit-identical, the backward differs only in the order its sums are formed.
//
// SUB: the wave serves ONE 8x8 quadrant (QX = 1) or one 16x8 half (QX = 2) of a 16x16 list tile -- four / two independent single-wave
// workgroups per tile, each walking the tile's list on its own.  For frames with fewer tiles than the chip has SIMDs (512x384: 768,
// 648x486: 1271, against 1024) the one-wave-per-tile form leaves SIMDs empty and the rest with a single, latency-bound wave; more waves per
// tile fill the chip and shorten every wave's serial chain to its own hits (split_parts below picks the form from the tile count).
// Per-pixel arithmetic and order are unchanged: bit-identical output in every form.
template <int QX, int QY, bool MAIN_ID, bool SUB = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(QX * QY == 8 ? 4 : 8, 8))) void raster_fwd_kernel(
    int tile_w, int tile_h, int W, int H, const float* __restrict__ rec, const int32_t* __restrict__ flatten_ids,
    const int32_t* __restrict__ offsets, int n_isects, const float* __restrict__ backgrounds,
    float* __restrict__ render_colors, float* __restrict__ render_alphas, float* __restrict__ final_T,
    int32_t* __restrict__ last_ids, int32_t* __restrict__ main_ids)
{
    static_assert(!SUB || (QX <= 2 && QY == 1), "SUB: a wave serves one quadrant or one 16x8 half of a 16x16 list tile");
    constexpr int NQ = QX * QY, TPW = SUB ? 16 : 8 * QX, TPH = SUB ? 16 : 8 * QY;
    constexpr int PARTS = SUB ? 4 / NQ : 1, PCOLS = SUB ? 2 / QX : 1; // footprints per list tile, and per row of it
    __shared__ float4 srec[64][3];
    const int n_tiles = tile_w * tile_h;
    // SUB: the parts of a tile are consecutive in the remapped order, i.e. on the same XCD (they read the same records)
    const int vtile = xcd_remap(blockIdx.x, PARTS * n_tiles);
    const int tile = vtile / PARTS, part = vtile % PARTS;
    const int tx = tile % tile_w, ty = tile / tile_w;
    const int px0 = tx * TPW + (part % PCOLS) * 8 * QX, py0 = ty * TPH + (part / PCOLS) * 8 * QY; // first pixel of the footprint
    const int lane = threadIdx.x;

    const int range_start = offsets[tile];
    const int range_end = (tile == n_tiles - 1) ? n_isects : offsets[tile + 1];

    PixFwd px[NQ];
    bool inside[NQ];
    unsigned live = 0u; // quadrants that still have an unfinished pixel (wave-uniform)
    unsigned long long done_m[NQ]; // wave-uniform lane masks (SGPR pairs): bit l = lane l's pixel of quadrant q is finished
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int ox = px0 + (q % QX) * 8, oy = py0 + (q / QX) * 8;
        const int pxi = ox + (lane & 7), pyi = oy + (lane >> 3);
        PixFwd& P = px[q];
        P.o0 = P.o1 = P.o2 = P.o3 = 0.f;
        P.T = 1.f; P.best_vis = 0.f; P.best_idx = -1;
        P.cur_idx = ADK_FWD_LEAN ? range_end - 1 : 0; // LEAN: the list index the backward starts at -- the tile's last unless the pixel stops earlier
        inside[q] = (pxi < W) && (pyi < H);
        done_m[q] = __builtin_amdgcn_ballot_w64(!inside[q]);
        if (__ballot(inside[q]) != 0ull) live |= 1u << q;
    }
    const float4* rec4 = reinterpret_cast<const float4*>(rec);
    // pixel centre in quadrant 0; quadrant q adds (8 (q % QX), 8 (q / QX)).  Register budget of the 16x16 form: 64 VGPRs, so that all
    // tiles of a 1080p frame (8160) are resident at once (8 waves/SIMD x 1024 SIMDs) and there is no second, half-empty round.
    const float fx0 = (float)(px0 + (lane & 7)) + 0.5f, fy0 = (float)(py0 + (lane >> 3)) + 0.5f;
    const float tox = (float)px0 + 0.5f, toy = (float)py0 + 0.5f; // centre of the footprint's first pixel

    for (int batch_start = range_start; batch_start < range_end && live; batch_start += 64) {
        const int batch_size = min(64, range_end - batch_start);
        const bool have = lane < batch_size;
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
        __syncthreads();
        if (have) {
            const int64_t g = flatten_ids[batch_start + lane];
            r0 = rec4[3 * g]; r1 = rec4[3 * g + 1];
            stage_splat(srec[lane], r0, r1, rec4[3 * g + 2]);
        }
        __syncthreads();
#if ADK_CLAMP_HOIST
        const unsigned long long clamp_m = __ballot(have && !(r0.z <= CLAMP_OPAC)); // staged splats whose alpha can reach the 0.999 clamp (NaN opacity: kept)
#endif
        // splat-parallel culling: this lane's splat against each live quadrant -> the lane's own quadrant bit set
#if ADK_CULL_BALLOT_FWD
        unsigned long long mq[NQ];
#else
        unsigned qmask = 0u;
#endif
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#if ADK_CULL_BALLOT_FWD
            mq[q] = 0ull;
#endif
            if ((live >> q) & 1u) {
                const float x0 = tox + (float)((q % QX) * 8), x1 = x0 + 7.0f, y0 = toy + (float)((q / QX) * 8), y1 = y0 + 7.0f;
                const bool hit = have && (r0.x + r0.w >= x0) && (r0.x - r0.w <= x1) && (r0.y + r1.w >= y0) && (r0.y - r1.w <= y1) &&
                                 splat_reaches_rect(r0.x, r0.y, r1.x, r1.y, r1.z, r0.z, x0, x1, y0, y1);
#if ADK_CULL_BALLOT_FWD
                mq[q] = __ballot(hit);
#else
                qmask |= hit ? (1u << q) : 0u;
#endif
            }
        }
#if ADK_CULL_BALLOT_FWD
        unsigned long long any = 0ull;
#pragma unroll
        for (int q = 0; q < NQ; ++q) any |= mq[q];
#else
        unsigned long long any = __ballot(qmask != 0u);
#endif
        while (any) {
            const int t = __builtin_ctzll(any);
            any &= any - 1;
#if ADK_CULL_BALLOT_FWD
            const unsigned long long bit = 1ull << t;
#define ADK_QHIT(q) (mq[q] & bit)
#else
            const unsigned qm = (unsigned)__builtin_amdgcn_readlane((int)qmask, t) & live; // wave-uniform: quadrants this splat can reach
            if (qm == 0u) continue;
#define ADK_QHIT(q) (qm & (1u << (q)))
#endif
            const float4 a = srec[t][0], cn = srec[t][1], col = srec[t][2];
            const int idx = batch_start + t;
            // mc (wave-uniform): this splat's opacity can reach 0.999, so min(0.999, alpha) is taken; for every other splat it is the identity and a
            // not-taken scalar branch skips it (asm volatile: as a select the compiler would if-convert it back into every evaluation)
#if ADK_CLAMP_HOIST
            const unsigned mc = (unsigned)(clamp_m >> t) & 1u;
#else
            const unsigned mc = 1u;
#endif
            {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (ADK_QHIT(q)) { // wave-uniform
                        // Every predicate lives as a wave-uniform 64-bit lane mask (one v_cmp each, combined on the scalar unit) and is
                        // applied as the mask operand of a v_cndmask: no 0/1 materialisation, no per-lane bit tests (a per-lane `done`
                        // bit field + `bool && bool` + __ballot compiled to v_and / v_cmp / v_cndmask 0,1 / v_or / v_cmp_ne chains:
                        // 24 VALU instructions per evaluated quadrant against 19; measured 0.270 -> 0.243 ms, round 2).  Measured and
                        // rejected: the two state updates as plain v_movs under EXEC = contrib_m (0.254 ms: EXEC writes stall).
                        PixFwd& P = px[q];
                        const float dx = a.x - (fx0 + (float)((q % QX) * 8)), dy = a.y - (fy0 + (float)((q / QX) * 8));
                        const float e = splat_exponent(a, cn, dx, dy);
                        float alpha = __builtin_amdgcn_exp2f(e);
                        if (__builtin_expect(scalar_flag(mc), !ADK_CLAMP_HOIST)) {
#if ADK_CLAMP_HOIST
                            asm volatile("s_nop 0\n\tv_min_f32_e32 %0, 0x3f7fbe77, %0" : "+v"(alpha)); // s_nop: v_exp result -> VALU use needs a wait state hipcc cannot see inside asm
#else
                            alpha = fminf(MAX_ALPHA, alpha);
#endif
                        }
                        const unsigned long long m_sig = __builtin_amdgcn_ballot_w64(!(e > a.z)); // e > log2(opacity) <=> sigma < 0
                        const unsigned long long m_thr = __builtin_amdgcn_ballot_w64(!(alpha < ALPHA_THR));
                        const unsigned long long valid_m = m_sig & m_thr & ~done_m[q];
#if ADK_FWD_LEAN
                        // ONE select: alpha zeroed in the lanes that do not blend this splat.  Such a lane keeps T (T * 1) and adds nothing (0 * T),
                        // and it cannot trip the terminate test: a pixel that is not finished has T > T_EPS, a finished one kept the T it had
                        // in front of the splat it stopped at.  A lane that DOES stop here is fixed up in the branch below (as often as a pixel
                        // finishes, i.e. once per pixel).  Same products in the same order as the three-select form: bit-identical output.
                        float alpha_v;
                        asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(alpha_v) : "v"(alpha), "s"(valid_m));
                        float next_T = P.T * (1.0f - alpha_v);
                        float vis = alpha_v * P.T;
                        const unsigned long long term_m = __builtin_amdgcn_ballot_w64(next_T <= T_EPS); // terminate BEFORE adding this splat
                        if (term_m != 0ull) {
                            asm("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(vis) : "s"(term_m));
                            asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(next_T) : "v"(P.T), "s"(term_m));
                            asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(P.cur_idx) : "v"(idx - 1), "s"(term_m)); // the backward starts in front of this splat
                            done_m[q] |= term_m;
                            if (~done_m[q] == 0ull) { // ... and it was the quadrant's last: later splats skip it
                                live &= ~(1u << q);
#if ADK_CULL_BALLOT_FWD
                                mq[q] = 0ull;
                                any = 0ull;
#pragma unroll
                                for (int qq = 0; qq < NQ; ++qq) any |= mq[qq];
                                any &= ~((bit << 1) - 1ull);
#endif
                            }
                        }
                        P.T = next_T;
                        P.o0 += col.x * vis; P.o1 += col.y * vis; P.o2 += col.z * vis; P.o3 += col.w * vis;
                        if (MAIN_ID && vis > P.best_vis) { P.best_vis = vis; P.best_idx = idx; }
#else
                        const float next_T = P.T * (1.0f - alpha);
                        const unsigned long long m_le = __builtin_amdgcn_ballot_w64(next_T <= T_EPS);
                        const unsigned long long term_m = valid_m & m_le;       // terminate BEFORE adding this splat
                        const unsigned long long contrib_m = valid_m & ~m_le;
                        float vis;
                        asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(vis) : "v"(alpha * P.T), "s"(contrib_m));
                        P.o0 += col.x * vis; P.o1 += col.y * vis; P.o2 += col.z * vis; P.o3 += col.w * vis;
                        if (MAIN_ID && vis > P.best_vis) { P.best_vis = vis; P.best_idx = idx; }
                        // cur_idx = idx, T = next_T in the contributing lanes (the wave is always full here: 64-thread block, only
                        // wave-uniform branches)
                        asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(P.cur_idx) : "v"(idx), "s"(contrib_m));
                        asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(P.T) : "v"(next_T), "s"(contrib_m));
                        if (term_m != 0ull) { // rare: some pixel just finished
                            done_m[q] |= term_m;
                            if (~done_m[q] == 0ull) { // ... and it was the quadrant's last: later splats skip it
                                live &= ~(1u << q);
#if ADK_CULL_BALLOT_FWD
                                mq[q] = 0ull;
                                any = 0ull;
#pragma unroll
                                for (int qq = 0; qq < NQ; ++qq) any |= mq[qq];
                                any &= ~((bit << 1) - 1ull);
#endif
                            }
                        }
#endif
                    }
                }
            }
            if (live == 0u) break;
        }
    }

#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (inside[q]) {
            PixFwd& P = px[q];
            const int pxi = px0 + (q % QX) * 8 + (lane & 7), pyi = py0 + (q / QX) * 8 + (lane >> 3);
            const int64_t pix = (int64_t)pyi * W + pxi;
            render_alphas[pix] = 1.0f - P.T;
            // The backward restarts its transmittance recurrence from T_final.  Recovering it as 1 - alpha (what upstream
            // does) loses up to 1e-4 RELATIVE when alpha is close to 1 (dense scenes: T_final ~ 1e-3, ulp(alpha) = 6e-8);
            // measured: every raster gradient was off by 1.3-1.5e-4 rel_l2 against the fp64 oracle because of it.
            final_T[pix] = P.T;
            float o0 = P.o0, o1 = P.o1, o2 = P.o2, o3 = P.o3;
            if (backgrounds) {
                o0 += P.T * backgrounds[0]; o1 += P.T * backgrounds[1]; o2 += P.T * backgrounds[2]; o3 += P.T * backgrounds[3];
            }
            reinterpret_cast<float4*>(render_colors)[pix] = make_float4(o0, o1, o2, o3);
            last_ids[pix] = P.cur_idx;
            if (MAIN_ID) main_ids[pix] = P.best_idx >= 0 ? flatten_ids[P.best_idx] : -1;
        }
    }
}

// ---------------------------------------------------------------------------------- backward
// Accumulator slots per splat: 0,1 sum v_sigma*(mean2d - pixel) (-> v_mean2d in project_bwd) | 2 v_opacity | 3,4,5 v_conic | 6..9 v_colour
#define NACC 10
__device__ __forceinline__ int acc_to_rec(int k) { return k < 3 ? k : (k < 6 ? k + 1 : k + 2); }

// Per-pixel backward state (one per quadrant the lane serves); the pixel centre is recomputed from the lane's quadrant-0 centre.
struct PixBwd {
#if ADK_BWD_LEAN
    float T, E;          // running transmittance; E = T_final*(v_alpha_out - <bg, v_render>) - <buffer, v_render>: the only combination the two were used in
#else
    float T, bdot, C0;   // running transmittance, <buffer, v_render>, T_final*(v_alpha_out - <bg, v_render>)
#endif
    float vr0, vr1, vr2, vr3;
    int bin_final;       // index of the last splat that contributed in the forward (-1: pixel outside the image)
};

// MEASURED on MI355X (tools/dpp_bench.hip): a DPP-modified VALU instruction issues at ~8.6 cycles per
// wave against 2.6 for a plain one (and a ds_bpermute shuffle+add pair at ~21).  Cross-lane reductions are
// therefore the most expensive thing this kernel does, and the design goal is ONE reduction per (splat,
// tile) instead of one per (splat, 8x8 quadrant):
//   * one wavefront per tile of QX x QY quadrants (2 x 2 = gsplat's 16x16 tile, 4 x 2 = the wide 32x16 internal tile of round 3:
//     26 % fewer (splat, tile) pairs on the bench cloud, hence 26 % fewer reductions / parkings / flush records);
//     lane l owns pixel l (8x8 raster order) of EACH quadrant;
//   * splat-parallel culling: every lane tests ITS staged splat against every quadrant and keeps the hits as a bit set in a
//     register; the wave walks the splats that hit anything and, per splat, reads that lane's bit set into an SGPR and evaluates
//     only those quadrants (wave-uniform branches), summing their contributions in the same 10 registers;
//   * the 10 sums are reduced once per (splat, tile): transposing DPP butterfly inside each 16-lane row (11 DPP), the
//     4 rows combined with v_permlane32_swap / v_permlane16_swap, and the 10 totals PARKED in an LDS table
//     sacc[staged splat][gradient-record dword]; after the batch of 64 staged splats the table is flushed with 16
//     wave instructions whose lanes cover 4 whole 48 B gradient records each.  Measured on MI355X (tools/lab, round 2):
//     one 10-lane atomic instruction per splat 0.716 ms -> parked + flushed 0.650 ms; a flush whose 64 lanes hit 64
//     DIFFERENT records was 2x SLOWER (1.55 ms: the memory side pays per cache line touched by an instruction, not per
//     lane), LDS ds_add_f32 for the row combine 0.85 ms.  Where the time goes (same measurements, ablations): cull test +
//     exponent + validity 0.22, gradient arithmetic 0.21, cross-lane reduction 0.18, atomics 0.09 ms.
// SUB: one wave per 8x8 quadrant / 16x8 half of a 16x16 list tile, as in the forward -- for frames with few tiles.  A splat then costs one
// reduction and one parked record per PART it contributes to instead of per tile: more total work, spread over more waves on a chip that
// was mostly idle (the sums are formed in a different order: gradients move by ~3e-7 relative).
#ifndef ADK_BWD_MINWAVES
#define ADK_BWD_MINWAVES 5   // 96 VGPRs, no scratch (99 -> 4 waves without it)
#endif
// Round 6: the halves form at 7 waves per SIMD (72 VGPRs, no scratch, once the parked totals share the staged records' LDS: 3 KB per wave instead of
// 6.4) -- 0.5177 -> 0.5078 ms at 1 M / 1080p; forced to 8 (64 VGPRs, 10 dwords of scratch) it LOSES 9 % there.  The kernel sits on the knee between
// issue-bound and latency-bound: an LDS-budget probe that left it 4 / 3 resident waves cost +13.6 % / +42 %, four padding VALU instructions per
// evaluated quadrant (+6.3 % instructions) +4.8 % whether of the 2.6-cycle or of the 4.2-cycle class (profiles/r06_bwd_probes.txt).
#ifndef ADK_BWD_MINWAVES_SUB
#define ADK_BWD_MINWAVES_SUB 7   // the halves / quadrants forms (SUB)
#endif
// WG2 (round 5, lab knob ADK_RASTER_BWD_WG2=1): the two 16x8 halves of a list tile as the two waves of ONE 128-thread workgroup instead of two
// single-wave workgroups -- the 64 splats of a batch are staged once, both waves park their totals in ONE shared table (LDS float adds) and
// the batch is flushed once: one 48 B atomic record per (splat, TILE) again instead of per (splat, half).  The price: three real workgroup
// barriers per batch between two waves whose hit counts differ.  Measured: DESIGN finding 42.
template <int QX, int QY, bool SUB = false, bool WG2 = false>
__global__ __launch_bounds__(WG2 ? 128 : 64) __attribute__((amdgpu_waves_per_eu(QX * QY == 8 ? 1 : (SUB ? ADK_BWD_MINWAVES_SUB : ADK_BWD_MINWAVES), 8))) void raster_bwd_kernel(
    int tile_w, int tile_h, int W, int H, const float* __restrict__ rec, const int32_t* __restrict__ flatten_ids,
    const int32_t* __restrict__ offsets, int n_isects, const float* __restrict__ backgrounds,
    const float* __restrict__ final_T, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_render_colors, const float* __restrict__ v_render_alphas,
    float* __restrict__ v_rec)
{
    static_assert(!SUB || (QX <= 2 && QY == 1), "SUB: a wave serves one quadrant or one 16x8 half of a 16x16 list tile");
    static_assert(!WG2 || (SUB && QX == 2 && QY == 1), "WG2: the two 16x8 halves of a tile as the two waves of one workgroup");
    constexpr int NQ = QX * QY, TPW = SUB ? 16 : 8 * QX, TPH = SUB ? 16 : 8 * QY;
    constexpr int PARTS = SUB ? 4 / NQ : 1, PCOLS = SUB ? 2 / QX : 1;
    // Round 6: ONE 3 KB table per wave.  A staged record's 12 dwords are dead once its splat has been evaluated (the wave holds them in registers), so
    // the splat's 10 totals are parked OVER its own record -- gradient-record dwords 0-2, 4-6, 8-11; dword 3 (1/opacity) is never written and dword 7
    // carries the Gaussian's id for the flush -- instead of in a second table + an id array (6.4 KB per wave capped the CU at 25 waves).
    // WG2 (lab form: two waves share the staging) keeps its own table.
    __shared__ float4 srec[64][3];
    __shared__ float sacc_wg2[WG2 ? 64 * 12 : 4];
    float (*sacc)[12] = WG2 ? reinterpret_cast<float (*)[12]>(&sacc_wg2[0]) : reinterpret_cast<float (*)[12]>(&srec[0][0]);
    __shared__ unsigned long long wg2_mask[2];
    __shared__ int wg2_final[2];
    const int n_tiles = tile_w * tile_h;
    const int wave = WG2 ? (int)(threadIdx.x >> 6) : 0;
    const int vtile = WG2 ? xcd_remap(blockIdx.x, n_tiles) * PARTS + wave : xcd_remap(blockIdx.x, PARTS * n_tiles);
    const int tile = vtile / PARTS, part = vtile % PARTS;
    const int tx = tile % tile_w, ty = tile / tile_w;
    const int px0 = tx * TPW + (part % PCOLS) * 8 * QX, py0 = ty * TPH + (part / PCOLS) * 8 * QY;
    const int lane = threadIdx.x & 63;

    const int range_start = offsets[tile];
    const int range_end = (tile == n_tiles - 1) ? n_isects : offsets[tile + 1];
    if (range_end <= range_start) return;

    float bgc[4] = {0.f, 0.f, 0.f, 0.f};
    if (backgrounds) { bgc[0] = backgrounds[0]; bgc[1] = backgrounds[1]; bgc[2] = backgrounds[2]; bgc[3] = backgrounds[3]; }

    // lane constants of the parking / flush steps.  Owner lane (row 0) -> the gradient-record dword of the total it holds after
    // wave_reduce10 and that dword's factor; flush lane (row r, dword d) -> bit r if dword d of a gradient record is live.
    int own_dword; float own_scale; bool own_is_opacity; unsigned flush_rowbit;
    {
        const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
        const int slot = (b3 ? 5 : 0) + (b2 ? (b1 ? 4 : 3) : (b1 ? 2 : (b0 ? 1 : 0)));
        own_dword = acc_to_rec(slot);
        own_is_opacity = own_dword == 2;
        own_scale = own_dword >= 8 ? 1.0f : ((own_dword == 4 || own_dword == 6) ? -0.5f : -1.0f);
        const int d = lane & 15;
        flush_rowbit = (d < 12 && d != 3 && d != 7) ? (1u << (lane >> 4)) : 0u;
    }
#if ADK_BWD_PAIR
    // lane roles after wave_reduce20 (lane = 16 r + 4 b + l): bank b serves splat b >> 1 and holds, in every lane of the bank, z0 = the total of
    // sum (b & 1) * 5 + {0, 2, 1, 3}[r] and z1 = the total of sum (b & 1) * 5 + 4.  Lane l = 0 of every (row, bank) parks z0, lane l = 1 of row 0 z1.
    int pk_dword; float pk_scale; bool pk_is_opacity, pk_active, pk_is_b, pk_use_z1;
    {
        const int r = lane >> 4, b = (lane >> 2) & 3, l = lane & 3;
        pk_use_z1 = (l == 1);
#if ADK_BWD_ROWS_FIRST
        // after wave_reduce20_rows_first: row r serves splat r >> 1 and sums (r & 1) * 5 + ..., bank b holds sum {0, 2, 1, 3}[b] (z0) and sum 4 (z1)
        pk_active = (l == 0) || (l == 1 && b == 0);
        pk_is_b = (r >> 1) != 0;
        const int slot = (r & 1) * 5 + (pk_use_z1 ? 4 : ((b == 1) ? 2 : (b == 2 ? 1 : b)));
#else
        pk_active = (l == 0) || (l == 1 && r == 0);
        pk_is_b = (b >> 1) != 0;
        const int slot = (b & 1) * 5 + (pk_use_z1 ? 4 : ((r == 1) ? 2 : (r == 2 ? 1 : r)));
#endif
        pk_dword = acc_to_rec(slot);
        pk_is_opacity = pk_dword == 2;
        pk_scale = pk_dword >= 8 ? 1.0f : ((pk_dword == 4 || pk_dword == 6) ? -0.5f : -1.0f);
    }
#endif
    PixBwd px[NQ];
    int quad_bin_final[NQ], tile_bin_final = -1;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int ox = px0 + (q % QX) * 8, oy = py0 + (q / QX) * 8;
        const int pxi = ox + (lane & 7), pyi = oy + (lane >> 3);
        PixBwd& P = px[q];
        P.vr0 = P.vr1 = P.vr2 = P.vr3 = 0.f; P.T = 1.f; P.bin_final = -1;
#if ADK_BWD_LEAN
        P.E = 0.f;
#else
        P.bdot = 0.f; P.C0 = 0.f;
#endif
        if (pxi < W && pyi < H) {
            const int64_t pix = (int64_t)pyi * W + pxi;
            const float T_final = final_T[pix];
            const float4 v = reinterpret_cast<const float4*>(v_render_colors)[pix];
            P.vr0 = v.x; P.vr1 = v.y; P.vr2 = v.z; P.vr3 = v.w;
            const float bg_dot = bgc[0] * v.x + bgc[1] * v.y + bgc[2] * v.z + bgc[3] * v.w;
#if ADK_BWD_LEAN
            P.E = T_final * (v_render_alphas[pix] - bg_dot);
#else
            P.C0 = T_final * (v_render_alphas[pix] - bg_dot);
#endif
            P.T = T_final;
            P.bin_final = last_ids[pix];
        }
        quad_bin_final[q] = wave_max_i(P.bin_final); // wave-uniform: nothing behind it can matter to this quadrant
        tile_bin_final = max(tile_bin_final, quad_bin_final[q]);
    }
    const float4* rec4 = reinterpret_cast<const float4*>(rec);
    const float fx0 = (float)(px0 + (lane & 7)) + 0.5f, fy0 = (float)(py0 + (lane >> 3)) + 0.5f;
    const float tox = (float)px0 + 0.5f, toy = (float)py0 + 0.5f;
    if constexpr (WG2) { // both waves walk the SAME batches (they share the staging and the barriers): the later of the two halves' last contributors
        if (lane == 0) wg2_final[wave] = tile_bin_final;
        for (int i = threadIdx.x; i < 64 * 12; i += 128) sacc_wg2[i] = 0.f; // parked totals are ADDED by both waves; the flush zeroes what it reads
        __syncthreads();
        tile_bin_final = max(wg2_final[0], wg2_final[1]);
    }

    // walk the tile's list back to front in groups of 64; groups entirely behind every pixel's last
    // contributor are skipped without being loaded
    const int first_end = min(range_end - 1, tile_bin_final);
    for (int batch_end = first_end; batch_end >= range_start; batch_end -= 64) {
        const int batch_size = min(64, batch_end + 1 - range_start);
        __syncthreads();
        int g = 0;
        bool have = lane < batch_size;
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
        if (have) {
            g = flatten_ids[batch_end - lane];
            r0 = rec4[3 * (int64_t)g]; r1 = rec4[3 * (int64_t)g + 1];
            if (!WG2 || wave == 0) { // WG2: one wave stages for both (the other still needs r0 / r1 for its own cull test)
                r2 = rec4[3 * (int64_t)g + 2];
                stage_splat(srec[lane], r0, r1, r2, __int_as_float(g));
            }
        }
        __syncthreads();
        unsigned long long touched_mask = 0ull; // staged splats whose totals were parked in sacc
#if ADK_CLAMP_HOIST_BWD
        const unsigned long long clamp_m = __ballot(have && !(r0.z <= CLAMP_OPAC)); // staged splats whose alpha can reach the 0.999 clamp
#endif
        // splat-parallel culling: this lane's splat against each quadrant -> the lane's own quadrant bit set
#if ADK_CULL_BALLOT_BWD
        unsigned long long mq[NQ];
#else
        unsigned qmask = 0u;
#endif
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float x0 = tox + (float)((q % QX) * 8), x1 = x0 + 7.0f, y0 = toy + (float)((q / QX) * 8), y1 = y0 + 7.0f;
            const bool hit = have && (batch_end - lane <= quad_bin_final[q]) &&
                             (r0.x + r0.w >= x0) && (r0.x - r0.w <= x1) && (r0.y + r1.w >= y0) && (r0.y - r1.w <= y1) &&
                             splat_reaches_rect(r0.x, r0.y, r1.x, r1.y, r1.z, r0.z, x0, x1, y0, y1);
#if ADK_CULL_BALLOT_BWD
            mq[q] = __ballot(hit);
#else
            qmask |= hit ? (1u << q) : 0u;
#endif
        }
#if ADK_CULL_BALLOT_BWD
        unsigned long long any = 0ull;
#pragma unroll
        for (int q = 0; q < NQ; ++q) any |= mq[q];
#else
        unsigned long long any = __ballot(qmask != 0u);
#endif
        // One staged splat against the quadrants it can reach: the 10 per-lane sums in acc, the splat's 1/opacity in inv_opac; false if
        // no pixel blended it (acc is then undefined and nothing is reduced or parked).
        // -Wsometimes-uninitialized is silenced for this lambda alone (not for the file): see c0..c9 below
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized"
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"
        auto eval_splat = [&](const int t, float (&acc)[NACC], float& inv_opac) -> bool {
#if ADK_CLAMP_HOIST_BWD
            const unsigned mc = (unsigned)(clamp_m >> t) & 1u; // wave-uniform: the splat's opacity can reach the 0.999 clamp (almost never)
#endif
#if ADK_CULL_BALLOT_BWD
            const unsigned long long bit = 1ull << t;
#define ADK_QHITB(q) (mq[q] & bit)
#else
            const unsigned qm = (unsigned)__builtin_amdgcn_readlane((int)qmask, t); // wave-uniform: quadrants this splat can reach
#define ADK_QHITB(q) (qm & (1u << (q)))
#endif
            const float4 a = srec[t][0], cn = srec[t][1], col = srec[t][2];
            inv_opac = a.w;
            const int idx = batch_end - t;
#if ADK_BWD_FIRST
            // FIRST-TOUCH form (round 4): the first quadrant that blends the splat WRITES its products (acc0..2 are t1, t2, gq themselves,
            // the other seven are plain v_mul), later ones accumulate: no 10 v_mov zero-fill per splat and no v_fmac onto a zero.
            // scalars, not an array (SROA turns a float[10] into one 10-register tuple and copies it whole), declared WITHOUT an initialiser.
            // The accumulate branch reads them only after a first-touch branch has written them (`touched`); the one place an indeterminate
            // value is ever touched is the copy-out at the end when NO quadrant blended the splat -- a copy whose result the caller discards
            // (`if (!eval_splat(...)) continue`).  Round 6 tried every defined spelling of "no value yet" and measured the code each gives for
            // the halves kernel (hipcc 7.2; the default build: 42 v_mov, 72 VGPRs, no scratch):
            //     = 0.f / any constant                      zero fills + copies at every merge (the form round 4 replaced)
            //     asm("" : "=v"(c)) empty definition        +40 v_mov
            //     __builtin_nondeterministic_value(c)       +40 v_mov (a frozen undef is a VALUE: the phi at each merge must copy it)
            //     copy-out under `if (touched)` / early return   +19 v_mov and 44 B of scratch
            // The register allocator keeps ONE register per sum through the first-touch / accumulate branch only when the incoming value is
            // absent, not when it is arbitrary.  So the declaration stays as it was, the diagnostic stays silenced FOR THIS LAMBDA ONLY, and
            // tests/test_codegen_guard.py pins what the optimiser makes of it (scratch 0, VGPR budget, v_mov count) so that a compiler
            // update that changes its mind fails a test instead of silently costing a wave per SIMD.
            float c0, c1, c2, c3, c4, c5, c6, c7, c8, c9;
#else
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
#endif
            int touched = 0; // wave-uniform; an int made uniform by readfirstlane so that `if (!touched)` below is a scalar branch (as a bool it was structurised into two ifs + copies)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (ADK_QHITB(q)) { // wave-uniform
                    PixBwd& P = px[q];
                    // Branch-free body: an invalid lane gets ov = 0 => alpha = 0, ra = 1, fac = 0 and every
                    // gradient term vanishes on its own.
                    const float dx = a.x - (fx0 + (float)((q % QX) * 8)), dy = a.y - (fy0 + (float)((q / QX) * 8));
                    const float e = splat_exponent(a, cn, dx, dy);
                    float ov = __builtin_amdgcn_exp2f(e); // opacity * exp(-sigma)
                    const bool valid = (idx <= P.bin_final) && !(e > a.z) && !(ov < ALPHA_THR);
                    if (__ballot(valid) == 0ull) continue; // nobody in this quadrant blended it (all finished earlier / below 1/255)
#if ADK_BWD_LEAN
                    // LEAN form (round 4): validity folded into ov ONCE (an invalid lane then has alpha = 0, ra = 1, fac = 0, gq = 0 on its own),
                    // the <buffer, v_render> recurrence kept as E = C0 - bdot (one register and one v_sub less per evaluation), and the
                    // alpha > 0.999 clamp -- which passes no gradient -- handled in a wave-uniform branch that is almost never taken.
                    float ov_v = valid ? ov : 0.f;
#if ADK_CLAMP_HOIST_BWD
                    // min(0.999, .) IN PLACE and only for a splat that can reach it (not-taken scalar branch): where alpha is not clamped it IS ov_v, where
                    // it is clamped the gradient is zeroed below, so one register serves as both (a separate alpha cost a v_mov per evaluation)
                    unsigned long long clamped_m = 0ull;
                    if (__builtin_expect(scalar_flag(mc), 0)) {
                        clamped_m = __ballot(ov_v > MAX_ALPHA);
                        asm volatile("v_min_f32_e32 %0, 0x3f7fbe77, %0" : "+v"(ov_v));
                    }
                    const float alpha = ov_v;
#else
                    float alpha;
                    asm("v_min_f32_e32 %0, 0x3f7fbe77, %1" : "=v"(alpha) : "v"(ov_v));
#endif
                    const float ra = __builtin_amdgcn_rcpf(1.0f - alpha); // 1 ulp v_rcp_f32; alpha <= 0.999
                    P.T *= ra;
                    const float fac = alpha * P.T;
                    const float S1 = col.x * P.vr0 + col.y * P.vr1 + col.z * P.vr2 + col.w * P.vr3;
                    const float v_alpha = P.T * S1 + ra * P.E;
                    P.E -= fac * S1;
                    float gq = ov_v * v_alpha;   // opacity * vis * v_alpha = -v_sigma
#if ADK_CLAMP_HOIST_BWD
                    if (__builtin_expect(scalar_flag(mc), 0) && clamped_m != 0ull) { // clamped alpha passes no gradient (opacity > 0.999 at the splat centre: almost never)
                        asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(gq) : "s"(clamped_m));
                    }
#else
                    if (__ballot(ov_v > MAX_ALPHA) != 0ull) { // clamped alpha passes no gradient (opacity > 0.999 at the splat centre: almost never)
                        asm volatile("; clamped alpha"); // keeps this a real, not-taken branch (if-converted it costs two compares and a select on every evaluation)
                        gq = (ov_v > MAX_ALPHA) ? 0.f : gq;
                    }
#endif
#else
                    float alpha_raw; // min(0.999, ov) on the v_exp result itself (fminf() puts a canonicalising v_max in front)
                    asm("s_nop 0\n\tv_min_f32_e32 %0, 0x3f7fbe77, %1" : "=v"(alpha_raw) : "v"(ov)); // s_nop: v_exp result -> VALU use needs 1 wait state, invisible to hipcc inside asm
                    const float alpha = valid ? alpha_raw : 0.f;
                    const float ra = __builtin_amdgcn_rcpf(1.0f - alpha); // 1 ulp v_rcp_f32; alpha <= 0.999
                    P.T *= ra;
                    const float fac = alpha * P.T;
                    const float S1 = col.x * P.vr0 + col.y * P.vr1 + col.z * P.vr2 + col.w * P.vr3;
                    const float v_alpha = P.T * S1 + ra * (P.C0 - P.bdot);
                    P.bdot += fac * S1;
                    // gq = opacity * vis * v_alpha = -v_sigma (clamped alpha passes no gradient); the opacity gradient is
                    // vis * v_alpha = gq / opacity, divided once per splat at the flush
                    const float gq = (valid && ov <= MAX_ALPHA) ? ov * v_alpha : 0.f;
#endif
#if ADK_BWD_FIRST
                    // Both forms are written as inline asm on the SAME tied ("+v") operands so that the register allocator keeps one physical
                    // register per sum through the branch (as plain C++ the products landed in fresh registers and were copied: 10 v_mov_b64).
#define ADK_MUL(d, x, y) asm("v_mul_f32_e32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define ADK_FMAC(d, x, y) asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(d) : "v"(x), "v"(y))
                    if (!touched) {
                        ADK_MUL(c0, gq, dx); ADK_MUL(c1, gq, dy);
                        c2 = gq; // plain assignment: coalesced into gq's register (as asm it was a v_mov onto itself)
                        ADK_MUL(c3, c0, dx); ADK_MUL(c4, c0, dy); ADK_MUL(c5, c1, dy);
                        ADK_MUL(c6, fac, P.vr0); ADK_MUL(c7, fac, P.vr1); ADK_MUL(c8, fac, P.vr2); ADK_MUL(c9, fac, P.vr3);
                    } else {
                        const float t1 = gq * dx, t2 = gq * dy;
                        asm("v_add_f32_e32 %0, %0, %1" : "+v"(c0) : "v"(t1));
                        asm("v_add_f32_e32 %0, %0, %1" : "+v"(c1) : "v"(t2));
                        asm("v_add_f32_e32 %0, %0, %1" : "+v"(c2) : "v"(gq));
                        ADK_FMAC(c3, t1, dx); ADK_FMAC(c4, t1, dy); ADK_FMAC(c5, t2, dy);
                        ADK_FMAC(c6, fac, P.vr0); ADK_FMAC(c7, fac, P.vr1); ADK_FMAC(c8, fac, P.vr2); ADK_FMAC(c9, fac, P.vr3);
                    }
#undef ADK_MUL
#undef ADK_FMAC
                    touched = __builtin_amdgcn_readfirstlane(1);
#else
                    const float t1 = gq * dx, t2 = gq * dy;
                    touched = 1;
                    acc[0] += t1;                      // v_mean2d = conic (acc[0], acc[1])^T is formed by project_bwd,
                    acc[1] += t2;                      //   once per Gaussian instead of once per (splat, pixel)
                    acc[2] += gq;                      // opacity * v_opacity
                    acc[3] += t1 * dx;                 // -2 * v_conic.a
                    acc[4] += t1 * dy;                 // -v_conic.b
                    acc[5] += t2 * dy;                 // -2 * v_conic.c
                    acc[6] += fac * P.vr0; acc[7] += fac * P.vr1; acc[8] += fac * P.vr2; acc[9] += fac * P.vr3;
#endif
                }
            }
#if ADK_BWD_FIRST
            acc[0] = c0; acc[1] = c1; acc[2] = c2; acc[3] = c3; acc[4] = c4; acc[5] = c5; acc[6] = c6; acc[7] = c7; acc[8] = c8; acc[9] = c9;
#endif
            return touched != 0;
        };
#pragma clang diagnostic pop

        while (any) {
            const int t = __builtin_ctzll(any); // staged slot t holds list index batch_end - t (0 = furthest back)
            any &= any - 1;
            float acc[NACC], inv_opac;
            if (!eval_splat(t, acc, inv_opac)) continue; // every sum is zero: no reduction, nothing parked
#if ADK_BWD_PAIR
            // PAIRED reduction (round 4): look for the next splat that blends anything and reduce BOTH splats' 20 sums in one butterfly
            // (wave_reduce20: 46 cross-lane instructions per pair against 2 x 32); an odd splat out takes the single reduction.
            int t2 = 0;
            float acc2[NACC], inv_opac2 = 0.f;
            bool paired = false;
            while (any) {
                t2 = __builtin_ctzll(any);
                any &= any - 1;
                if (eval_splat(t2, acc2, inv_opac2)) { paired = true; break; }
            }
            if (paired) {
#if ADK_BWD_ROWS_FIRST
                const Reduce20 red = wave_reduce20_rows_first(acc, acc2);
#else
                const Reduce20 red = wave_reduce20(acc, acc2);
#endif
                touched_mask |= (1ull << t) | (1ull << t2);
                if (pk_active) {
                    const float v = pk_use_z1 ? red.z1 : red.z0;
                    const float io = pk_is_b ? inv_opac2 : inv_opac;
                    const float pv = v * (pk_is_opacity ? io : pk_scale);
                    if constexpr (WG2) atomicAdd(&sacc[pk_is_b ? t2 : t][pk_dword], pv); // the other half's wave parks into the same table
                    else sacc[pk_is_b ? t2 : t][pk_dword] = pv;
                }
                continue;
            }
#endif
            // ONE 64-lane reduction per (splat, tile); the 10 lanes of row 0 that own a total park it in the LDS table
            const Reduce10 red = wave_reduce10(acc, lane);
            touched_mask |= 1ull << t;
            // per-splat factors applied by the owner lane as it parks its total: -1 on the v_sigma-weighted dwords, -1/2 on the
            // symmetric conic entries, 1/opacity (staged in a.w) on the opacity gradient
            if (red.is_owner) {
                const float pv = red.value * (own_is_opacity ? inv_opac : own_scale);
                if constexpr (WG2) atomicAdd(&sacc[t][own_dword], pv);
                else sacc[t][own_dword] = pv;
            }
        }
        // Flush the batch: one instruction = 4 staged splats x the 12 dwords of their gradient records (10 live), so its
        // atomics fall into 4 records.  The per-splat factors were applied when the totals were parked, the lane's role
        // (dword live?) is a kernel-lifetime constant and the "was this splat touched" test is one bit test against a
        // scalar nibble: 2 VALU + 2 LDS reads + 1 atomic per instruction (a per-lane 64-bit shift, three compares on the
        // dword index, an LDS read of the opacity and a v_rcp per instruction before: 0.599 -> 0.580 ms).
        if constexpr (WG2) { // both waves' marks, then ONE flush of the batch, its 16 instructions dealt out between the two waves
            if (lane == 0) wg2_mask[wave] = touched_mask;
            __syncthreads();
            touched_mask = wg2_mask[0] | wg2_mask[1];
        }
        if (WG2 || touched_mask) {
            if constexpr (!WG2) __syncthreads();
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                if (WG2 && (j & 1) != wave) continue;
                const unsigned nib = (unsigned)(touched_mask >> (4 * j)) & 0xFu; // scalar: the 4 splats this instruction covers
                if (!nib) continue;
                if (nib & flush_rowbit) { // this lane's dword is live and its splat was touched
                    const int sp = 4 * j + (lane >> 4);
                    const float total = sacc[sp][lane & 15];
                    if constexpr (WG2) sacc[sp][lane & 15] = 0.f; // the table is ready for the next batch's adds
                    if (total != 0.f) unsafeAtomicAdd(v_rec + 12 * (int64_t)__float_as_int(srec[sp][1].w) + (lane & 15), total);
                }
            }
        }
    }
}

} // namespace adk

// How many waves serve a 16x16 list tile: 1 (the whole tile), 2 (its 16x8 halves) or 4 (its 8x8 quadrants).  A tile is one wave, the chip
// has 1 024 SIMDs with 8 wave slots each: with few tiles the one-wave form leaves SIMDs empty or with a single latency-bound wave, and
// splitting shortens every wave's serial chain to its own hits; with many tiles the duplicated staging / culling (and, backward, one
// reduction + one parked record per PART a splat contributes to instead of per tile) costs more than it hides.  Same-box A/B on MI355X
// (tools/lab/ab_split.py -> profiles/r03_ab_split.json; ms, tile / half / quadrant):
//     tiles      768 (1 M, 512x384)   1 271 (648x486)    1 900             3 072             5 700             8 160 (1080p)      19 764 (4 M)
//     forward    .127 / .079 / .054   .160 / .103 / .087  .135 / .092 / .085  .155 / .113 / .113  .228 / .197 / .212  .259 / .246 / .288  .516 / .487 / .606
//     backward   .189 / .120 / .104   .236 / .176 / .171  .223 / .168 / .181  .259 / .229 / .240  .439 / .409 / .455  .543 / .558 / .630  1.06 / 1.09 / 1.33
// Round 4, with the backward's first-touch / paired-reduction / lean body (profiles/r04_ab_split.json; a reduction per PART is cheaper now, so
// the halves win further up):
//     backward   .199 / .126 / .104   .243 / .166 / .156  .221 / .164 / .173  .253 / .206 / .225  .434 / .375 / .421  .515 / .502 / .582  .996 / .996 / 1.23
// Measured with the split forms and dropped: requesting the NEXT group's list entries + records before the current group is walked (13
// more VGPRs): forward +-1 %, backward 0.160 -> 0.168 ms in the quadrant form at 200 k / 648x486 -- the waves do not wait on those loads.
// ADK_RASTER_SPLIT_FWD / _BWD = 0 (tile) | 2 (halves) | 1 (quadrants) force a form (read per launch: the labs and tests flip it in-process).
static int split_parts(int n_tiles, bool bwd) {
    const char* e = getenv(bwd ? "ADK_RASTER_SPLIT_BWD" : "ADK_RASTER_SPLIT_FWD");
    if (e && (e[0] == '0' || e[0] == '1' || e[0] == '2')) return e[0] == '0' ? 1 : (e[0] == '1' ? 4 : 2);
    if (bwd) return n_tiles < 1600 ? 4 : (n_tiles < 20000 ? 2 : 1);   // round 4 (paired reduction): the halves win or tie up to 19 764 tiles, see the table above
    return n_tiles < 3072 ? 4 : 2;
}

// ADK_RASTER_BWD_WG2=1 (read per launch): the two-halves backward as one 128-thread workgroup per tile (lab form, DESIGN finding 42)
static bool wg2_form() { const char* e = getenv("ADK_RASTER_BWD_WG2"); return e && e[0] == '1'; }

// render_colors [H,W,4], render_alphas [H,W], final_T [H,W] (exact final transmittance, consumed by adk_raster_bwd),
// last_ids [H,W]; backgrounds [4] or NULL; main_ids [H,W] (Gaussian id with the largest alpha*T per pixel, -1 if none) or NULL.
// tile_px_w x tile_px_h: the tile shape the lists (flatten_ids / offsets) were binned for: 16x16 (gsplat's) or 32x16.
extern "C" int adk_raster_fwd_t(int width, int height, int tile_px_w, int tile_px_h, const float* rec, const int32_t* flatten_ids,
                                const int32_t* offsets, int64_t n_isects, const float* backgrounds,
                                float* render_colors, float* render_alphas, float* final_T, int32_t* last_ids,
                                int32_t* main_ids, hipStream_t stream)
{
    if (width <= 0 || height <= 0 || n_isects < 0 || n_isects >= ((int64_t)1 << 31)) return ADK_EINVAL;
    if (!offsets || !render_colors || !render_alphas || !final_T || !last_ids) return ADK_EINVAL;
    if (n_isects > 0 && (!rec || !flatten_ids)) return ADK_EINVAL;
    if (((uintptr_t)rec & 15) || ((uintptr_t)render_colors & 15)) return ADK_EINVAL;
    const bool wide = tile_px_w == 32 && tile_px_h == 16;
    if (!wide && !(tile_px_w == 16 && tile_px_h == 16)) return ADK_EUNSUPPORTED;
    const int tile_w = (width + tile_px_w - 1) / tile_px_w, tile_h = (height + tile_px_h - 1) / tile_px_h;
#define ADK_FWD(QX, QY, MID) hipLaunchKernelGGL((adk::raster_fwd_kernel<QX, QY, MID>), dim3(tile_w * tile_h), dim3(64), 0, stream, tile_w, tile_h, \
        width, height, rec, flatten_ids, offsets, (int)n_isects, backgrounds, render_colors, render_alphas, final_T, last_ids, main_ids)
    if (wide) { if (main_ids) ADK_FWD(4, 2, true); else ADK_FWD(4, 2, false); }
    else {
#define ADK_FWD_SUB(QX, MID) hipLaunchKernelGGL((adk::raster_fwd_kernel<QX, 1, MID, true>), dim3(4 / QX * tile_w * tile_h), dim3(64), 0, stream, tile_w, \
        tile_h, width, height, rec, flatten_ids, offsets, (int)n_isects, backgrounds, render_colors, render_alphas, final_T, last_ids, main_ids)
        const int parts = split_parts(tile_w * tile_h, false);
        if (parts == 4) { if (main_ids) ADK_FWD_SUB(1, true); else ADK_FWD_SUB(1, false); }
        else if (parts == 2) { if (main_ids) ADK_FWD_SUB(2, true); else ADK_FWD_SUB(2, false); }
        else { if (main_ids) ADK_FWD(2, 2, true); else ADK_FWD(2, 2, false); }
#undef ADK_FWD_SUB
    }
#undef ADK_FWD
    ADK_RETURN_LAST_ERROR();
}
extern "C" int adk_raster_fwd(int width, int height, const float* rec, const int32_t* flatten_ids,
                              const int32_t* offsets, int64_t n_isects, const float* backgrounds,
                              float* render_colors, float* render_alphas, float* final_T, int32_t* last_ids,
                              int32_t* main_ids, hipStream_t stream)
{
    return adk_raster_fwd_t(width, height, 16, 16, rec, flatten_ids, offsets, n_isects, backgrounds, render_colors, render_alphas, final_T,
                            last_ids, main_ids, stream);
}

// v_rec [N,12] must be zero-initialised by the caller; gradients are accumulated into it.
extern "C" int adk_raster_bwd_t(int width, int height, int tile_px_w, int tile_px_h, const float* rec, const int32_t* flatten_ids,
                                const int32_t* offsets, int64_t n_isects, const float* backgrounds,
                                const float* final_T, const int32_t* last_ids, const float* v_render_colors,
                                const float* v_render_alphas, float* v_rec, hipStream_t stream)
{
    if (width <= 0 || height <= 0 || n_isects < 0 || n_isects >= ((int64_t)1 << 31)) return ADK_EINVAL;
    if (n_isects == 0) return 0;
    if (!rec || !flatten_ids || !offsets || !final_T || !last_ids || !v_render_colors || !v_render_alphas || !v_rec) return ADK_EINVAL;
    if (((uintptr_t)rec & 15) || ((uintptr_t)v_render_colors & 15)) return ADK_EINVAL;
    const bool wide = tile_px_w == 32 && tile_px_h == 16;
    if (!wide && !(tile_px_w == 16 && tile_px_h == 16)) return ADK_EUNSUPPORTED;
    const int tile_w = (width + tile_px_w - 1) / tile_px_w, tile_h = (height + tile_px_h - 1) / tile_px_h;
    if (wide)
        hipLaunchKernelGGL((adk::raster_bwd_kernel<4, 2>), dim3(tile_w * tile_h), dim3(64), 0, stream, tile_w, tile_h, width, height,
                           rec, flatten_ids, offsets, (int)n_isects, backgrounds, final_T, last_ids, v_render_colors, v_render_alphas, v_rec);
    else if (const int parts = split_parts(tile_w * tile_h, true); parts == 4)
        hipLaunchKernelGGL((adk::raster_bwd_kernel<1, 1, true>), dim3(4 * tile_w * tile_h), dim3(64), 0, stream, tile_w, tile_h, width, height,
                           rec, flatten_ids, offsets, (int)n_isects, backgrounds, final_T, last_ids, v_render_colors, v_render_alphas, v_rec);
    else if (parts == 2 && wg2_form())
        hipLaunchKernelGGL((adk::raster_bwd_kernel<2, 1, true, true>), dim3(tile_w * tile_h), dim3(128), 0, stream, tile_w, tile_h, width, height,
                           rec, flatten_ids, offsets, (int)n_isects, backgrounds, final_T, last_ids, v_render_colors, v_render_alphas, v_rec);
    else if (parts == 2)
        hipLaunchKernelGGL((adk::raster_bwd_kernel<2, 1, true>), dim3(2 * tile_w * tile_h), dim3(64), 0, stream, tile_w, tile_h, width, height,
                           rec, flatten_ids, offsets, (int)n_isects, backgrounds, final_T, last_ids, v_render_colors, v_render_alphas, v_rec);
    else
        hipLaunchKernelGGL((adk::raster_bwd_kernel<2, 2>), dim3(tile_w * tile_h), dim3(64), 0, stream, tile_w, tile_h, width, height,
                           rec, flatten_ids, offsets, (int)n_isects, backgrounds, final_T, last_ids, v_render_colors, v_render_alphas, v_rec);
    ADK_RETURN_LAST_ERROR();
}
extern "C" int adk_raster_bwd(int width, int height, const float* rec, const int32_t* flatten_ids,
                              const int32_t* offsets, int64_t n_isects, const float* backgrounds,
                              const float* final_T, const int32_t* last_ids, const float* v_render_colors,
                              const float* v_render_alphas, float* v_rec, hipStream_t stream)
{
    return adk_raster_bwd_t(width, height, 16, 16, rec, flatten_ids, offsets, n_isects, backgrounds, final_T, last_ids, v_render_colors,
                            v_render_alphas, v_rec, stream);
}
