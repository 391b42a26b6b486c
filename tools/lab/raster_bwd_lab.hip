// DEV TOOL (not part of libartdeco_hip.so): ablations and candidate designs of raster_bwd_kernel, timed on the real
// workload by tools/lab/run_bwd_lab.py.  Each variant is the product kernel (artdeco_amd/csrc/raster_tiles.hip) with one
// thing removed or replaced, selected at compile time by -DLAB_VARIANT=n and exported as lab_raster_bwd:
//   0  product kernel as is
//   1  no atomics        (the reduced totals go to a per-lane sink that is stored once at the end)
//   2  no reduction, no atomics (the 10 per-lane sums go to the sink)
//   3  cull test + sigma/exp/validity only (no gradient math, no reduction)
//   4  reduction through LDS (transpose: 10 ds_write_b32 per lane, 10 lanes x 4 partial columns, DPP finish)
//   5  row butterfly (11 DPP) then the 4 row sums of every slot go to an LDS table sacc[64 splats][10] with ds_add_f32;
//      after the batch of 64 staged splats the table is flushed with 10 FULL-WAVE atomic instructions (lane = splat)
//      instead of one 10-lane atomic instruction per splat
//   6  as 5 but the 4 rows are combined in registers with v_permlane32_swap / v_permlane16_swap and 10 lanes ds_write
//   7  as 6 + hand-scheduled body (exp2 with folded log2 constants, opacity folded into the exponent)
// The answer to "where do the 0.73 ms go" is the difference between consecutive variants.
#include "adk_common.hpp"

#ifndef LAB_VARIANT
#define LAB_VARIANT 0
#endif

namespace adk {

#define TILE 16
#define MAX_ALPHA 0.999f
#define ALPHA_THR (1.0f / 255.0f)
#define NACC 10

__device__ __forceinline__ bool splat_reaches_rect(float mx, float my, float a, float b, float c, float opac,
                                                   float x0, float x1, float y0, float y1)
{
    const float dxl = mx - x1, dxh = mx - x0, dyl = my - y1, dyh = my - y0;
    if (dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f) return true;
    const float tau = __logf(opac * 255.0f) + 1e-3f;
    const float nb_c = -b * __builtin_amdgcn_rcpf(c), nb_a = -b * __builtin_amdgcn_rcpf(a);
    float best;
    { const float dy = fminf(fmaxf(nb_c * dxl, dyl), dyh); best = 0.5f * (a * dxl * dxl + c * dy * dy) + b * dxl * dy; }
    { const float dy = fminf(fmaxf(nb_c * dxh, dyl), dyh); best = fminf(best, 0.5f * (a * dxh * dxh + c * dy * dy) + b * dxh * dy); }
    { const float dx = fminf(fmaxf(nb_a * dyl, dxl), dxh); best = fminf(best, 0.5f * (a * dx * dx + c * dyl * dyl) + b * dx * dyl); }
    { const float dx = fminf(fmaxf(nb_a * dyh, dxl), dxh); best = fminf(best, 0.5f * (a * dx * dx + c * dyh * dyh) + b * dx * dyh); }
    return best <= tau;
}

__device__ __forceinline__ int acc_to_rec(int k) { return k < 3 ? k : (k < 6 ? k + 1 : k + 2); }

struct PixBwd {
    float fx, fy, T, bdot, C0, vr0, vr1, vr2, vr3;
    int bin_final;
};

#if LAB_VARIANT == 8
#define LAB_OCC __attribute__((amdgpu_waves_per_eu(6, 6)))
#else
#define LAB_OCC
#endif
__global__ __launch_bounds__(64) LAB_OCC void lab_bwd_kernel(
    int tile_w, int tile_h, int W, int H, const float* __restrict__ rec, const int32_t* __restrict__ flatten_ids,
    const int32_t* __restrict__ offsets, int n_isects, const float* __restrict__ backgrounds,
    const float* __restrict__ final_T, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_render_colors, const float* __restrict__ v_render_alphas,
    float* __restrict__ v_rec, float* __restrict__ sink_out)
{
    __shared__ float4 srec[64][3];
    __shared__ int sid[64];
#if LAB_VARIANT == 4
    __shared__ float sred[NACC][65];
#endif
#if LAB_VARIANT >= 5
    __shared__ float sacc[64][12]; // [staged splat][gradient-record dword]: a flush instruction covers 4 whole records
#endif
    const int n_tiles = tile_w * tile_h;
    const int tile = xcd_remap(blockIdx.x, n_tiles);
    const int tx = tile % tile_w, ty = tile / tile_w;
    const int lane = threadIdx.x;
    float sink = 0.f;

    const int range_start = offsets[tile];
    const int range_end = (tile == n_tiles - 1) ? n_isects : offsets[tile + 1];
    if (range_end <= range_start) return;

    float bgc[4] = {0.f, 0.f, 0.f, 0.f};
    if (backgrounds) { bgc[0] = backgrounds[0]; bgc[1] = backgrounds[1]; bgc[2] = backgrounds[2]; bgc[3] = backgrounds[3]; }

    PixBwd px[4];
    float qx0[4], qy0[4];
    int quad_bin_final[4], tile_bin_final = -1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ox = tx * TILE + (q & 1) * 8, oy = ty * TILE + (q >> 1) * 8;
        qx0[q] = (float)ox + 0.5f; qy0[q] = (float)oy + 0.5f;
        const int pxi = ox + (lane & 7), pyi = oy + (lane >> 3);
        PixBwd& P = px[q];
        P.fx = (float)pxi + 0.5f; P.fy = (float)pyi + 0.5f;
        P.vr0 = P.vr1 = P.vr2 = P.vr3 = 0.f; P.T = 1.f; P.bdot = 0.f; P.C0 = 0.f; P.bin_final = -1;
        if (pxi < W && pyi < H) {
            const int64_t pix = (int64_t)pyi * W + pxi;
            const float T_final = final_T[pix];
            const float4 v = reinterpret_cast<const float4*>(v_render_colors)[pix];
            P.vr0 = v.x; P.vr1 = v.y; P.vr2 = v.z; P.vr3 = v.w;
            const float bg_dot = bgc[0] * v.x + bgc[1] * v.y + bgc[2] * v.z + bgc[3] * v.w;
            P.C0 = T_final * (v_render_alphas[pix] - bg_dot);
            P.T = T_final;
            P.bin_final = last_ids[pix];
        }
        quad_bin_final[q] = wave_max_i(P.bin_final);
        tile_bin_final = max(tile_bin_final, quad_bin_final[q]);
    }
    const float4* rec4 = reinterpret_cast<const float4*>(rec);

    const int first_end = min(range_end - 1, tile_bin_final);
    for (int batch_end = first_end; batch_end >= range_start; batch_end -= 64) {
        const int batch_size = min(64, batch_end + 1 - range_start);
        __syncthreads();
        int g = 0;
        bool have = lane < batch_size;
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
        if (have) {
            g = flatten_ids[batch_end - lane];
            r0 = rec4[3 * (int64_t)g]; r1 = rec4[3 * (int64_t)g + 1]; r2 = rec4[3 * (int64_t)g + 2];
#if LAB_VARIANT >= 7 || defined(LAB_EXP2)
            {   // pre-scaled per-splat constants: exponent of 2 instead of e, opacity folded in as log2(opacity)
                const float L2E = 1.4426950408889634f;
                sid[lane] = g;
                srec[lane][0] = make_float4(r0.x, r0.y, __log2f(r0.z), r0.z);
                srec[lane][1] = make_float4(-0.5f * L2E * r1.x, -L2E * r1.y, -0.5f * L2E * r1.z, 0.f);
                srec[lane][2] = r2;
            }
#else
            sid[lane] = g; srec[lane][0] = r0; srec[lane][1] = r1; srec[lane][2] = r2;
#endif
        }
#if LAB_VARIANT >= 5
#if LAB_VARIANT == 5
#pragma unroll
        for (int k = 0; k < 12; ++k) (&sacc[0][0])[k * 64 + lane] = 0.f;
#endif
        unsigned long long touched_mask = 0ull;
#endif
        __syncthreads();
        unsigned long long mq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x0 = qx0[q], x1 = x0 + 7.0f, y0 = qy0[q], y1 = y0 + 7.0f;
            const bool hit = have && (batch_end - lane <= quad_bin_final[q]) &&
                             (r0.x + r0.w >= x0) && (r0.x - r0.w <= x1) && (r0.y + r1.w >= y0) && (r0.y - r1.w <= y1) &&
                             splat_reaches_rect(r0.x, r0.y, r1.x, r1.y, r1.z, r0.z, x0, x1, y0, y1);
            mq[q] = __ballot(hit);
        }
        unsigned long long any = (mq[0] | mq[1]) | (mq[2] | mq[3]);
        while (any) {
            const int t = __builtin_ctzll(any);
            const unsigned long long bit = 1ull << t;
            any &= any - 1;
            const float4 a = srec[t][0], cn = srec[t][1], col = srec[t][2];
#if LAB_VARIANT >= 7 || defined(LAB_EXP2)
            const float opac = a.w;
#else
            const float opac = a.z;
#endif
            const int idx = batch_end - t;
            float acc[NACC];
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
            bool touched = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (mq[q] & bit) {
                    PixBwd& P = px[q];
                    const float dx = a.x - P.fx, dy = a.y - P.fy;
#if LAB_VARIANT >= 7 || defined(LAB_EXP2)
                    const float e = fmaf(dx, fmaf(cn.x, dx, cn.y * dy), fmaf(cn.z * dy, dy, a.z)); // log2(opacity * exp(-sigma))
                    float ov = __builtin_amdgcn_exp2f(e);
                    const bool valid = (idx <= P.bin_final) && !(e > a.z) && !(ov < ALPHA_THR);
                    if (__ballot(valid) == 0ull) continue;
                    touched = true;
                    ov = valid ? ov : 0.f;
                    const float alpha = fminf(MAX_ALPHA, ov);
                    const float ra = __builtin_amdgcn_rcpf(1.0f - alpha);
                    P.T *= ra;
                    const float fac = alpha * P.T;
                    const float S1 = col.x * P.vr0 + col.y * P.vr1 + col.z * P.vr2 + col.w * P.vr3;
                    const float v_alpha = P.T * S1 + ra * (P.C0 - P.bdot);
                    P.bdot += fac * S1;
                    const float gq = (ov <= MAX_ALPHA) ? ov * v_alpha : 0.f; // = opacity * vis * v_alpha
                    const float t1 = gq * dx, t2 = gq * dy;
                    acc[0] += t1; acc[1] += t2; acc[2] += gq;
                    acc[3] += t1 * dx; acc[4] += t1 * dy; acc[5] += t2 * dy;
                    acc[6] += fac * P.vr0; acc[7] += fac * P.vr1; acc[8] += fac * P.vr2; acc[9] += fac * P.vr3;
#else
                    const float sigma = 0.5f * (cn.x * dx * dx + cn.z * dy * dy) + cn.y * dx * dy;
                    float vis = __expf(-sigma);
                    const bool valid = (idx <= P.bin_final) && !(sigma < 0.f) && !(opac * vis < ALPHA_THR);
                    if (__ballot(valid) == 0ull) continue;
                    touched = true;
#if LAB_VARIANT == 3
                    sink += valid ? vis : 0.f;
#else
                    vis = valid ? vis : 0.f;
                    const float ov = opac * vis;
                    const float alpha = fminf(MAX_ALPHA, ov);
                    const float ra = __builtin_amdgcn_rcpf(1.0f - alpha);
                    P.T *= ra;
                    const float fac = alpha * P.T;
                    const float S1 = col.x * P.vr0 + col.y * P.vr1 + col.z * P.vr2 + col.w * P.vr3;
                    const float v_alpha = P.T * S1 + ra * (P.C0 - P.bdot);
                    P.bdot += fac * S1;
                    const float gop = (ov <= MAX_ALPHA) ? vis * v_alpha : 0.f;
                    const float t1 = gop * dx, t2 = gop * dy;
                    acc[0] += t1; acc[1] += t2; acc[2] += gop;
                    acc[3] += t1 * dx; acc[4] += t1 * dy; acc[5] += t2 * dy;
                    acc[6] += fac * P.vr0; acc[7] += fac * P.vr1; acc[8] += fac * P.vr2; acc[9] += fac * P.vr3;
#endif
#endif
                }
            }
            if (!touched) continue;
#if LAB_VARIANT == 3
            continue;
#elif LAB_VARIANT == 2
#pragma unroll
            for (int k = 0; k < NACC; ++k) sink += acc[k];
#elif LAB_VARIANT >= 5
            {
                touched_mask |= bit;
                // intra-row transposing butterfly only (11 DPP): afterwards the 10 owner lanes of EACH 16-lane row hold
                // that row's sum of slot o.slot
                const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
                float r[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const float keep = b3 ? acc[j + 5] : acc[j], send = b3 ? acc[j] : acc[j + 5];
                    r[j] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x140, 0xf, 0xf, false));
                }
                const float k0 = b2 ? r[3] : r[0], k1 = b2 ? r[4] : r[1], s0 = b2 ? r[0] : r[3], s1 = b2 ? r[1] : r[4];
                const float u0 = k0 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s0), 0x141, 0xf, 0xf, false));
                const float u1 = k1 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x141, 0xf, 0xf, false));
                const float u2 = r[2] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r[2]), 0x141, 0xf, 0xf, false));
                const float k30 = b2 ? (b1 ? u1 : u0) : (b1 ? u2 : u0);
                const float s30 = b2 ? (b1 ? u0 : u1) : (b1 ? u0 : u2);
                const float w0 = k30 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s30), 0x1B, 0xf, 0xf, false));
                const float w1 = u1 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u1), 0x1B, 0xf, 0xf, false));
                const bool two = !b2 && !b1;
                const float k4 = (two && b0) ? w1 : w0;
                const float s4 = two ? (b0 ? w0 : w1) : w0;
                float f = k4 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s4), 0xB1, 0xf, 0xf, false));
                const int slot = (b3 ? 5 : 0) + (b2 ? (b1 ? 4 : 3) : (b1 ? 2 : (b0 ? 1 : 0)));
                const bool row_owner = two || !b0;
#if LAB_VARIANT == 5
                if (row_owner) atomicAdd(&sacc[t][acc_to_rec(slot)], f);   // ds_add_f32, 40 lanes, 4 rows per address
#else
                {   // the 4 rows in registers: x = f, y = f; permlane32_swap -> (lo,lo),(hi,hi); permlane16_swap -> even/odd rows
                    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
                    u2v p = __builtin_amdgcn_permlane32_swap(__float_as_uint(f), __float_as_uint(f), false, false);
                    f = __uint_as_float(p.x) + __uint_as_float(p.y);
                    p = __builtin_amdgcn_permlane16_swap(__float_as_uint(f), __float_as_uint(f), false, false);
                    f = __uint_as_float(p.x) + __uint_as_float(p.y);
                    if (row_owner && lane < 16) sacc[t][acc_to_rec(slot)] = f;
                }
#endif
            }
#elif LAB_VARIANT == 4
            // LDS transpose: lane l writes its 10 sums to column l; then lane (k, part) = (l % 10, l / 10), l < 60,
            // adds 64/6 ~ 11 entries of row k; 6 partials per slot are combined with 3 DPP-free LDS reads by the owner.
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NACC; ++k) sred[k][lane] = acc[k];
            __syncthreads();
            {
                const int k = lane % 10, part = lane / 10; // part 0..5 (lanes 60..63 idle)
                float s = 0.f;
                if (lane < 60) {
#pragma unroll
                    for (int j = 0; j < 11; ++j) { const int c = part * 11 + j; s += (c < 64) ? sred[k][c] : 0.f; }
                }
                __syncthreads();
                if (lane < 60) sred[k][part] = s; // reuse: 6 partials per slot
                __syncthreads();
                if (lane < 10) {
                    float tot = ((sred[lane][0] + sred[lane][1]) + (sred[lane][2] + sred[lane][3])) + (sred[lane][4] + sred[lane][5]);
                    const float scale = (lane == 2 || lane >= 6) ? 1.0f : ((lane == 3 || lane == 5) ? -0.5f * opac : -opac);
                    tot *= scale;
                    if (tot != 0.f) unsafeAtomicAdd(v_rec + 12 * (int64_t)sid[t] + acc_to_rec(lane), tot);
                }
            }
#else
            const Reduce10 red = wave_reduce10(acc, lane);
#ifdef LAB_EXP2
            const float scale = red.slot >= 6 ? 1.0f : (red.slot == 2 ? __builtin_amdgcn_rcpf(opac) : ((red.slot == 3 || red.slot == 5) ? -0.5f : -1.0f));
#else
            const float scale = (red.slot == 2 || red.slot >= 6) ? 1.0f : ((red.slot == 3 || red.slot == 5) ? -0.5f * opac : -opac);
#endif
            const float total = red.value * scale;
#if LAB_VARIANT == 1 && defined(LAB_COUNT)
            sink += (red.is_owner && total != 0.f) ? 1.f : 0.f;      // how many lane-atomics would have been issued
            if (lane == 63) sink += 1024.f;                           // ... and how many (splat, tile) pairs reduced
#elif LAB_VARIANT == 1
            sink += red.is_owner ? total : 0.f;
#else
            if (red.is_owner && total != 0.f)
                unsafeAtomicAdd(v_rec + 12 * (int64_t)sid[t] + acc_to_rec(red.slot), total);
#endif
#endif
        }
#if LAB_VARIANT >= 5
        // flush: one instruction = 4 staged splats x their 12-dword gradient records (10 live dwords each), so the
        // atomics of an instruction fall into 4 records (full-wave instructions over 64 DIFFERENT records were measured
        // 2x slower than the per-splat 10-lane form: the memory side pays per cache line touched, not per lane)
        __syncthreads();
        {
            const int d = lane & 15;
            const bool live_d = d < 12 && d != 3 && d != 7;
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const int sp = 4 * j + (lane >> 4);
                if (!((touched_mask >> (4 * j)) & 0xFull)) continue; // wave-uniform: none of these 4 splats was touched
                if (live_d && ((touched_mask >> sp) & 1ull)) {
#if LAB_VARIANT >= 7
                    const float opac = srec[sp][0].w; // sums carry opacity already: v_opacity = sum / opacity
                    const float scale = d >= 8 ? 1.0f : (d == 2 ? __builtin_amdgcn_rcpf(opac) : ((d == 4 || d == 6) ? -0.5f : -1.0f));
#else
                    const float opac = srec[sp][0].z;
                    const float scale = (d == 2 || d >= 8) ? 1.0f : ((d == 4 || d == 6) ? -0.5f * opac : -opac);
#endif
                    const float tot = sacc[sp][d] * scale;
                    if (tot != 0.f) unsafeAtomicAdd(v_rec + 12 * (int64_t)sid[sp] + d, tot);
                }
            }
        }
#endif
    }
    if (LAB_VARIANT != 0 && LAB_VARIANT < 4) sink_out[(int64_t)tile * 64 + lane] = sink;
}

} // namespace adk

extern "C" int lab_variant(void) { return LAB_VARIANT; }

extern "C" int lab_raster_bwd(int width, int height, const float* rec, const int32_t* flatten_ids,
                              const int32_t* offsets, int64_t n_isects, const float* backgrounds,
                              const float* final_T, const int32_t* last_ids, const float* v_render_colors,
                              const float* v_render_alphas, float* v_rec, float* sink, hipStream_t stream)
{
    const int tile_w = (width + 15) / 16, tile_h = (height + 15) / 16;
    hipLaunchKernelGGL(adk::lab_bwd_kernel, dim3(tile_w * tile_h), dim3(64), 0, stream, tile_w, tile_h, width, height,
                       rec, flatten_ids, offsets, (int)n_isects, backgrounds, final_T, last_ids, v_render_colors,
                       v_render_alphas, v_rec, sink);
    return (int)hipGetLastError();
}
