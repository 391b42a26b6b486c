// Per-tile alpha compositing, forward and backward, for gfx950 (CDNA4).
//
// Replaces gsplat's rasterize_to_pixels fwd/bwd [UPSTREAM gsplat >= 1.5, not vendored; semantics
// per SURVEY.md App. A items 4-5], reached from gsplat.rendering.rasterization at
// Reconstruct/scene/scene_models/h3dgsv3.py:664-680.  Per-pixel results follow the upstream
// control flow exactly (skip sigma<0 or alpha<1/255, terminate BEFORE adding when
// T(1-alpha) <= 1e-4, last contributing index, back-to-front gradient recurrences).
//
// Wave64 design (not a port of the 32-lane warp tiling):
//   * a workgroup = one 16x16 tile = 4 wavefronts; each wavefront owns an 8x8 pixel QUADRANT
//     (lane -> (lane&7, lane>>3)), the squarest footprint 64 lanes can have, so a splat that
//     misses a quadrant is skipped by that whole wave;
//   * splats are staged 256 at a time in LDS as packed 48 B records (three ds_read_b128 per
//     splat per wave, wave-uniform address => broadcast reads, no bank conflicts);
//   * SPLAT-PARALLEL CULLING: before walking a group of 64 staged splats, each LANE tests ONE
//     splat's screen-space box against the wave's quadrant; __ballot gives a 64-bit mask in
//     SGPRs and the wave then walks only the set bits (s_ff1 / s_flbit), in order.  64 cull
//     tests cost what one used to, and culled splats cost nothing in the compositing loop.
//     The test is conservative w.r.t. the alpha>=1/255 rule (the radii bound exactly that).
//   * backward: per-splat gradients are reduced across the 64 lanes, accumulated per staged
//     splat in LDS across the 4 quadrant waves, and flushed once per (splat, tile) with
//     hardware fp32 atomics into the packed 48 B gradient record.
//   * tiles are mapped to workgroups through an XCD-aware bijection so that neighbouring tiles
//     (which share Gaussians) run on the same XCD and hit the same 4 MiB L2.
#include "adk_common.hpp"

namespace adk {

#define TILE 16
#define BATCH 256
#define MAX_ALPHA 0.999f
#define ALPHA_THR (1.0f / 255.0f)
#define T_EPS 1e-4f

struct TileCtx {
    int tile, tx, ty;
    int px, py;      // this lane's pixel
    float qx0, qx1, qy0, qy1; // pixel-centre extent of this wave's quadrant
    bool inside;
};

__device__ __forceinline__ TileCtx make_ctx(int tile_w, int n_tiles, int W, int H) {
    TileCtx c;
    c.tile = xcd_remap(blockIdx.x, n_tiles);
    c.tx = c.tile % tile_w; c.ty = c.tile / tile_w;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int qx = (wv & 1) * 8, qy = (wv >> 1) * 8;
    c.px = c.tx * TILE + qx + (lane & 7);
    c.py = c.ty * TILE + qy + (lane >> 3);
    c.qx0 = (float)(c.tx * TILE + qx) + 0.5f; c.qx1 = c.qx0 + 7.0f;
    c.qy0 = (float)(c.ty * TILE + qy) + 0.5f; c.qy1 = c.qy0 + 7.0f;
    c.inside = (c.px < W) && (c.py < H);
    return c;
}

// ---------------------------------------------------------------------------------- forward
template <bool MAIN_ID>
__global__ __launch_bounds__(256) void raster_fwd_kernel(
    int tile_w, int tile_h, int W, int H, const float* __restrict__ rec, const int32_t* __restrict__ flatten_ids,
    const int32_t* __restrict__ offsets, int n_isects, const float* __restrict__ backgrounds,
    float* __restrict__ render_colors, float* __restrict__ render_alphas, int32_t* __restrict__ last_ids,
    int32_t* __restrict__ main_ids)
{
    __shared__ float4 srec[BATCH][3];
    const int n_tiles = tile_w * tile_h;
    const TileCtx c = make_ctx(tile_w, n_tiles, W, H);
    const int lane = threadIdx.x & 63;
    const float fx = (float)c.px + 0.5f, fy = (float)c.py + 0.5f;

    const int range_start = offsets[c.tile];
    const int range_end = (c.tile == n_tiles - 1) ? n_isects : offsets[c.tile + 1];
    const int num_batches = (range_end - range_start + BATCH - 1) / BATCH;

    bool done = !c.inside;
    float T = 1.0f;
    int cur_idx = 0;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
    float best_vis = 0.f;   // MAIN_ID: largest alpha*T seen and the list index that produced it
    int best_idx = -1;
    const float4* rec4 = reinterpret_cast<const float4*>(rec);

    for (int b = 0; b < num_batches; ++b) {
        if (__syncthreads_and(done)) break;
        const int batch_start = range_start + BATCH * b;
        const int idx = batch_start + (int)threadIdx.x;
        if (idx < range_end) {
            const int64_t g = flatten_ids[idx];
            srec[threadIdx.x][0] = rec4[3 * g];
            srec[threadIdx.x][1] = rec4[3 * g + 1];
            srec[threadIdx.x][2] = rec4[3 * g + 2];
        }
        __syncthreads();
        const int batch_size = min(BATCH, range_end - batch_start);
        if (__ballot(!done) == 0ull) continue; // this wave's quadrant is finished; keep serving loads
        for (int sub = 0; sub < batch_size; sub += 64) {
            const int s = sub + lane;
            bool hit = false;
            if (s < batch_size) {
                const float4 a = srec[s][0];
                const float ry = srec[s][1].w;
                hit = (a.x + a.w >= c.qx0) && (a.x - a.w <= c.qx1) && (a.y + ry >= c.qy0) && (a.y - ry <= c.qy1);
            }
            unsigned long long mask = __ballot(hit);
            while (mask) {
                const int t = sub + __builtin_ctzll(mask);
                mask &= mask - 1;
                if (!done) {
                    const float4 a = srec[t][0];
                    const float4 cn = srec[t][1];
                    const float dx = a.x - fx, dy = a.y - fy;
                    const float sigma = 0.5f * (cn.x * dx * dx + cn.z * dy * dy) + cn.y * dx * dy;
                    const float alpha = fminf(MAX_ALPHA, a.z * __expf(-sigma));
                    if (!(sigma < 0.f || alpha < ALPHA_THR)) {
                        const float next_T = T * (1.0f - alpha);
                        if (next_T <= T_EPS) {
                            done = true;
                        } else {
                            const float4 col = srec[t][2];
                            const float vis = alpha * T;
                            o0 += col.x * vis; o1 += col.y * vis; o2 += col.z * vis; o3 += col.w * vis;
                            cur_idx = batch_start + t;
                            if (MAIN_ID && vis > best_vis) { best_vis = vis; best_idx = cur_idx; }
                            T = next_T;
                        }
                    }
                }
                if (__ballot(!done) == 0ull) { mask = 0; sub = batch_size; }
            }
        }
    }

    if (c.inside) {
        const int64_t pix = (int64_t)c.py * W + c.px;
        render_alphas[pix] = 1.0f - T;
        if (backgrounds) {
            o0 += T * backgrounds[0]; o1 += T * backgrounds[1]; o2 += T * backgrounds[2]; o3 += T * backgrounds[3];
        }
        reinterpret_cast<float4*>(render_colors)[pix] = make_float4(o0, o1, o2, o3);
        last_ids[pix] = cur_idx;
        if (MAIN_ID) main_ids[pix] = best_idx >= 0 ? flatten_ids[best_idx] : -1;
    }
}

// ---------------------------------------------------------------------------------- backward
// Accumulator slots per staged splat: 0,1 v_mean2d | 2 v_opacity | 3,4,5 v_conic | 6..9 v_colour
#define NACC 10
__device__ __forceinline__ constexpr int acc_to_rec(int k) { return k < 3 ? k : (k < 6 ? k + 1 : k + 2); }

__global__ __launch_bounds__(256) void raster_bwd_kernel(
    int tile_w, int tile_h, int W, int H, const float* __restrict__ rec, const int32_t* __restrict__ flatten_ids,
    const int32_t* __restrict__ offsets, int n_isects, const float* __restrict__ backgrounds,
    const float* __restrict__ render_alphas, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_render_colors, const float* __restrict__ v_render_alphas,
    float* __restrict__ v_rec)
{
    __shared__ float4 srec[BATCH][3];
    __shared__ int sid[BATCH];
    __shared__ float sacc[BATCH][NACC + 1]; // +1 pad: stride 11 dwords keeps the flush conflict-free
    const int n_tiles = tile_w * tile_h;
    const TileCtx c = make_ctx(tile_w, n_tiles, W, H);
    const int lane = threadIdx.x & 63;
    const float fx = (float)c.px + 0.5f, fy = (float)c.py + 0.5f;

    const int range_start = offsets[c.tile];
    const int range_end = (c.tile == n_tiles - 1) ? n_isects : offsets[c.tile + 1];
    const int num_batches = (range_end - range_start + BATCH - 1) / BATCH;

    const int64_t pix = (int64_t)c.py * W + c.px;
    float T_final = 1.f, vr0 = 0.f, vr1 = 0.f, vr2 = 0.f, vr3 = 0.f, v_render_a = 0.f;
    int bin_final = -1;
    if (c.inside) {
        T_final = 1.0f - render_alphas[pix];
        const float4 v = reinterpret_cast<const float4*>(v_render_colors)[pix];
        vr0 = v.x; vr1 = v.y; vr2 = v.z; vr3 = v.w;
        v_render_a = v_render_alphas[pix];
        bin_final = last_ids[pix];
    }
    float T = T_final;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f; // colour accumulated BEHIND the current splat
    float bg_dot = 0.f;
    if (backgrounds) bg_dot = backgrounds[0] * vr0 + backgrounds[1] * vr1 + backgrounds[2] * vr2 + backgrounds[3] * vr3;
    const int wave_bin_final = wave_max_i(bin_final);
    const float4* rec4 = reinterpret_cast<const float4*>(rec);

#pragma unroll
    for (int k = 0; k < NACC; ++k) sacc[threadIdx.x][k] = 0.f;

    for (int b = 0; b < num_batches; ++b) {
        __syncthreads();
        const int batch_end = range_end - 1 - BATCH * b;
        const int batch_size = min(BATCH, batch_end + 1 - range_start);
        const int idx = batch_end - (int)threadIdx.x;
        if (idx >= range_start) {
            const int g = flatten_ids[idx];
            sid[threadIdx.x] = g;
            srec[threadIdx.x][0] = rec4[3 * (int64_t)g];
            srec[threadIdx.x][1] = rec4[3 * (int64_t)g + 1];
            srec[threadIdx.x][2] = rec4[3 * (int64_t)g + 2];
        }
        __syncthreads();
        // staged slot s holds global list index batch_end - s (s = 0 is the furthest back)
        for (int sub = 0; sub < batch_size; sub += 64) {
            const int s = sub + lane;
            bool hit = false;
            if (s < batch_size && (batch_end - s) <= wave_bin_final) {
                const float4 a = srec[s][0];
                const float ry = srec[s][1].w;
                hit = (a.x + a.w >= c.qx0) && (a.x - a.w <= c.qx1) && (a.y + ry >= c.qy0) && (a.y - ry <= c.qy1);
            }
            unsigned long long mask = __ballot(hit);
            while (mask) {
                const int t = sub + __builtin_ctzll(mask);
                mask &= mask - 1;
                bool valid = c.inside && (batch_end - t <= bin_final);
                float alpha = 0.f, opac = 0.f, vis = 0.f, dx = 0.f, dy = 0.f;
                float4 cn = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid) {
                    const float4 a = srec[t][0];
                    cn = srec[t][1];
                    opac = a.z;
                    dx = a.x - fx; dy = a.y - fy;
                    const float sigma = 0.5f * (cn.x * dx * dx + cn.z * dy * dy) + cn.y * dx * dy;
                    vis = __expf(-sigma);
                    alpha = fminf(MAX_ALPHA, opac * vis);
                    if (sigma < 0.f || alpha < ALPHA_THR) valid = false;
                }
                if (__ballot(valid) == 0ull) continue;
                float acc[NACC];
#pragma unroll
                for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
                if (valid) {
                    const float4 col = srec[t][2];
                    const float ra = 1.0f / (1.0f - alpha);
                    T *= ra;
                    const float fac = alpha * T;
                    acc[6] = fac * vr0; acc[7] = fac * vr1; acc[8] = fac * vr2; acc[9] = fac * vr3;
                    float v_alpha = (col.x * T - b0 * ra) * vr0 + (col.y * T - b1 * ra) * vr1 +
                                    (col.z * T - b2 * ra) * vr2 + (col.w * T - b3 * ra) * vr3;
                    v_alpha += T_final * ra * v_render_a;
                    if (backgrounds) v_alpha += -T_final * ra * bg_dot;
                    if (opac * vis <= MAX_ALPHA) {
                        const float v_sigma = -opac * vis * v_alpha;
                        acc[3] = 0.5f * v_sigma * dx * dx;
                        acc[4] = v_sigma * dx * dy;
                        acc[5] = 0.5f * v_sigma * dy * dy;
                        acc[0] = v_sigma * (cn.x * dx + cn.y * dy);
                        acc[1] = v_sigma * (cn.y * dx + cn.z * dy);
                        acc[2] = vis * v_alpha;
                    }
                    b0 += col.x * fac; b1 += col.y * fac; b2 += col.z * fac; b3 += col.w * fac;
                }
// 64 -> 4 with DPP row reductions (VALU only), then ONE ds_add_f32 in which lane
                // (row*16 + k) adds row `row`'s partial of value k: the 4 rows meet in the LDS atomic.
#ifndef ADK_ABLATE_NO_REDUCE
#pragma unroll
                for (int k = 0; k < NACC; ++k) acc[k] = row16_allreduce_sum(acc[k]);
#endif
#ifndef ADK_ABLATE_NO_LDSADD
                {
                    const int kk = lane & 15;
                    float v = acc[0];
#pragma unroll
                    for (int k = 1; k < NACC; ++k) v = (kk == k) ? acc[k] : v;
                    if (kk < NACC && v != 0.f) unsafeAtomicAdd(&sacc[t][kk], v);
                }
#else
                { float v = 0.f;
#pragma unroll
                  for (int k = 0; k < NACC; ++k) v += acc[k];
                  if (v == 123.456f) sacc[t][0] = v; }
#endif
            }
        }
        __syncthreads();
        // flush: one thread per staged splat, hardware fp32 atomics into the packed gradient record
        if ((int)threadIdx.x < batch_size) {
            const int64_t g = sid[threadIdx.x];
            float* dst = v_rec + 12 * g;
#pragma unroll
            for (int k = 0; k < NACC; ++k) {
                const float v = sacc[threadIdx.x][k];
#ifndef ADK_ABLATE_NO_FLUSH
                if (v != 0.f) { unsafeAtomicAdd(dst + acc_to_rec(k), v); sacc[threadIdx.x][k] = 0.f; }
#else
                if (v == 123.456f) dst[acc_to_rec(k)] = v;
                sacc[threadIdx.x][k] = 0.f;
#endif
            }
        }
    }
}

} // namespace adk

// render_colors [H,W,4], render_alphas [H,W], last_ids [H,W]; backgrounds [4] or NULL; main_ids [H,W]
// (Gaussian id with the largest alpha*T per pixel, -1 if none) or NULL.
extern "C" int adk_raster_fwd(int width, int height, const float* rec, const int32_t* flatten_ids,
                              const int32_t* offsets, int64_t n_isects, const float* backgrounds,
                              float* render_colors, float* render_alphas, int32_t* last_ids, int32_t* main_ids,
                              hipStream_t stream)
{
    if (width <= 0 || height <= 0 || n_isects < 0 || n_isects >= ((int64_t)1 << 31)) return ADK_EINVAL;
    if (!offsets || !render_colors || !render_alphas || !last_ids) return ADK_EINVAL;
    if (n_isects > 0 && (!rec || !flatten_ids)) return ADK_EINVAL;
    if (((uintptr_t)rec & 15) || ((uintptr_t)render_colors & 15)) return ADK_EINVAL;
    const int tile_w = (width + 15) / 16, tile_h = (height + 15) / 16;
    if (main_ids)
        hipLaunchKernelGGL(adk::raster_fwd_kernel<true>, dim3(tile_w * tile_h), dim3(256), 0, stream, tile_w, tile_h, width, height,
                           rec, flatten_ids, offsets, (int)n_isects, backgrounds, render_colors, render_alphas, last_ids, main_ids);
    else
        hipLaunchKernelGGL(adk::raster_fwd_kernel<false>, dim3(tile_w * tile_h), dim3(256), 0, stream, tile_w, tile_h, width, height,
                           rec, flatten_ids, offsets, (int)n_isects, backgrounds, render_colors, render_alphas, last_ids, nullptr);
    ADK_RETURN_LAST_ERROR();
}

// v_rec [N,12] must be zero-initialised by the caller; gradients are accumulated into it.
extern "C" int adk_raster_bwd(int width, int height, const float* rec, const int32_t* flatten_ids,
                              const int32_t* offsets, int64_t n_isects, const float* backgrounds,
                              const float* render_alphas, const int32_t* last_ids, const float* v_render_colors,
                              const float* v_render_alphas, float* v_rec, hipStream_t stream)
{
    if (width <= 0 || height <= 0 || n_isects < 0 || n_isects >= ((int64_t)1 << 31)) return ADK_EINVAL;
    if (n_isects == 0) return 0;
    if (!rec || !flatten_ids || !offsets || !render_alphas || !last_ids || !v_render_colors || !v_render_alphas || !v_rec) return ADK_EINVAL;
    if (((uintptr_t)rec & 15) || ((uintptr_t)v_render_colors & 15)) return ADK_EINVAL;
    const int tile_w = (width + 15) / 16, tile_h = (height + 15) / 16;
    hipLaunchKernelGGL(adk::raster_bwd_kernel, dim3(tile_w * tile_h), dim3(256), 0, stream, tile_w, tile_h, width, height,
                       rec, flatten_ids, offsets, (int)n_isects, backgrounds, render_alphas, last_ids, v_render_colors,
                       v_render_alphas, v_rec);
    ADK_RETURN_LAST_ERROR();
}
