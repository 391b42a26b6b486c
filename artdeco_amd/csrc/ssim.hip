// Fused SSIM forward / backward for gfx950 (CDNA4).
//
// Replaces Reconstruct/submodules/fused-ssim/ssim.cu (kernels :62-278 fwd,
// :286-427 bwd; host entry points :434-473, :481-517).  Same math (11-tap
// sigma=1.5 separable Gaussian, zero "same" padding, C1/C2 passed in), new
// decomposition designed for 64-wide wavefronts:
//
//   * one wavefront per workgroup; lane = image column, the wave marches down
//     a strip of RH output rows (+10 halo rows).  A single-wave workgroup makes
//     __syncthreads() a free s_barrier, so there are no block-level stalls.
//   * horizontal 11-tap pass: the input row (64+10 columns, both images) is
//     staged in a tiny LDS row buffer; each lane reads its 11 neighbours with
//     conflict-free consecutive-address ds_reads.
//   * vertical 11-tap pass: an 11-row ring of the 5 (fwd) / 3 (bwd) horizontal
//     sums lives in VGPRs (loop unrolled by 11 so every ring index is static);
//     no second LDS round trip, no 26x16x5 scratch tile as in the reference.
//   * all global loads/stores are 64-lane row-contiguous (256 B) and input rows
//     are prefetched into registers two iterations ahead of their use.
//
// HBM traffic per pixel-channel: fwd(train) 8 B in + 16 B out, bwd 24 B in +
// 4 B out (+ halo re-reads that hit L2).  Roofline: HBM.
#include "adk_common.hpp"

namespace adk {

// gaussian(11, 1.5) in fp32 -- identical to the table at ssim.cu:12-24 and to
// fused-ssim/tests/test.py:14-16 evaluated in fp32.
__constant__ float kGauss[11] = {
    0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f, 0.10936068743467331f,
    0.21300552785396576f,  0.26601171493530273f,   0.21300552785396576f,  0.10936068743467331f,
    0.036000773310661316f, 0.0075987582094967365f, 0.001028380123898387f};

// Round 5: which strip segment a workgroup serves.  The dispatcher deals consecutive workgroups round-robin over the 8 XCDs, each with its own
// L2: with the hardware's (x fastest) order the four neighbours of a segment -- which re-read its 10 halo rows / 10 halo columns -- all run on
// OTHER XCDs, and every halo is fetched from HBM again (counters: 1.57x / 1.53x the formula bytes for fwd / bwd at 1080p, exactly
// (RH + 10) / RH x 74 / 64).  With the XCD-aware remap each XCD owns a contiguous range of logical segments, walked DOWN a strip first, so a
// segment's vertical neighbours run on the same XCD at about the same time and the halo rows come out of its L2.  0 = hardware order.
#ifndef ADK_SSIM_XCD
#define ADK_SSIM_XCD 1
#endif
struct SsimSeg { int bx, by, bz; };
__device__ __forceinline__ SsimSeg ssim_segment() {
#if ADK_SSIM_XCD
    const int gx = (int)gridDim.x, gy = (int)gridDim.y, total = gx * gy * (int)gridDim.z;
    const int lin = ((int)blockIdx.z * gy + (int)blockIdx.y) * gx + (int)blockIdx.x;
    const int v = xcd_remap(lin, total);
    return {(v / gy) % gx, v % gy, v / (gy * gx)};
#else
    return {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
#endif
}

#define SSIM_HALO 5
#define SSIM_ROWBUF 80 // 64 + 10 halo, padded

// SUM: the strip's share of sum(ssim_map) goes to block_sums[linear block index] (the mapper's loss needs only the
// mean, h3dgsv3.py:441: a second pass over the map and, with ssim_map == NULL, the map itself are saved).
template <int RH, bool TRAIN, bool SUM>
__global__ __launch_bounds__(64) void ssim_fwd_kernel(
    int H, int W, float C1, float C2,
    const float* __restrict__ img1, const float* __restrict__ img2,
    float* __restrict__ ssim_map, float* __restrict__ dm_dmu1,
    float* __restrict__ dm_dsigma1_sq, float* __restrict__ dm_dsigma12, float* __restrict__ block_sums)
{
    __shared__ float rowbuf[2][2][SSIM_ROWBUF]; // [parity][image][column]

    const int lane = threadIdx.x;
    const SsimSeg seg = ssim_segment();
    const int x0 = seg.bx * 64;
    const int y0 = seg.by * RH;
    const int64_t plane = (int64_t)seg.bz * H * W;
    const float* p1 = img1 + plane;
    const float* p2 = img2 + plane;

    float g[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) g[i] = kGauss[i];

    // column this lane loads for the row buffer (main slot + 10 extra slots)
    const int xa = x0 - SSIM_HALO + lane;
    const int xb = x0 - SSIM_HALO + 64 + lane; // lanes 0..9 only
    const bool xa_ok = (xa >= 0) && (xa < W);
    const bool xb_ok = (lane < 2 * SSIM_HALO) && (xb < W);
    const int x = x0 + lane; // output column

    float win[11][5];
    float strip_sum = 0.f;

    constexpr int NROWS = RH + 2 * SSIM_HALO;

    // Rows are fetched TWO iterations ahead of their use: one wavefront keeps only a handful of 256 B loads in flight,
    // and with a one-row lead every row waited most of an HBM round trip (the kernel ran at 1.5 TB/s).
    float a1, a2, b1, b2;     // row i   (written to the row buffer this iteration)
    float n1, n2, m1, m2;     // row i+1 (in flight or landed)
    auto fetch_row = [&](int i, float& r1, float& r2, float& e1, float& e2) {
        const int r = y0 - SSIM_HALO + i;
        const bool rok = (r >= 0) && (r < H) && (i < NROWS);
        const int64_t off = (int64_t)r * W;
        r1 = (rok && xa_ok) ? p1[off + xa] : 0.f;
        r2 = (rok && xa_ok) ? p2[off + xa] : 0.f;
        e1 = (rok && xb_ok) ? p1[off + xb] : 0.f;
        e2 = (rok && xb_ok) ? p2[off + xb] : 0.f;
    };
    fetch_row(0, a1, a2, b1, b2);
    fetch_row(1, n1, n2, m1, m2);

    for (int base = 0; base < NROWS; base += 11) {
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const int i = base + k;
            if (i < NROWS) {
                const int par = k & 1; // two barriers per row make any slot choice safe; parity just
                                       // lets the next row's ds_writes start without aliasing.
                rowbuf[par][0][lane] = a1;
                rowbuf[par][1][lane] = a2;
                if (lane < 2 * SSIM_HALO) {
                    rowbuf[par][0][64 + lane] = b1;
                    rowbuf[par][1][64 + lane] = b2;
                }
                // rotate the prefetch queue and fetch row i+2 while this one is being reduced
                a1 = n1; a2 = n2; b1 = m1; b2 = m2;
                fetch_row(i + 2, n1, n2, m1, m2);
                __syncthreads();

                // horizontal 11-tap pass (pairs around the centre, as ssim.cu:134-158)
                float sX = 0.f, sX2 = 0.f, sY = 0.f, sY2 = 0.f, sXY = 0.f;
                const float* rb1 = &rowbuf[par][0][lane];
                const float* rb2 = &rowbuf[par][1][lane];
#pragma unroll
                for (int d = 1; d <= SSIM_HALO; ++d) {
                    const float w = g[SSIM_HALO - d];
                    const float Xl = rb1[SSIM_HALO - d], Xr = rb1[SSIM_HALO + d];
                    const float Yl = rb2[SSIM_HALO - d], Yr = rb2[SSIM_HALO + d];
                    sX += (Xl + Xr) * w;
                    sX2 += (Xl * Xl + Xr * Xr) * w;
                    sY += (Yl + Yr) * w;
                    sY2 += (Yl * Yl + Yr * Yr) * w;
                    sXY += (Xl * Yl + Xr * Yr) * w;
                }
                {
                    const float Xc = rb1[SSIM_HALO], Yc = rb2[SSIM_HALO], wc = g[SSIM_HALO];
                    sX += Xc * wc;
                    sX2 += Xc * Xc * wc;
                    sY += Yc * wc;
                    sY2 += Yc * Yc * wc;
                    sXY += Xc * Yc * wc;
                }
                win[k][0] = sX; win[k][1] = sX2; win[k][2] = sY; win[k][3] = sY2; win[k][4] = sXY;
                __syncthreads(); // row buffer may be overwritten two rows later; keeps parity reuse safe

                if (i >= 2 * SSIM_HALO) {
                    const int yo = y0 + i - 2 * SSIM_HALO;
                    // vertical 11-tap pass from the register ring; tap j sits in slot (k+1+j)%11
                    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f, o4 = 0.f;
#pragma unroll
                    for (int d = 1; d <= SSIM_HALO; ++d) {
                        const float w = g[SSIM_HALO - d];
                        const int st = (k + 1 + SSIM_HALO - d) % 11, sb = (k + 1 + SSIM_HALO + d) % 11;
                        o0 += (win[st][0] + win[sb][0]) * w;
                        o1 += (win[st][1] + win[sb][1]) * w;
                        o2 += (win[st][2] + win[sb][2]) * w;
                        o3 += (win[st][3] + win[sb][3]) * w;
                        o4 += (win[st][4] + win[sb][4]) * w;
                    }
                    {
                        const int sc = (k + 1 + SSIM_HALO) % 11;
                        const float wc = g[SSIM_HALO];
                        o0 += win[sc][0] * wc; o1 += win[sc][1] * wc; o2 += win[sc][2] * wc;
                        o3 += win[sc][3] * wc; o4 += win[sc][4] * wc;
                    }
                    if (yo < H && x < W) {
                        const float mu1 = o0, mu2 = o2;
                        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2;
                        const float sigma1_sq = o1 - mu1_sq;
                        const float sigma2_sq = o3 - mu2_sq;
                        const float sigma12 = o4 - mu1 * mu2;
                        const float A = mu1_sq + mu2_sq + C1;
                        const float Bv = sigma1_sq + sigma2_sq + C2;
                        const float C_ = 2.f * mu1 * mu2 + C1;
                        const float D_ = 2.f * sigma12 + C2;
                        const float inv_AB = 1.f / (A * Bv);
                        const int64_t o = plane + (int64_t)yo * W + x;
                        const float ssim = (C_ * D_) * inv_AB;
                        if (!SUM || ssim_map) ssim_map[o] = ssim; // SUM: uniform pointer test
                        if (SUM) strip_sum += ssim;
                        if (TRAIN) {
                            // d(ssim)/d(mu1), d/d(sigma1^2), d/d(sigma12): ssim.cu:260-274
                            const float inv_A = 1.f / A, inv_B = 1.f / Bv;
                            const float d_mu1 = (mu2 * 2.f * D_) * inv_AB - (mu2 * 2.f * C_) * inv_AB
                                              - (mu1 * 2.f * C_ * D_) * inv_AB * inv_A
                                              + (mu1 * 2.f * C_ * D_) * inv_AB * inv_B;
                            dm_dmu1[o] = d_mu1;
                            dm_dsigma1_sq[o] = (-C_ * D_) * inv_AB * inv_B;
                            dm_dsigma12[o] = (2.f * C_) * inv_AB;
                        }
                    }
                }
            }
        }
    }
    if (SUM) { // every lane is back here: fixed DPP order => the same bits on every run
        const float t = wave_sum_to_lane63(strip_sum);
        if (lane == 63) block_sums[((int64_t)seg.bz * gridDim.y + seg.by) * gridDim.x + seg.bx] = t;
    }
}

template <int RH>
__global__ __launch_bounds__(64) void ssim_bwd_kernel(
    int H, int W,
    const float* __restrict__ img1, const float* __restrict__ img2,
    const float* __restrict__ dL_dmap,  // per-pixel map, or nullptr => uniform dL_scalar
    float dL_scalar,
    const float* __restrict__ dm_dmu1, const float* __restrict__ dm_dsigma1_sq,
    const float* __restrict__ dm_dsigma12, float* __restrict__ dL_dimg1)
{
    __shared__ float rowbuf[2][3][SSIM_ROWBUF];

    const int lane = threadIdx.x;
    const SsimSeg seg = ssim_segment();
    const int x0 = seg.bx * 64;
    const int y0 = seg.by * RH;
    const int64_t plane = (int64_t)seg.bz * H * W;

    float g[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) g[i] = kGauss[i];

    const int xa = x0 - SSIM_HALO + lane;
    const int xb = x0 - SSIM_HALO + 64 + lane;
    const bool xa_ok = (xa >= 0) && (xa < W);
    const bool xb_ok = (lane < 2 * SSIM_HALO) && (xb < W);
    const int x = x0 + lane;

    float win[11][3];
    constexpr int NROWS = RH + 2 * SSIM_HALO;

    // fused (dm_d* x dL_dmap) products for one row, main + extra slot; fetched two rows ahead of their use
    float fa[3], fb[3], na[3], nb[3];
    auto fetch = [&](int i, float* ra, float* rb) {
        const int r = y0 - SSIM_HALO + i;
        const bool rok = (r >= 0) && (r < H) && (i < NROWS);
        const int64_t off = plane + (int64_t)r * W;
        if (rok && xa_ok) {
            const float c = dL_dmap ? dL_dmap[off + xa] : dL_scalar;
            ra[0] = dm_dmu1[off + xa] * c; ra[1] = dm_dsigma1_sq[off + xa] * c; ra[2] = dm_dsigma12[off + xa] * c;
        } else { ra[0] = ra[1] = ra[2] = 0.f; }
        if (rok && xb_ok) {
            const float c = dL_dmap ? dL_dmap[off + xb] : dL_scalar;
            rb[0] = dm_dmu1[off + xb] * c; rb[1] = dm_dsigma1_sq[off + xb] * c; rb[2] = dm_dsigma12[off + xb] * c;
        } else { rb[0] = rb[1] = rb[2] = 0.f; }
    };
    fetch(0, fa, fb);
    fetch(1, na, nb);

    for (int base = 0; base < NROWS; base += 11) {
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const int i = base + k;
            if (i < NROWS) {
                const int par = k & 1;
#pragma unroll
                for (int q = 0; q < 3; ++q) rowbuf[par][q][lane] = fa[q];
                if (lane < 2 * SSIM_HALO) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) rowbuf[par][q][64 + lane] = fb[q];
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) { fa[q] = na[q]; fb[q] = nb[q]; }
                fetch(i + 2, na, nb);
                __syncthreads();

                float s0 = 0.f, s1 = 0.f, s2 = 0.f;
                const float* r0 = &rowbuf[par][0][lane];
                const float* r1 = &rowbuf[par][1][lane];
                const float* r2 = &rowbuf[par][2][lane];
#pragma unroll
                for (int d = 1; d <= SSIM_HALO; ++d) {
                    const float w = g[SSIM_HALO - d];
                    s0 += (r0[SSIM_HALO - d] + r0[SSIM_HALO + d]) * w;
                    s1 += (r1[SSIM_HALO - d] + r1[SSIM_HALO + d]) * w;
                    s2 += (r2[SSIM_HALO - d] + r2[SSIM_HALO + d]) * w;
                }
                s0 += r0[SSIM_HALO] * g[SSIM_HALO];
                s1 += r1[SSIM_HALO] * g[SSIM_HALO];
                s2 += r2[SSIM_HALO] * g[SSIM_HALO];
                win[k][0] = s0; win[k][1] = s1; win[k][2] = s2;
                __syncthreads();

                if (i >= 2 * SSIM_HALO) {
                    const int yo = y0 + i - 2 * SSIM_HALO;
                    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
                    for (int d = 1; d <= SSIM_HALO; ++d) {
                        const float w = g[SSIM_HALO - d];
                        const int st = (k + 1 + SSIM_HALO - d) % 11, sb = (k + 1 + SSIM_HALO + d) % 11;
                        t0 += (win[st][0] + win[sb][0]) * w;
                        t1 += (win[st][1] + win[sb][1]) * w;
                        t2 += (win[st][2] + win[sb][2]) * w;
                    }
                    {
                        const int sc = (k + 1 + SSIM_HALO) % 11;
                        t0 += win[sc][0] * g[SSIM_HALO];
                        t1 += win[sc][1] * g[SSIM_HALO];
                        t2 += win[sc][2] * g[SSIM_HALO];
                    }
                    if (yo < H && x < W) {
                        const int64_t o = plane + (int64_t)yo * W + x;
                        const float px1 = img1[o], px2 = img2[o];
                        // ssim.cu:420
                        dL_dimg1[o] = t0 + (2.f * px1) * t1 + px2 * t2;
                    }
                }
            }
        }
    }
}

} // namespace adk

// strip height: 32 rows when that still yields >= ~4 waves per SIMD, else 16
static int ssim_strip_rows(int B, int CH, int H, int W)
{
    const int64_t waves32 = adk::ceil_div(W, 64) * adk::ceil_div(H, 32) * B * CH;
    return waves32 >= 4096 ? 32 : 16;
}

static int ssim_fwd_launch(const float* img1, const float* img2, int B, int CH, int H, int W, float C1, float C2, float* ssim_map,
                           float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, float* block_sums, hipStream_t stream)
{
    const bool train = dm_dmu1 != nullptr;
    if (train && (!dm_dsigma1_sq || !dm_dsigma12)) return ADK_EINVAL;
    if ((int64_t)B * CH > 65535) return ADK_EUNSUPPORTED;
    const int rh = ssim_strip_rows(B, CH, H, W);
    const dim3 block(64), grid((unsigned)adk::ceil_div(W, 64), (unsigned)adk::ceil_div(H, rh), (unsigned)(B * CH));
#define SSIM_FWD(RH, TRAIN, SUM) hipLaunchKernelGGL((adk::ssim_fwd_kernel<RH, TRAIN, SUM>), grid, block, 0, stream, H, W, C1, C2, img1, img2, \
                                                    ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, block_sums)
    if (block_sums) { if (rh == 32) SSIM_FWD(32, true, true); else SSIM_FWD(16, true, true); }
    else if (train) { if (rh == 32) SSIM_FWD(32, true, false); else SSIM_FWD(16, true, false); }
    else { if (rh == 32) SSIM_FWD(32, false, false); else SSIM_FWD(16, false, false); }
#undef SSIM_FWD
    ADK_RETURN_LAST_ERROR();
}

extern "C" int adk_fused_ssim_fwd(const float* img1, const float* img2, int B, int CH, int H, int W,
                                  float C1, float C2, float* ssim_map, float* dm_dmu1,
                                  float* dm_dsigma1_sq, float* dm_dsigma12, hipStream_t stream)
{
    if (B < 0 || CH < 0 || H < 0 || W < 0) return ADK_EINVAL;
    if ((int64_t)B * CH * H * W == 0) return 0;
    if (!img1 || !img2 || !ssim_map) return ADK_EINVAL;
    return ssim_fwd_launch(img1, img2, B, CH, H, W, C1, C2, ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, nullptr, stream);
}

extern "C" int64_t adk_fused_ssim_fwd_sums_count(int B, int CH, int H, int W)
{
    if (B < 0 || CH < 0 || H < 0 || W < 0) return -1;
    if ((int64_t)B * CH * H * W == 0) return 0;
    return adk::ceil_div(W, 64) * adk::ceil_div(H, ssim_strip_rows(B, CH, H, W)) * (int64_t)B * CH;
}

// Training forward that also (or only: ssim_map may be NULL) leaves sum(ssim_map) as adk_fused_ssim_fwd_sums_count(...)
// partial sums, one per strip, in a fixed order.
extern "C" int adk_fused_ssim_fwd_sums(const float* img1, const float* img2, int B, int CH, int H, int W, float C1, float C2,
                                       float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                                       float* block_sums, hipStream_t stream)
{
    if (B < 0 || CH < 0 || H < 0 || W < 0) return ADK_EINVAL;
    if ((int64_t)B * CH * H * W == 0) return 0;
    if (!img1 || !img2 || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !block_sums) return ADK_EINVAL;
    return ssim_fwd_launch(img1, img2, B, CH, H, W, C1, C2, ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, block_sums, stream);
}

extern "C" int adk_fused_ssim_bwd(const float* img1, const float* img2, const float* dL_dmap, float dL_scalar,
                                  const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                                  int B, int CH, int H, int W, float* dL_dimg1, hipStream_t stream)
{
    if (B < 0 || CH < 0 || H < 0 || W < 0) return ADK_EINVAL;
    if ((int64_t)B * CH * H * W == 0) return 0;
    if (!img1 || !img2 || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1) return ADK_EINVAL;
    if ((int64_t)B * CH > 65535) return ADK_EUNSUPPORTED;
    const int64_t waves32 = adk::ceil_div(W, 64) * adk::ceil_div(H, 32) * B * CH;
    const dim3 block(64);
    if (waves32 >= 2048) {
        const dim3 grid((unsigned)adk::ceil_div(W, 64), (unsigned)adk::ceil_div(H, 32), (unsigned)(B * CH));
        hipLaunchKernelGGL((adk::ssim_bwd_kernel<32>), grid, block, 0, stream, H, W, img1, img2, dL_dmap, dL_scalar, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
    } else {
        const dim3 grid((unsigned)adk::ceil_div(W, 64), (unsigned)adk::ceil_div(H, 16), (unsigned)(B * CH));
        hipLaunchKernelGGL((adk::ssim_bwd_kernel<16>), grid, block, 0, stream, H, W, img1, img2, dL_dmap, dL_scalar, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
    }
    ADK_RETURN_LAST_ERROR();
}
