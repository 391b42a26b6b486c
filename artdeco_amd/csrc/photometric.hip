// Image-space half of the mapper's training step, fused, for gfx950.
//
// Replaces the ~90 small torch kernels that SceneModel.render / render_from_id / optimization_step launch per
// step between the rasteriser output and loss.backward() (Reconstruct/scene/scene_models/h3dgsv3.py):
//     :690-694  render = colors[...,:3] + (1 - alpha) * bg ; invdepth = 1 / colors[...,3]
//     :611-614  render = clamp(exposure[:3,:3] @ render + exposure[:3,3], 0, 1)
//     :432-439  (not is_important) outlier mask from rdk * |render - gt| > 0.2, applied to render, gt, invdepth, mono
//     :440-448  l1 = mean(rdk |render - gt|) ; depth = mean(rdk |invdepth - mono|) ;
//               loss = lambda (1 - ssim) + (1 - lambda) l1 + w_depth depth
// and their autograd backward.  Three streaming kernels:
//   photometric_fwd   36 B/px in, 16..28 B/px out; per-workgroup partial sums of the two L1 terms
//   photometric_loss  two-stage deterministic reduction of the partials and of the SSIM map -> 4 scalars
//   photometric_bwd   52 B/px in, 20 B/px out: gradient w.r.t. the rasteriser's [H,W,4] colours and [H,W] alphas
//                     (the layouts raster_bwd consumes, no permutes) + the 12 exposure gradients.
// The SSIM term itself stays in ssim.hip; its image gradient enters photometric_bwd as an input.
// Also here: the 6D-pose -> view matrix map of Keyframe.get_Rt (scene/keyframe.py:150-154, utils.py:223-229) with
// its analytic backward, and the visibility masks of render (h3dgsv3.py:695-698).
#include "adk_common.hpp"
#include "adk_internal.hpp"
#include "pose6d.hpp"

namespace adk {

#define PHOTO_THREADS 256
#define PHOTO_MAX_BLOCKS 2048
#define PHOTO_SSIM_BLOCKS 1024
#ifndef ADK_PHOTO_V4
#define ADK_PHOTO_V4 1
#endif

struct Exposure { float e[12]; };

__device__ __forceinline__ float sgnf(float x) { return (x > 0.f ? 1.f : 0.f) - (x < 0.f ? 1.f : 0.f); }

// one pixel of the forward chain; returns the outlier mask (1 = keep)
struct PixFwd { float c[3], u[3], e[3], invd, m; };

__device__ __forceinline__ PixFwd photo_pixel(const float4 col, float alpha, const float* bg, const Exposure& X,
                                              const float gt[3], float rdk, bool mask_outliers)
{
    PixFwd p;
    const float T = 1.0f - alpha;
#pragma unroll
    for (int k = 0; k < 3; ++k) p.c[k] = (k == 0 ? col.x : (k == 1 ? col.y : col.z)) + T * bg[k];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        p.u[i] = X.e[4 * i] * p.c[0] + X.e[4 * i + 1] * p.c[1] + X.e[4 * i + 2] * p.c[2] + X.e[4 * i + 3];
        p.e[i] = fminf(fmaxf(p.u[i], 0.f), 1.f);
    }
    p.invd = 1.0f / col.w;
    p.m = 1.f;
    if (mask_outliers) { // h3dgsv3.py:433-435 (channel 1 is tested twice there, channel 2 never)
        const float e0 = rdk * fabsf(p.e[0] - gt[0]), e1 = rdk * fabsf(p.e[1] - gt[1]);
        p.m = (e0 > 0.2f || e1 > 0.2f) ? 0.f : 1.f;
    }
    return p;
}

__global__ __launch_bounds__(PHOTO_THREADS) void photometric_fwd_kernel(
    int64_t P, const float4* __restrict__ colors4, const float* __restrict__ alphas, const float* __restrict__ bg,
    const float* __restrict__ E, const float* __restrict__ gt, const float* __restrict__ mono, const float* __restrict__ rdk,
    int mask_outliers, float* __restrict__ image, float* __restrict__ gt_used, float* __restrict__ invdepth,
    float* __restrict__ partials /* [gridDim.x][2] */)
{
    __shared__ float red[PHOTO_THREADS / 64][2];
    Exposure X;
#pragma unroll
    for (int i = 0; i < 12; ++i) X.e[i] = E[i];
    const float b[3] = {bg[0], bg[1], bg[2]};
    float s_l1 = 0.f, s_d = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        const float g[3] = {gt[p], gt[P + p], gt[2 * P + p]};
        const float w = rdk[p], mo = mono[p];
        const PixFwd px = photo_pixel(colors4[p], alphas[p], b, X, g, w, mask_outliers != 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float em = px.e[i] * px.m, gm = g[i] * px.m;
            image[i * P + p] = em;
            if (mask_outliers) gt_used[i * P + p] = gm;
            s_l1 += w * fabsf(em - gm);
        }
        invdepth[p] = px.invd;
        s_d += w * fabsf(px.invd * px.m - mo * px.m);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float a = wave_sum_to_lane63(s_l1), d = wave_sum_to_lane63(s_d);
    if (lane == 63) { red[wv][0] = a; red[wv][1] = d; }
    __syncthreads();
    if (threadIdx.x < 2) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < PHOTO_THREADS / 64; ++k) t += red[k][threadIdx.x];
        partials[2 * blockIdx.x + threadIdx.x] = t;
    }
}

// stage 1 of the SSIM-map sum: fixed slices, fixed order
__global__ __launch_bounds__(256) void photometric_ssim_partials_kernel(const float* __restrict__ ssim_map, int64_t n,
                                                                        float* __restrict__ out /* [gridDim.x] */)
{
    __shared__ float red[4];
    float s = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += ssim_map[i];
    const float t = wave_sum_to_lane63(s);
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// stage 2: one workgroup; loss[0] = total, [1] = l1, [2] = ssim (mean), [3] = depth
__global__ __launch_bounds__(256) void photometric_loss_kernel(const float* __restrict__ l1d_partials, int n_l1d,
                                                               const float* __restrict__ ssim_partials, int n_ssim,
                                                               int64_t P, float lambda_dssim, float depth_weight,
                                                               float* __restrict__ loss, float* __restrict__ total_out /* nullable */)
{
    __shared__ float red[4][3];
    float a = 0.f, d = 0.f, s = 0.f;
    for (int i = threadIdx.x; i < n_l1d; i += 256) { a += l1d_partials[2 * i]; d += l1d_partials[2 * i + 1]; }
    for (int i = threadIdx.x; i < n_ssim; i += 256) s += ssim_partials[i];
    const float ta = wave_sum_to_lane63(a), td = wave_sum_to_lane63(d), ts = wave_sum_to_lane63(s);
    if ((threadIdx.x & 63) == 63) { red[threadIdx.x >> 6][0] = ta; red[threadIdx.x >> 6][1] = td; red[threadIdx.x >> 6][2] = ts; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l1 = ((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])) / (float)(3 * P);
        const float dl = ((red[0][1] + red[1][1]) + (red[2][1] + red[3][1])) / (float)P;
        const float ss = ((red[0][2] + red[1][2]) + (red[2][2] + red[3][2])) / (float)(3 * P);
        const float total = lambda_dssim * (1.f - ss) + (1.f - lambda_dssim) * l1 + depth_weight * dl;
        loss[0] = total; loss[1] = l1; loss[2] = ss; loss[3] = dl;
        if (total_out) *total_out = total;
    }
}

// One pixel of the backward chain: gradients of the rasteriser's colour / depth / alpha at this pixel, and the pixel's 12 exposure terms.
__device__ __forceinline__ void photo_pixel_bwd(const float4 col, float alpha, const float* b, const Exposure& X, const float g[3], float w, float mo,
                                                bool mask_outliers, const float vs[3], float vl, float l1_coeff, float depth_coeff,
                                                float4& v_col, float& v_alpha, float (&acc)[12])
{
    const PixFwd px = photo_pixel(col, alpha, b, X, g, w, mask_outliers);
    float vu[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float ve = (l1_coeff * w * sgnf(px.e[i] * px.m - g[i] * px.m) + vs[i]) * vl * px.m;
        vu[i] = (px.u[i] >= 0.f && px.u[i] <= 1.f) ? ve : 0.f; // clamp passes the gradient on [0,1]
    }
    float vc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) vc[j] = X.e[j] * vu[0] + X.e[4 + j] * vu[1] + X.e[8 + j] * vu[2];
    const float vinvd = depth_coeff * w * sgnf(px.invd * px.m - mo * px.m) * vl * px.m;
    v_col = make_float4(vc[0], vc[1], vc[2], -vinvd * px.invd * px.invd);
    v_alpha = -(vc[0] * b[0] + vc[1] * b[1] + vc[2] * b[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        acc[4 * i] += vu[i] * px.c[0]; acc[4 * i + 1] += vu[i] * px.c[1]; acc[4 * i + 2] += vu[i] * px.c[2]; acc[4 * i + 3] += vu[i];
    }
}

// V4 (round 4): a thread takes FOUR consecutive pixels -- every planar input (gt x 3, rdk, mono, alphas, the SSIM gradient x 3) and the
// alpha gradient move as one 16 B access per lane instead of four 4 B ones (1 KB instead of 256 B per wave instruction); needs P % 4 == 0
// and 16 B-aligned planes, else the one-pixel form runs.  Same arithmetic per pixel; the 12 exposure sums add their pixels in another order.
template <bool V4>
__global__ __launch_bounds__(PHOTO_THREADS) void photometric_bwd_kernel(
    int64_t P, const float4* __restrict__ colors4, const float* __restrict__ alphas, const float* __restrict__ bg,
    const float* __restrict__ E, const float* __restrict__ gt, const float* __restrict__ mono, const float* __restrict__ rdk,
    int mask_outliers, const float* __restrict__ v_image_ssim /* [3,P] for v_loss = 1 */, const float* __restrict__ v_loss,
    float l1_coeff /* (1 - lambda) / (3P) */, float depth_coeff /* w_depth / P */,
    float4* __restrict__ v_colors4, float* __restrict__ v_alphas, float* __restrict__ v_E /* [12], zeroed */)
{
    __shared__ float red[PHOTO_THREADS / 64][12];
    Exposure X;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { X.e[i] = E[i]; acc[i] = 0.f; }
    const float b[3] = {bg[0], bg[1], bg[2]};
    const float vl = v_loss[0];
    const bool mo_flag = mask_outliers != 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (V4) {
        const int64_t Q = P >> 2;
        const float4* gt4 = reinterpret_cast<const float4*>(gt);
        const float4* vs4 = reinterpret_cast<const float4*>(v_image_ssim);
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < Q; q += stride) {
            const float4 g0 = gt4[q], g1 = gt4[Q + q], g2 = gt4[2 * Q + q];
            const float4 s0 = vs4[q], s1 = vs4[Q + q], s2 = vs4[2 * Q + q];
            const float4 w4 = reinterpret_cast<const float4*>(rdk)[q], m4 = reinterpret_cast<const float4*>(mono)[q];
            const float4 a4 = reinterpret_cast<const float4*>(alphas)[q];
            const float4 c0 = colors4[4 * q], c1 = colors4[4 * q + 1], c2 = colors4[4 * q + 2], c3 = colors4[4 * q + 3];
            float4 vc0, vc1, vc2, vc3, va;
            { const float g[3] = {g0.x, g1.x, g2.x}, vs[3] = {s0.x, s1.x, s2.x};
              photo_pixel_bwd(c0, a4.x, b, X, g, w4.x, m4.x, mo_flag, vs, vl, l1_coeff, depth_coeff, vc0, va.x, acc); }
            { const float g[3] = {g0.y, g1.y, g2.y}, vs[3] = {s0.y, s1.y, s2.y};
              photo_pixel_bwd(c1, a4.y, b, X, g, w4.y, m4.y, mo_flag, vs, vl, l1_coeff, depth_coeff, vc1, va.y, acc); }
            { const float g[3] = {g0.z, g1.z, g2.z}, vs[3] = {s0.z, s1.z, s2.z};
              photo_pixel_bwd(c2, a4.z, b, X, g, w4.z, m4.z, mo_flag, vs, vl, l1_coeff, depth_coeff, vc2, va.z, acc); }
            { const float g[3] = {g0.w, g1.w, g2.w}, vs[3] = {s0.w, s1.w, s2.w};
              photo_pixel_bwd(c3, a4.w, b, X, g, w4.w, m4.w, mo_flag, vs, vl, l1_coeff, depth_coeff, vc3, va.w, acc); }
            v_colors4[4 * q] = vc0; v_colors4[4 * q + 1] = vc1; v_colors4[4 * q + 2] = vc2; v_colors4[4 * q + 3] = vc3;
            reinterpret_cast<float4*>(v_alphas)[q] = va;
        }
    } else {
        for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
            const float g[3] = {gt[p], gt[P + p], gt[2 * P + p]};
            const float vs[3] = {v_image_ssim[p], v_image_ssim[P + p], v_image_ssim[2 * P + p]};
            float4 vc; float va;
            photo_pixel_bwd(colors4[p], alphas[p], b, X, g, rdk[p], mono[p], mo_flag, vs, vl, l1_coeff, depth_coeff, vc, va, acc);
            v_colors4[p] = vc;
            v_alphas[p] = va;
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float s = wave_sum_to_lane63(acc[i]);
        if (lane == 63) red[wv][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < PHOTO_THREADS / 64; ++k) t += red[k][threadIdx.x];
        if (t != 0.f) unsafeAtomicAdd(v_E + threadIdx.x, t);
    }
}

// ---- 6D pose <-> view matrix (single thread; 9 numbers): device functions in pose6d.hpp --------------------
__global__ void pose6d_fwd_kernel(const float* __restrict__ r6, const float* __restrict__ t, float* __restrict__ Rt) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const Pose6 q = pose6_of(r6);
#pragma unroll
    for (int i = 0; i < 3; ++i) { Rt[4 * i] = q.b1[i]; Rt[4 * i + 1] = q.b2[i]; Rt[4 * i + 2] = q.b3[i]; Rt[4 * i + 3] = t[i]; }
    Rt[12] = 0.f; Rt[13] = 0.f; Rt[14] = 0.f; Rt[15] = 1.f;
}

// Keyframe.set_Rt (scene/keyframe.py:156-159): rW2C <- Rt[:3, :2], tW2C <- Rt[:3, 3], approx_centre = -Rt[:3, :3]^T Rt[:3, 3] as ONE launch
// (the reference: two copy_ and a slice / transpose / matmul / negate chain, ~8 launches per keyframe inside run_system.py's SLAM-keyframe loop)
__global__ void pose6d_set_kernel(const float* __restrict__ Rt, float* __restrict__ r6, float* __restrict__ t, float* __restrict__ centre) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float R[3][3], tt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { R[i][0] = Rt[4 * i]; R[i][1] = Rt[4 * i + 1]; R[i][2] = Rt[4 * i + 2]; tt[i] = Rt[4 * i + 3]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { r6[2 * i] = R[i][0]; r6[2 * i + 1] = R[i][1]; t[i] = tt[i]; }
#pragma unroll
    for (int j = 0; j < 3; ++j) centre[j] = -((R[0][j] * tt[0] + R[1][j] * tt[1]) + R[2][j] * tt[2]);
}

__global__ void pose6d_bwd_kernel(const float* __restrict__ r6, const float* __restrict__ v_Rt, float* __restrict__ v_r6,
                                  float* __restrict__ v_t) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    pose6d_bwd_body(r6, v_Rt, v_r6, v_t);
}

// pose6d_fwd + zero fill of up to two spans in the same launch (the one-call step: the voxel visibility mask the projection sets bits in,
// the voxel-feature gradient the LoD backward scatters into -- two ~4 us launches less per step).  Thread 0 of block 0 builds the matrix;
// every thread zeroes its share: 16 B per lane where the span is aligned, bytes at the ragged ends.
__device__ __forceinline__ void zero_span(unsigned char* p, int64_t nbytes, int64_t tid, int64_t nthreads) {
    if (p == nullptr || nbytes <= 0) return;
    const int64_t head = min(nbytes, (int64_t)((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15));
    const int64_t n16 = (nbytes - head) >> 4;
    float4* q = reinterpret_cast<float4*>(p + head);
    for (int64_t i = tid; i < n16; i += nthreads) q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = tid; i < head; i += nthreads) p[i] = 0;
    for (int64_t i = head + (n16 << 4) + tid; i < nbytes; i += nthreads) p[i] = 0;
}
__global__ __launch_bounds__(256) void pose6d_fwd_clear_kernel(const float* __restrict__ r6, const float* __restrict__ t, float* __restrict__ Rt,
                                                               unsigned char* a, int64_t na, unsigned char* b, int64_t nb) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
    zero_span(a, na, tid, nthreads);
    zero_span(b, nb, tid, nthreads);
    if (tid != 0) return;
    const Pose6 q = pose6_of(r6);
#pragma unroll
    for (int i = 0; i < 3; ++i) { Rt[4 * i] = q.b1[i]; Rt[4 * i + 1] = q.b2[i]; Rt[4 * i + 2] = q.b3[i]; Rt[4 * i + 3] = t[i]; }
    Rt[12] = 0.f; Rt[13] = 0.f; Rt[14] = 0.f; Rt[15] = 1.f;
}

// ---- visibility masks (h3dgsv3.py:695-698) -------------------------------------------------------------------
__global__ __launch_bounds__(256) void visibility_masks_kernel(int N, const int* __restrict__ radii, const int64_t* __restrict__ cls_id,
                                                               int64_t V, uint8_t* __restrict__ vis, uint8_t* __restrict__ gvis)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const int2 r = reinterpret_cast<const int2*>(radii)[g];
    const bool v = r.x > 0 && r.y > 0;
    vis[g] = v ? 1 : 0;
    if (v && gvis) {
        const int64_t c = cls_id[g];
        if (c >= 0 && c < V) gvis[c] = 1; // every writer stores the same byte
    }
}

} // namespace adk

static inline int photo_grid(int64_t P) {
    int64_t nb = adk::ceil_div(P, (int64_t)PHOTO_THREADS);
    return (int)(nb > PHOTO_MAX_BLOCKS ? PHOTO_MAX_BLOCKS : (nb < 1 ? 1 : nb));
}

// workspace: [PHOTO_MAX_BLOCKS][2] L1/depth partials | [PHOTO_SSIM_BLOCKS] SSIM partials
extern "C" int64_t adk_photometric_workspace_bytes(int W, int H)
{
    if (W < 0 || H < 0) return ADK_EINVAL;
    return (int64_t)(2 * PHOTO_MAX_BLOCKS + PHOTO_SSIM_BLOCKS) * (int64_t)sizeof(float);
}

extern "C" int adk_photometric_fwd(int W, int H, const float* colors4, const float* alphas, const float* bg,
                                   const float* exposure, const float* gt_image, const float* mono_idepth, const float* rdk,
                                   int mask_outliers, float* image, float* gt_used, float* invdepth, void* workspace,
                                   int64_t workspace_bytes, hipStream_t stream)
{
    if (W < 0 || H < 0) return ADK_EINVAL;
    const int64_t P = (int64_t)W * H;
    if (P == 0) return 0;
    if (!colors4 || !alphas || !bg || !exposure || !gt_image || !mono_idepth || !rdk || !image || !invdepth || !workspace) return ADK_EINVAL;
    if (mask_outliers && !gt_used) return ADK_EINVAL;
    if ((uintptr_t)colors4 & 15) return ADK_EINVAL;
    if (workspace_bytes < adk_photometric_workspace_bytes(W, H)) return ADK_EWORKSPACE;
    const int nb = photo_grid(P);
    hipLaunchKernelGGL(adk::photometric_fwd_kernel, dim3(nb), dim3(PHOTO_THREADS), 0, stream, P, (const float4*)colors4, alphas, bg,
                       exposure, gt_image, mono_idepth, rdk, mask_outliers, image, gt_used, invdepth, (float*)workspace);
    ADK_RETURN_LAST_ERROR();
}

// loss_out [4] = {loss, l1, ssim, depth}; must follow adk_photometric_fwd on the same workspace and stream.
extern "C" int adk_photometric_loss(int W, int H, const float* ssim_map, float lambda_dssim, float depth_weight,
                                    void* workspace, int64_t workspace_bytes, float* loss_out, hipStream_t stream)
{
    if (W < 0 || H < 0) return ADK_EINVAL;
    const int64_t P = (int64_t)W * H;
    if (P == 0) return ADK_EINVAL;
    if (!ssim_map || !workspace || !loss_out) return ADK_EINVAL;
    if (workspace_bytes < adk_photometric_workspace_bytes(W, H)) return ADK_EWORKSPACE;
    float* l1d = (float*)workspace;
    float* sp = l1d + 2 * PHOTO_MAX_BLOCKS;
    int ns = (int)adk::ceil_div(3 * P, (int64_t)1024);
    if (ns > PHOTO_SSIM_BLOCKS) ns = PHOTO_SSIM_BLOCKS;
    hipLaunchKernelGGL(adk::photometric_ssim_partials_kernel, dim3(ns), dim3(256), 0, stream, ssim_map, 3 * P, sp);
    hipLaunchKernelGGL(adk::photometric_loss_kernel, dim3(1), dim3(256), 0, stream, (const float*)l1d, photo_grid(P),
                       (const float*)sp, ns, P, lambda_dssim, depth_weight, loss_out, (float*)nullptr);
    ADK_RETURN_LAST_ERROR();
}

// The same four scalars from the per-strip sums of adk_fused_ssim_fwd_sums (n_sums of them) instead of the SSIM map: one
// launch, no pass over the map.  total_out (nullable): a second copy of loss_out[0] in a buffer of its own, so that a binding
// can hand out the differentiable scalar and the by-product vector as two tensors without a copy kernel.
extern "C" int adk_photometric_loss_sums(int W, int H, const float* ssim_block_sums, int64_t n_sums, float lambda_dssim,
                                         float depth_weight, void* workspace, int64_t workspace_bytes, float* loss_out,
                                         float* total_out, hipStream_t stream)
{
    if (W < 0 || H < 0) return ADK_EINVAL;
    const int64_t P = (int64_t)W * H;
    if (P == 0) return ADK_EINVAL;
    if (!ssim_block_sums || n_sums <= 0 || n_sums > 0x7fffffff || !workspace || !loss_out) return ADK_EINVAL;
    if (workspace_bytes < adk_photometric_workspace_bytes(W, H)) return ADK_EWORKSPACE;
    hipLaunchKernelGGL(adk::photometric_loss_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, photo_grid(P),
                       ssim_block_sums, (int)n_sums, P, lambda_dssim, depth_weight, loss_out, total_out);
    ADK_RETURN_LAST_ERROR();
}

// v_image_ssim: gradient of the SSIM term w.r.t. the image for v_loss = 1 (adk_fused_ssim_bwd with
// dL_scalar = -lambda / (3 W H)); v_loss: device scalar (autograd's incoming gradient of the loss).
extern "C" int adk_photometric_bwd(int W, int H, const float* colors4, const float* alphas, const float* bg,
                                   const float* exposure, const float* gt_image, const float* mono_idepth, const float* rdk,
                                   int mask_outliers, const float* v_image_ssim, const float* v_loss, float lambda_dssim,
                                   float depth_weight, float* v_colors4, float* v_alphas, float* v_exposure, hipStream_t stream)
{
    if (W < 0 || H < 0) return ADK_EINVAL;
    const int64_t P = (int64_t)W * H;
    if (!v_exposure) return ADK_EINVAL;
    int e = adk::clear_bytes(v_exposure, 12 * sizeof(float), stream); // a kernel, not hipMemsetAsync: hipGraph-safe
    if (e != 0) return e;
    if (P == 0) return 0;
    if (!colors4 || !alphas || !bg || !exposure || !gt_image || !mono_idepth || !rdk || !v_image_ssim || !v_loss || !v_colors4 || !v_alphas) return ADK_EINVAL;
    if (((uintptr_t)colors4 | (uintptr_t)v_colors4) & 15) return ADK_EINVAL;
    const bool v4 = ADK_PHOTO_V4 && (P & 3) == 0 &&
                    ((((uintptr_t)alphas | (uintptr_t)gt_image | (uintptr_t)mono_idepth | (uintptr_t)rdk | (uintptr_t)v_image_ssim | (uintptr_t)v_alphas) & 15) == 0);
    const float l1c = (1.f - lambda_dssim) / (float)(3 * P), dc = depth_weight / (float)P;
    if (v4) {
        // four pixels per thread; half as many workgroups as the one-pixel form would get (each ends in 12 same-address atomics)
        int64_t nb = adk::ceil_div(P >> 2, (int64_t)PHOTO_THREADS);
        nb = nb > PHOTO_MAX_BLOCKS / 2 ? PHOTO_MAX_BLOCKS / 2 : (nb < 1 ? 1 : nb);
        hipLaunchKernelGGL(adk::photometric_bwd_kernel<true>, dim3((unsigned)nb), dim3(PHOTO_THREADS), 0, stream, P, (const float4*)colors4, alphas, bg,
                           exposure, gt_image, mono_idepth, rdk, mask_outliers, v_image_ssim, v_loss, l1c, dc, (float4*)v_colors4, v_alphas, v_exposure);
    } else {
        hipLaunchKernelGGL(adk::photometric_bwd_kernel<false>, dim3(photo_grid(P)), dim3(PHOTO_THREADS), 0, stream, P, (const float4*)colors4, alphas, bg,
                           exposure, gt_image, mono_idepth, rdk, mask_outliers, v_image_ssim, v_loss, l1c, dc, (float4*)v_colors4, v_alphas, v_exposure);
    }
    ADK_RETURN_LAST_ERROR();
}

extern "C" int adk_pose6d_fwd(const float* r6, const float* t, float* Rt, hipStream_t stream)
{
    if (!r6 || !t || !Rt) return ADK_EINVAL;
    hipLaunchKernelGGL(adk::pose6d_fwd_kernel, dim3(1), dim3(64), 0, stream, r6, t, Rt);
    ADK_RETURN_LAST_ERROR();
}

int adk::pose6d_fwd_clear(const float* r6, const float* t, float* Rt, void* a, int64_t na, void* b, int64_t nb, hipStream_t stream)
{
    if (!r6 || !t || !Rt || na < 0 || nb < 0) return ADK_EINVAL;
    const int64_t work = adk::ceil_div((na > nb ? na : nb), 16);
    const int blocks = (int)(work <= 256 ? 1 : (adk::ceil_div(work, 256) > 2048 ? 2048 : adk::ceil_div(work, 256)));
    hipLaunchKernelGGL(adk::pose6d_fwd_clear_kernel, dim3(blocks), dim3(256), 0, stream, r6, t, Rt, static_cast<unsigned char*>(a), na,
                       static_cast<unsigned char*>(b), nb);
    ADK_RETURN_LAST_ERROR();
}

extern "C" int adk_pose6d_set(const float* Rt, float* r6, float* t, float* centre, hipStream_t stream)
{
    if (!Rt || !r6 || !t || !centre) return ADK_EINVAL;
    hipLaunchKernelGGL(adk::pose6d_set_kernel, dim3(1), dim3(64), 0, stream, Rt, r6, t, centre);
    ADK_RETURN_LAST_ERROR();
}

extern "C" int adk_pose6d_bwd(const float* r6, const float* v_Rt, float* v_r6, float* v_t, hipStream_t stream)
{
    if (!r6 || !v_Rt || !v_r6 || !v_t) return ADK_EINVAL;
    hipLaunchKernelGGL(adk::pose6d_bwd_kernel, dim3(1), dim3(64), 0, stream, r6, v_Rt, v_r6, v_t);
    ADK_RETURN_LAST_ERROR();
}

// vis [N] and gvis [V] are bytes (torch.bool storage); gvis may be NULL.
extern "C" int adk_visibility_masks(int N, const int* radii, const int64_t* cls_id, int64_t V, uint8_t* vis, uint8_t* gvis,
                                    hipStream_t stream)
{
    if (N < 0 || V < 0) return ADK_EINVAL;
    if (gvis && V > 0) {
        int e = adk::clear_bytes(gvis, V, stream);
        if (e != 0) return e;
    }
    if (N == 0) return 0;
    if (!radii || !vis || (gvis && !cls_id)) return ADK_EINVAL;
    hipLaunchKernelGGL(adk::visibility_masks_kernel, dim3((unsigned)adk::ceil_div(N, 256)), dim3(256), 0, stream, N, radii, cls_id, V, vis, gvis);
    ADK_RETURN_LAST_ERROR();
}
