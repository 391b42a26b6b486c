// Per-Gaussian stages of the rasteriser for gfx950: projection + SH colour (forward),
// and their joint backward.
//
// Replaces gsplat's fully_fused_projection fwd/bwd and spherical_harmonics fwd/bwd
// [UPSTREAM gsplat >= 1.5, not vendored; semantics per SURVEY.md App. A items 1, 2, 6]
// as invoked through gsplat.rendering.rasterization at
// Reconstruct/scene/scene_models/h3dgsv3.py:664-680 (packed=False, pinhole, classic).
//
// MI355X design: both per-Gaussian stages are pure HBM streaming, so they are FUSED -- one
// kernel reads each Gaussian once (44 B + 192 B of SH for visible ones) and writes one packed
// 48 B "splat record" that the tile kernels gather with three 16 B loads, plus the depth sort
// key and the tile count.  The backward likewise consumes the packed 48 B gradient record and
// emits every per-attribute gradient in one pass; the camera gradient (12 + 3 values summed
// over all Gaussians) is wave-reduced, block-reduced in LDS and lands with 15 atomics/block.
//
// Arithmetic policy: this file is compiled with -ffp-contract=off and the forward path is
// written as the same chain of single IEEE fp32 operations as oracle/gsplat_oracle.py:project,
// so depth sort keys, radii and tile ranges are bit-identical to the oracle.  log() on the
// radius path is evaluated in fp64 and rounded (one per Gaussian; free on an HBM-bound kernel).
#include "adk_common.hpp"
#include "adk_internal.hpp"
#include "lod_core.hpp"
#include "pose6d.hpp"

namespace adk {

#define ADK_ALPHA_THRESHOLD (1.0f / 255.0f)

struct Cam {
    float R[3][3];
    float t[3];
    float fx, fy, cx, cy;
    float campos[3];
};

__device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
    return (a0 * b0 + a1 * b1) + a2 * b2;
}

__device__ __forceinline__ Cam load_cam(const float* __restrict__ viewmat, const float* __restrict__ K) {
    Cam c;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) c.R[i][j] = viewmat[i * 4 + j];
        c.t[i] = viewmat[i * 4 + 3];
    }
    c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
    // campos = -R^-1 t via the adjugate (oracle: camera_position)
    const float (*R)[3] = c.R;
    const float c00 = R[1][1] * R[2][2] - R[1][2] * R[2][1];
    const float c01 = R[0][2] * R[2][1] - R[0][1] * R[2][2];
    const float c02 = R[0][1] * R[1][2] - R[0][2] * R[1][1];
    const float c10 = R[1][2] * R[2][0] - R[1][0] * R[2][2];
    const float c11 = R[0][0] * R[2][2] - R[0][2] * R[2][0];
    const float c12 = R[0][2] * R[1][0] - R[0][0] * R[1][2];
    const float c20 = R[1][0] * R[2][1] - R[1][1] * R[2][0];
    const float c21 = R[0][1] * R[2][0] - R[0][0] * R[2][1];
    const float c22 = R[0][0] * R[1][1] - R[0][1] * R[1][0];
    const float det = (R[0][0] * c00 + R[0][1] * c10) + R[0][2] * c20;
    const float id = 1.0f / det;
    c.campos[0] = -(((c00 * id) * c.t[0] + (c01 * id) * c.t[1]) + (c02 * id) * c.t[2]);
    c.campos[1] = -(((c10 * id) * c.t[0] + (c11 * id) * c.t[1]) + (c12 * id) * c.t[2]);
    c.campos[2] = -(((c20 * id) * c.t[0] + (c21 * id) * c.t[1]) + (c22 * id) * c.t[2]);
    return c;
}

// Everything the forward computes that the backward needs again.
struct Proj {
    float mc[3];       // camera-space mean
    float Rq[3][3];    // rotation from the normalised quaternion
    float qn[4];       // normalised quaternion (w,x,y,z)
    float inv_qnorm;
    float M[3][3];     // Rq * diag(s)
    float cov[3][3];   // world covariance (symmetric)
    float C[3][3];     // camera-space covariance (symmetric)
    float rz, rz2, tx, ty;
    bool x_in, y_in;   // x/z, y/z inside the fov clamp
    float j00, j02, j11, j12;
    float c00, c01, c11; // blurred 2D covariance
    float det;
    float m2x, m2y;
    float ca, cb, cc;  // conic
};

// Returns false if culled by near/far or det<=0 (before the radius stage).
__device__ __forceinline__ bool project_core(const Cam& cam, float x, float y, float z, const float q[4],
                                             const float s[3], int width, int height, float eps2d,
                                             float near_plane, float far_plane, Proj& P)
{
    const float (*R)[3] = cam.R;
#pragma unroll
    for (int i = 0; i < 3; ++i) P.mc[i] = dot3(R[i][0], x, R[i][1], y, R[i][2], z) + cam.t[i];
    if (P.mc[2] < near_plane || P.mc[2] > far_plane) return false;

    {
        float w = q[0], qx = q[1], qy = q[2], qz = q[3];
        const float inv_norm = 1.0f / sqrtf(((qx * qx + qy * qy) + qz * qz) + w * w);
        qx *= inv_norm; qy *= inv_norm; qz *= inv_norm; w *= inv_norm;
        P.inv_qnorm = inv_norm;
        P.qn[0] = w; P.qn[1] = qx; P.qn[2] = qy; P.qn[3] = qz;
        const float x2 = qx * qx, y2 = qy * qy, z2 = qz * qz;
        const float xy = qx * qy, xz = qx * qz, yz = qy * qz;
        const float wx = w * qx, wy = w * qy, wz = w * qz;
        P.Rq[0][0] = 1.0f - 2.0f * (y2 + z2); P.Rq[0][1] = 2.0f * (xy - wz); P.Rq[0][2] = 2.0f * (xz + wy);
        P.Rq[1][0] = 2.0f * (xy + wz); P.Rq[1][1] = 1.0f - 2.0f * (x2 + z2); P.Rq[1][2] = 2.0f * (yz - wx);
        P.Rq[2][0] = 2.0f * (xz - wy); P.Rq[2][1] = 2.0f * (yz + wx); P.Rq[2][2] = 1.0f - 2.0f * (x2 + y2);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) P.M[i][j] = P.Rq[i][j] * s[j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) {
            P.cov[i][j] = dot3(P.M[i][0], P.M[j][0], P.M[i][1], P.M[j][1], P.M[i][2], P.M[j][2]);
            P.cov[j][i] = P.cov[i][j];
        }
    float A[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) A[i][j] = dot3(R[i][0], P.cov[0][j], R[i][1], P.cov[1][j], R[i][2], P.cov[2][j]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) {
            P.C[i][j] = dot3(A[i][0], R[j][0], A[i][1], R[j][1], A[i][2], R[j][2]);
            P.C[j][i] = P.C[i][j];
        }

    const float fx = cam.fx, fy = cam.fy, cx = cam.cx, cy = cam.cy;
    const float tan_fovx = (0.5f * (float)width) / fx;
    const float tan_fovy = (0.5f * (float)height) / fy;
    const float lim_x_pos = ((float)width - cx) / fx + 0.3f * tan_fovx;
    const float lim_x_neg = cx / fx + 0.3f * tan_fovx;
    const float lim_y_pos = ((float)height - cy) / fy + 0.3f * tan_fovy;
    const float lim_y_neg = cy / fy + 0.3f * tan_fovy;
    const float rz = 1.0f / P.mc[2];
    const float rz2 = rz * rz;
    const float xr = P.mc[0] * rz, yr = P.mc[1] * rz;
    P.x_in = (xr <= lim_x_pos) && (xr >= -lim_x_neg);
    P.y_in = (yr <= lim_y_pos) && (yr >= -lim_y_neg);
    const float tx = P.mc[2] * fminf(lim_x_pos, fmaxf(-lim_x_neg, xr));
    const float ty = P.mc[2] * fminf(lim_y_pos, fmaxf(-lim_y_neg, yr));
    P.rz = rz; P.rz2 = rz2; P.tx = tx; P.ty = ty;
    const float j00 = fx * rz;
    const float j02 = ((-fx) * tx) * rz2;
    const float j11 = fy * rz;
    const float j12 = ((-fy) * ty) * rz2;
    P.j00 = j00; P.j02 = j02; P.j11 = j11; P.j12 = j12;
    const float v00 = j00 * P.C[0][0] + j02 * P.C[0][2];
    const float v01 = j00 * P.C[0][1] + j02 * P.C[1][2];
    const float v02 = j00 * P.C[0][2] + j02 * P.C[2][2];
    const float v11 = j11 * P.C[1][1] + j12 * P.C[1][2];
    const float v12 = j11 * P.C[1][2] + j12 * P.C[2][2];
    float c00 = v00 * j00 + v02 * j02;
    const float c01 = v01 * j11 + v02 * j12;
    float c11 = v11 * j11 + v12 * j12;
    P.m2x = (fx * P.mc[0]) * rz + cx;
    P.m2y = (fy * P.mc[1]) * rz + cy;
    c00 = c00 + eps2d;
    c11 = c11 + eps2d;
    const float det = c00 * c11 - c01 * c01;
    P.c00 = c00; P.c01 = c01; P.c11 = c11; P.det = det;
    if (!(det > 0.0f)) return false;
    const float inv_det = 1.0f / det;
    P.ca = c11 * inv_det;
    P.cb = (-c01) * inv_det;
    P.cc = c00 * inv_det;
    return true;
}

// SH basis for the normalised direction (x,y,z); b[0..(deg+1)^2).
__device__ __forceinline__ void sh_basis(int degree, float x, float y, float z, float* b) {
    b[0] = 0.2820947917738781f;
    if (degree >= 1) {
        b[1] = -0.48860251190292f * y; b[2] = 0.48860251190292f * z; b[3] = -0.48860251190292f * x;
    }
    if (degree >= 2) {
        const float z2 = z * z;
        const float fTmp0B = -1.092548430592079f * z;
        const float fC1 = x * x - y * y;
        const float fS1 = 2.0f * x * y;
        b[4] = 0.5462742152960395f * fS1;
        b[5] = fTmp0B * y;
        b[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
        b[7] = fTmp0B * x;
        b[8] = 0.5462742152960395f * fC1;
        if (degree >= 3) {
            const float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
            const float fTmp1B = 1.445305721320277f * z;
            const float fC2 = x * fC1 - y * fS1;
            const float fS2 = x * fS1 + y * fC1;
            b[9] = -0.5900435899266435f * fS2;
            b[10] = fTmp1B * fS1;
            b[11] = fTmp0C * y;
            b[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
            b[13] = fTmp0C * x;
            b[14] = fTmp1B * fC1;
            b[15] = -0.5900435899266435f * fC2;
        }
    }
}

// d(basis)/d(x,y,z)
__device__ __forceinline__ void sh_basis_grad(int degree, float x, float y, float z, float* dx, float* dy, float* dz) {
    dx[0] = dy[0] = dz[0] = 0.f;
    if (degree >= 1) {
        const float C1 = 0.48860251190292f;
        dx[1] = 0.f; dy[1] = -C1; dz[1] = 0.f;
        dx[2] = 0.f; dy[2] = 0.f; dz[2] = C1;
        dx[3] = -C1; dy[3] = 0.f; dz[3] = 0.f;
    }
    if (degree >= 2) {
        const float k4 = 0.5462742152960395f, k5 = -1.092548430592079f, k6 = 0.9461746957575601f;
        dx[4] = 2.f * k4 * y; dy[4] = 2.f * k4 * x; dz[4] = 0.f;
        dx[5] = 0.f; dy[5] = k5 * z; dz[5] = k5 * y;
        dx[6] = 0.f; dy[6] = 0.f; dz[6] = 2.f * k6 * z;
        dx[7] = k5 * z; dy[7] = 0.f; dz[7] = k5 * x;
        dx[8] = 2.f * k4 * x; dy[8] = -2.f * k4 * y; dz[8] = 0.f;
    }
    if (degree >= 3) {
        const float k9 = -0.5900435899266435f, k10 = 1.445305721320277f;
        const float z2 = z * z;
        const float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
        const float dTmp0C = -4.570457994644658f * z;
        dx[9] = k9 * 6.f * x * y; dy[9] = k9 * 3.f * (x * x - y * y); dz[9] = 0.f;
        dx[10] = 2.f * k10 * y * z; dy[10] = 2.f * k10 * x * z; dz[10] = 2.f * k10 * x * y;
        dx[11] = 0.f; dy[11] = fTmp0C; dz[11] = y * dTmp0C;
        dx[12] = 0.f; dy[12] = 0.f; dz[12] = 3.f * 1.865881662950577f * z2 - 1.119528997770346f;
        dx[13] = fTmp0C; dy[13] = 0.f; dz[13] = x * dTmp0C;
        dx[14] = 2.f * k10 * z * x; dy[14] = -2.f * k10 * z * y; dz[14] = k10 * (x * x - y * y);
        dx[15] = k9 * 3.f * (x * x - y * y); dy[15] = -k9 * 6.f * x * y; dz[15] = 0.f;
    }
}

__device__ __forceinline__ void tile_range(float mx, float my, float rx, float ry, int tile_w, int tile_h,
                                           int& x0, int& x1, int& y0, int& y1)
{
    const float ts = 16.0f;
    const float tx = mx / ts, ty = my / ts, trx = rx / ts, try_ = ry / ts;
    x0 = (int)fminf(fmaxf(floorf(tx - trx), 0.f), (float)tile_w);
    x1 = (int)fminf(fmaxf(ceilf(tx + trx), 0.f), (float)tile_w);
    y0 = (int)fminf(fmaxf(floorf(ty - try_), 0.f), (float)tile_h);
    y1 = (int)fminf(fmaxf(ceilf(ty + try_), 0.f), (float)tile_h);
}

// ---------------------------------------------------------------------------------- forward
// rec layout per Gaussian (12 floats = three float4):
//   [0] mean2d.x  [1] mean2d.y  [2] opacity  [3] radius_x
//   [4] conic a   [5] conic b   [6] conic c  [7] radius_y
//   [8] r         [9] g         [10] b       [11] depth      (channels per `color_mode`)
// color_mode: 0 = SH -> rgb (+depth in ch 3); 1 = colors_in[N,3] copied (+depth); 2 = depth only in ch 0.
// One Gaussian of the forward: everything after its inputs are in registers (the stand-alone kernel loads them; the fused LoD + projection
// kernel of the one-call step has just computed them).
template <int SH_DEG>
__device__ __forceinline__ void project_fwd_one(
    const int g, const Cam& cam, const float x, const float y, const float z, const float (&q)[4], const float (&s)[3], const float opac,
    const float* __restrict__ colors_in, const float* __restrict__ sh_rest, int sh_K, int color_mode, int width, int height, int tile_w,
    int tile_h, float eps2d, float near_plane, float far_plane, float radius_clip, int inv_depth, float* __restrict__ rec,
    int32_t* __restrict__ radii, uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ gauss_ids, int32_t* __restrict__ tiles_per_gauss,
    const ProjectMasks& masks)
{
    Proj P;
    bool valid = project_core(cam, x, y, z, q, s, width, height, eps2d, near_plane, far_plane, P);
    float rad_x = 0.f, rad_y = 0.f;
    if (valid) {
        const float thr = ADK_ALPHA_THRESHOLD;
        if (opac < thr) valid = false;
        else {
            const float lg = (float)log((double)(opac / thr));
            const float extend = fminf(3.33f, sqrtf(2.0f * lg));
            const float b = 0.5f * (P.c00 + P.c11);
            const float tmp = sqrtf(fmaxf(0.01f, b * b - P.det));
            const float v1 = b + tmp;
            const float r1 = extend * sqrtf(v1);
            rad_x = ceilf(fminf(extend * sqrtf(P.c00), r1));
            rad_y = ceilf(fminf(extend * sqrtf(P.c11), r1));
            if (rad_x <= radius_clip && rad_y <= radius_clip) valid = false;
            else if (P.m2x + rad_x <= 0.f || P.m2x - rad_x >= (float)width || P.m2y + rad_y <= 0.f ||
                     P.m2y - rad_y >= (float)height) valid = false;
        }
    }

    float4* r4 = reinterpret_cast<float4*>(rec) + 3 * (int64_t)g;
    gauss_ids[g] = (uint32_t)g;
    if (!valid) {
        radii[2 * g] = 0; radii[2 * g + 1] = 0;
        if (masks.vis) masks.vis[g] = 0;
        depth_keys[g] = 0xFFFFFFFFu; // sorts behind every visible Gaussian
        tiles_per_gauss[g] = 0;
        r4[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        r4[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        r4[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }

    radii[2 * g] = (int32_t)rad_x; radii[2 * g + 1] = (int32_t)rad_y;
    if (masks.vis) { // adk_visibility_masks' rule on the radii just written (h3dgsv3.py:695-698)
        const bool seen = (int32_t)rad_x > 0 && (int32_t)rad_y > 0;
        masks.vis[g] = seen ? 1 : 0;
        if (seen && masks.gvis) {
            const int64_t c = masks.cls_id[g];
            if (c >= 0 && c < masks.V) masks.gvis[c] = 1; // every writer stores the same byte
        }
    }
    depth_keys[g] = __float_as_uint(P.mc[2]);
    int x0, x1, y0, y1;
    tile_range(P.m2x, P.m2y, rad_x, rad_y, tile_w, tile_h, x0, x1, y0, y1);
    tiles_per_gauss[g] = (y1 - y0) * (x1 - x0);

    // depth channel: z (gsplat "RGB+D") or 1/z (the on-the-fly-nvs GaussianRasterizer adapter)
    const float dchan = inv_depth ? P.rz : P.mc[2];
    float col[4] = {0.f, 0.f, 0.f, dchan};
    if (color_mode == 0) {
        float dx = x - cam.campos[0], dy = y - cam.campos[1], dz = z - cam.campos[2];
        const float inorm = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
        dx *= inorm; dy *= inorm; dz *= inorm;
        float b[16];
        sh_basis(SH_DEG, dx, dy, dz, b);
        constexpr int NB = (SH_DEG + 1) * (SH_DEG + 1);
        float acc[3] = {0.f, 0.f, 0.f};
        if (sh_rest) { // band 0 in colors_in [N,1,3], bands 1.. in sh_rest [N,sh_K-1,3] (ARTDECO's f_dc / f_rest)
            const float* c0 = colors_in + (int64_t)g * 3;
            const float* cr = sh_rest + (int64_t)g * (sh_K - 1) * 3;
            acc[0] = b[0] * c0[0]; acc[1] = b[0] * c0[1]; acc[2] = b[0] * c0[2];
#pragma unroll
            for (int k = 1; k < NB; ++k) { acc[0] += b[k] * cr[3 * k - 3]; acc[1] += b[k] * cr[3 * k - 2]; acc[2] += b[k] * cr[3 * k - 1]; }
        } else {
        const float* c = colors_in + (int64_t)g * sh_K * 3;
        if (NB == 16 && sh_K == 16) {            const float4* c4 = reinterpret_cast<const float4*>(c); // 48 floats, 192 B aligned stride
            float v[48];
#pragma unroll
            for (int i = 0; i < 12; ++i) { const float4 t = c4[i]; v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w; }
#pragma unroll
            for (int k = 0; k < 16; ++k) { acc[0] += b[k] * v[3 * k]; acc[1] += b[k] * v[3 * k + 1]; acc[2] += b[k] * v[3 * k + 2]; }
        } else {
#pragma unroll
            for (int k = 0; k < NB; ++k) { acc[0] += b[k] * c[3 * k]; acc[1] += b[k] * c[3 * k + 1]; acc[2] += b[k] * c[3 * k + 2]; }
        }
        }
        col[0] = fmaxf(acc[0] + 0.5f, 0.f); col[1] = fmaxf(acc[1] + 0.5f, 0.f); col[2] = fmaxf(acc[2] + 0.5f, 0.f);
    } else if (color_mode == 1) {
        col[0] = colors_in[3 * g]; col[1] = colors_in[3 * g + 1]; col[2] = colors_in[3 * g + 2];
    } else {
        col[0] = dchan; col[3] = 0.f;
    }
    r4[0] = make_float4(P.m2x, P.m2y, opac, rad_x);
    r4[1] = make_float4(P.ca, P.cb, P.cc, rad_y);
    r4[2] = make_float4(col[0], col[1], col[2], col[3]);
}

template <int SH_DEG>
__global__ __launch_bounds__(256) void project_fwd_kernel(
    int N, const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
    const float* __restrict__ opacities, const float* __restrict__ colors_in, const float* __restrict__ sh_rest,
    int sh_K, int color_mode,
    const float* __restrict__ viewmat, const float* __restrict__ Kmat, int width, int height, int tile_w, int tile_h,
    float eps2d, float near_plane, float far_plane, float radius_clip, int inv_depth,
    float* __restrict__ rec, int32_t* __restrict__ radii, uint32_t* __restrict__ depth_keys,
    uint32_t* __restrict__ gauss_ids, int32_t* __restrict__ tiles_per_gauss, const ProjectMasks masks)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const Cam cam = load_cam(viewmat, Kmat);

    const float x = means[3 * g], y = means[3 * g + 1], z = means[3 * g + 2];
    const float4 q4 = reinterpret_cast<const float4*>(quats)[g];
    const float q[4] = {q4.x, q4.y, q4.z, q4.w};
    const float s[3] = {scales[3 * g], scales[3 * g + 1], scales[3 * g + 2]};
    const float opac = opacities[g];
    project_fwd_one<SH_DEG>(g, cam, x, y, z, q, s, opac, colors_in, sh_rest, sh_K, color_mode, width, height, tile_w, tile_h, eps2d, near_plane,
                            far_plane, radius_clip, inv_depth, rec, radii, depth_keys, gauss_ids, tiles_per_gauss, masks);
}

// The stand-alone LoD / mlp_cov forward (adk_lod_params_fwd): here rather than in lod_params.hip so that it and the fused kernel below call ONE
// compiled lod_forward_one.
__global__ __launch_bounds__(256) void lod_params_fwd_kernel(
    int N, const float* __restrict__ xyz, const float* __restrict__ opacity_raw, const float* __restrict__ scaling_raw,
    const float* __restrict__ rotation, const float* __restrict__ local_feat, const float* __restrict__ global_feat,
    const int64_t* __restrict__ cls_id, const float* __restrict__ d_max, const float* __restrict__ W1,
    const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ viewmat, float* __restrict__ opac_eff, float* __restrict__ scale_eff,
    float* __restrict__ quat_eff, uint8_t* __restrict__ selected)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const LodOut o = lod_forward_one(g, xyz, opacity_raw, scaling_raw, rotation, local_feat, global_feat, cls_id, d_max, W1, b1, W2, b2, viewmat);
    selected[g] = o.selected ? 1 : 0;
    opac_eff[g] = o.opac;
    scale_eff[3 * g] = o.scale[0]; scale_eff[3 * g + 1] = o.scale[1]; scale_eff[3 * g + 2] = o.scale[2];
    reinterpret_cast<float4*>(quat_eff)[g] = o.quat;
}

// LoD / mlp_cov forward + projection forward of ONE Gaussian in one kernel (the one-call step; SH colours in ARTDECO's split layout).  The first
// phase is 1 248 FMAs per Gaussian on scalar-cache weights, the second streams 260 B per Gaussian: run back to back as two kernels they
// cannot overlap, inside one the waves of a SIMD are in different phases.  The activated parameters are still written (the backward reads
// them) but never re-read, and the arithmetic of each phase is its own file's (lod_core.hpp pins the contraction mode of lod_params.hip).
#ifndef ADK_LOD_PROJECT_WAVES
#define ADK_LOD_PROJECT_WAVES 6
#endif
template <int SH_DEG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ADK_LOD_PROJECT_WAVES, 8))) void lod_project_fwd_kernel(
    int N, const float* __restrict__ xyz, const float* __restrict__ opacity_raw, const float* __restrict__ scaling_raw,
    const float* __restrict__ rotation, const float* __restrict__ local_feat, const float* __restrict__ global_feat,
    const int64_t* __restrict__ cls_id, const float* __restrict__ d_max, const float* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ opac_eff, float* __restrict__ scale_eff,
    float* __restrict__ quat_eff, uint8_t* __restrict__ selected,
    const float* __restrict__ colors_in, const float* __restrict__ sh_rest, int sh_K,
    const float* __restrict__ viewmat, const float* __restrict__ Kmat, int width, int height, int tile_w, int tile_h,
    float eps2d, float near_plane, float far_plane, float radius_clip,
    float* __restrict__ rec, int32_t* __restrict__ radii, uint32_t* __restrict__ depth_keys,
    uint32_t* __restrict__ gauss_ids, int32_t* __restrict__ tiles_per_gauss, const ProjectMasks masks)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const LodOut o = lod_forward_one(g, xyz, opacity_raw, scaling_raw, rotation, local_feat, global_feat, cls_id, d_max, W1, b1, W2, b2, viewmat);
    selected[g] = o.selected ? 1 : 0;
    opac_eff[g] = o.opac;
    scale_eff[3 * g] = o.scale[0]; scale_eff[3 * g + 1] = o.scale[1]; scale_eff[3 * g + 2] = o.scale[2];
    reinterpret_cast<float4*>(quat_eff)[g] = o.quat;

    // The projection phase sees its inputs through values the optimiser cannot identify with the first phase's: both phases invert the view
    // matrix and both read the position, and a product with two users is no longer folded into the first phase's FMAs -- the camera centre (and
    // with it the fade factor of some Gaussians) would move by an ulp against the stand-alone LoD kernel.
    const float* vm_proj = viewmat;
    const float* xyz_proj = xyz;
    asm volatile("" : "+s"(vm_proj), "+s"(xyz_proj));
    float q[4] = {o.quat.x, o.quat.y, o.quat.z, o.quat.w};
    float s[3] = {o.scale[0], o.scale[1], o.scale[2]};
    float opac = o.opac;
    asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(opac));
    const Cam cam = load_cam(vm_proj, Kmat);
    const float x = xyz_proj[3 * g], y = xyz_proj[3 * g + 1], z = xyz_proj[3 * g + 2];
    project_fwd_one<SH_DEG>(g, cam, x, y, z, q, s, opac, colors_in, sh_rest, sh_K, 0, width, height, tile_w, tile_h, eps2d, near_plane, far_plane,
                            radius_clip, 0, rec, radii, depth_keys, gauss_ids, tiles_per_gauss, masks);
}

// Optional in-kernel optimiser for the SH coefficients (see adk_project_bwd_adam): the gradient of coefficient
// (k, ch) is the product b_k * v_rgb[ch] of two numbers this thread already holds, and the coefficient itself was
// just read for the forward value -- so the sparse-Adam update of f_dc / f_rest is applied right here instead of
// writing 48 gradients per Gaussian for a second kernel to read back together with the parameters.
struct ColorAdam {
    float* m_dc; float* v_dc; float* m_rest; float* v_rest; // exp_avg / exp_avg_sq of colors [N,1,3] and sh_rest [N,K-1,3]
    const float* lr_dc; const float* lr_rest;               // 0-dim device tensors (optimizers.py:70-73)
    float b1, b2, eps;
};

// identical expression order to adam.hip:adam_elem (bit-identical updates; this file is built -ffp-contract=off too)
__device__ __forceinline__ void color_adam_elem(float* __restrict__ p_ptr, float p, float g, float* __restrict__ m_ptr,
                                                float* __restrict__ v_ptr, float lr, float b1, float b2, float eps)
{
    const float m = b1 * *m_ptr + (1.0f - b1) * g;
    const float v = b2 * *v_ptr + (1.0f - b2) * g * g;
    const float step = -lr * m / (sqrtf(v) + eps);
    *m_ptr = m; *v_ptr = v; *p_ptr = p + step;
}

// ---------------------------------------------------------------------------------- backward
// v_rec layout mirrors rec: [0] s_x [1] s_y [2] v_opacity | [4] v_ca [5] v_cb [6] v_cc | [8..11] v_colour channels,
// where (s_x, s_y) = sum over pixels of v_sigma * (mean2d - pixel): the raster backward leaves the multiplication by the
// conic, v_mean2d = conic (s_x, s_y)^T, to this kernel (once per Gaussian instead of once per (splat, pixel)).
// cam_grad: 16 DOUBLES (128 B, 8-byte aligned; round 6) -- v_R (9, row-major) | v_t (3) | v_campos (3) | pad, accumulated with fp64 atomics.
#ifndef ADK_PROJECT_BWD_WAVES
#define ADK_PROJECT_BWD_WAVES 4
#endif
template <int SH_DEG, bool FUSE_ADAM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ADK_PROJECT_BWD_WAVES, 8))) void project_bwd_kernel(
    int N, const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
    const float* colors_in, const float* sh_rest, int sh_K, int color_mode,
    const float* __restrict__ viewmat, const float* __restrict__ Kmat, int width, int height,
    float eps2d, float near_plane, float far_plane, int inv_depth,
    const int32_t* __restrict__ radii, const float* __restrict__ v_rec,
    float* __restrict__ v_means, float* __restrict__ v_quats, float* __restrict__ v_scales,
    float* __restrict__ v_opacities, float* __restrict__ v_colors, float* __restrict__ v_sh_rest,
    float* __restrict__ cam_grad, ColorAdam opt)
{
    __shared__ double red[4][16];
    // FUSE_ADAM: per-Gaussian (b_k, v_rgb, live) for the coalesced optimiser phase (17 B-conflict-free row pitch)
    __shared__ float adam_b[FUSE_ADAM ? 256 : 1][17];
    __shared__ float adam_v[FUSE_ADAM ? 256 : 1][3];
    __shared__ uint8_t adam_live[FUSE_ADAM ? 256 : 1];
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const Cam cam = load_cam(viewmat, Kmat);
    float cg[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) cg[i] = 0.f;

    const bool live = (g < N) && (radii[2 * g] > 0) && (radii[2 * g + 1] > 0);
    if (FUSE_ADAM) adam_live[threadIdx.x] = live ? 1 : 0;
    if (g < N && !live) {
        if (v_means) { v_means[3 * g] = 0.f; v_means[3 * g + 1] = 0.f; v_means[3 * g + 2] = 0.f; }
        if (v_quats) reinterpret_cast<float4*>(v_quats)[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v_scales) { v_scales[3 * g] = 0.f; v_scales[3 * g + 1] = 0.f; v_scales[3 * g + 2] = 0.f; }
        if (v_opacities) v_opacities[g] = 0.f;
        if (v_colors) {
            if (color_mode == 0 && sh_rest) {
                v_colors[3 * g] = 0.f; v_colors[3 * g + 1] = 0.f; v_colors[3 * g + 2] = 0.f;
                float* o = v_sh_rest + (int64_t)g * (sh_K - 1) * 3; for (int i = 0; i < (sh_K - 1) * 3; ++i) o[i] = 0.f;
            }
            else if (color_mode == 0) { float* o = v_colors + (int64_t)g * sh_K * 3; for (int i = 0; i < sh_K * 3; ++i) o[i] = 0.f; }
            else if (color_mode == 1) { v_colors[3 * g] = 0.f; v_colors[3 * g + 1] = 0.f; v_colors[3 * g + 2] = 0.f; }
        }
    }
    if (live) {
        const float x = means[3 * g], y = means[3 * g + 1], z = means[3 * g + 2];
        const float4 q4 = reinterpret_cast<const float4*>(quats)[g];
        const float q[4] = {q4.x, q4.y, q4.z, q4.w};
        const float s[3] = {scales[3 * g], scales[3 * g + 1], scales[3 * g + 2]};
        Proj P;
        project_core(cam, x, y, z, q, s, width, height, eps2d, near_plane, far_plane, P);

        const float4* vr = reinterpret_cast<const float4*>(v_rec) + 3 * (int64_t)g;
        const float4 g0 = vr[0], g1 = vr[1], g2 = vr[2];
        const float v_mx = P.ca * g0.x + P.cb * g0.y, v_my = P.cb * g0.x + P.cc * g0.y, v_op = g0.z;
        const float v_ca = g1.x, v_cb = g1.y, v_cc = g1.z;
        float v_col[3] = {g2.x, g2.y, g2.z};
        float v_depth = g2.w;
        if (color_mode == 2) { v_depth = g2.x; v_col[0] = v_col[1] = v_col[2] = 0.f; }

        // (1) conic -> blurred cov2d:  v_X = -Y G Y,  G = [[v_ca, v_cb/2],[v_cb/2, v_cc]]
        const float Ya = P.ca, Yb = P.cb, Yc = P.cc;
        const float G00 = v_ca, G01 = 0.5f * v_cb, G11 = v_cc;
        const float T00 = Ya * G00 + Yb * G01, T01 = Ya * G01 + Yb * G11;
        const float T10 = Yb * G00 + Yc * G01, T11 = Yb * G01 + Yc * G11;
        const float X00 = -(T00 * Ya + T01 * Yb);
        const float X01 = -(T00 * Yb + T01 * Yc);
        const float X11 = -(T10 * Yb + T11 * Yc);

        // (2) perspective projection vjp
        const float fx = cam.fx, fy = cam.fy;
        const float j00 = P.j00, j02 = P.j02, j11 = P.j11, j12 = P.j12;
        // v_C = J^T X J   (J = [[j00,0,j02],[0,j11,j12]])
        float vC[3][3];
        vC[0][0] = j00 * X00 * j00;
        vC[0][1] = j00 * X01 * j11;
        vC[0][2] = j00 * (X00 * j02 + X01 * j12);
        vC[1][1] = j11 * X11 * j11;
        vC[1][2] = j11 * (X01 * j02 + X11 * j12);
        vC[2][2] = j02 * (X00 * j02 + X01 * j12) + j12 * (X01 * j02 + X11 * j12);
        vC[1][0] = vC[0][1]; vC[2][0] = vC[0][2]; vC[2][1] = vC[1][2];
        // v_J = 2 X J C  (2x3)
        float JC[2][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            JC[0][k] = j00 * P.C[0][k] + j02 * P.C[2][k];
            JC[1][k] = j11 * P.C[1][k] + j12 * P.C[2][k];
        }
        float vJ[2][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            vJ[0][k] = 2.f * (X00 * JC[0][k] + X01 * JC[1][k]);
            vJ[1][k] = 2.f * (X01 * JC[0][k] + X11 * JC[1][k]);
        }
        const float rz = P.rz, rz2 = P.rz2, rz3 = rz2 * rz;
        float vmc[3];
        vmc[0] = fx * rz * v_mx;
        vmc[1] = fy * rz * v_my;
        vmc[2] = -(fx * P.mc[0] * v_mx + fy * P.mc[1] * v_my) * rz2;
        if (P.x_in) vmc[0] += -fx * rz2 * vJ[0][2]; else vmc[2] += -fx * rz3 * vJ[0][2] * P.tx;
        if (P.y_in) vmc[1] += -fy * rz2 * vJ[1][2]; else vmc[2] += -fy * rz3 * vJ[1][2] * P.ty;
        vmc[2] += -fx * rz2 * vJ[0][0] - fy * rz2 * vJ[1][1] + 2.f * fx * P.tx * rz3 * vJ[0][2] + 2.f * fy * P.ty * rz3 * vJ[1][2];
        // (3) depth channel (z, or 1/z with d(1/z)/dz = -1/z^2)
        vmc[2] += inv_depth ? -v_depth * rz2 : v_depth;

        // (4) p_c = R p + t
        const float (*R)[3] = cam.R;
        const float p[3] = {x, y, z};
        float vR[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) vR[i][j] = vmc[i] * p[j];
        float vp[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) vp[j] = R[0][j] * vmc[0] + R[1][j] * vmc[1] + R[2][j] * vmc[2];

        // (5) C = R cov R^T :  v_R += 2 vC R cov ; v_cov = R^T vC R   (vC, cov symmetric)
        float Rcov[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Rcov[i][j] = R[i][0] * P.cov[0][j] + R[i][1] * P.cov[1][j] + R[i][2] * P.cov[2][j];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                vR[i][j] += 2.f * (vC[i][0] * Rcov[0][j] + vC[i][1] * Rcov[1][j] + vC[i][2] * Rcov[2][j]);
        float vCR[3][3]; // vC * R
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) vCR[i][j] = vC[i][0] * R[0][j] + vC[i][1] * R[1][j] + vC[i][2] * R[2][j];
        float vcov[3][3]; // R^T * vC * R
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) vcov[i][j] = R[0][i] * vCR[0][j] + R[1][i] * vCR[1][j] + R[2][i] * vCR[2][j];

        // (6) cov = M M^T, M = Rq S :  v_M = 2 vcov M (vcov symmetric)
        float vM[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) vM[i][j] = 2.f * (vcov[i][0] * P.M[0][j] + vcov[i][1] * P.M[1][j] + vcov[i][2] * P.M[2][j]);
        float vs[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) vs[j] = P.Rq[0][j] * vM[0][j] + P.Rq[1][j] * vM[1][j] + P.Rq[2][j] * vM[2][j];
        float Gq[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Gq[i][j] = vM[i][j] * s[j];
        const float w = P.qn[0], qx = P.qn[1], qy = P.qn[2], qz = P.qn[3];
        float vqn[4];
        vqn[0] = 2.f * (qx * (Gq[2][1] - Gq[1][2]) + qy * (Gq[0][2] - Gq[2][0]) + qz * (Gq[1][0] - Gq[0][1]));
        vqn[1] = 2.f * (-2.f * qx * (Gq[1][1] + Gq[2][2]) + qy * (Gq[0][1] + Gq[1][0]) + qz * (Gq[0][2] + Gq[2][0]) + w * (Gq[2][1] - Gq[1][2]));
        vqn[2] = 2.f * (qx * (Gq[0][1] + Gq[1][0]) - 2.f * qy * (Gq[0][0] + Gq[2][2]) + qz * (Gq[1][2] + Gq[2][1]) + w * (Gq[0][2] - Gq[2][0]));
        vqn[3] = 2.f * (qx * (Gq[0][2] + Gq[2][0]) + qy * (Gq[1][2] + Gq[2][1]) - 2.f * qz * (Gq[0][0] + Gq[1][1]) + w * (Gq[1][0] - Gq[0][1]));
        const float dq = vqn[0] * w + vqn[1] * qx + vqn[2] * qy + vqn[3] * qz;
        const float vq[4] = {(vqn[0] - dq * w) * P.inv_qnorm, (vqn[1] - dq * qx) * P.inv_qnorm,
                             (vqn[2] - dq * qy) * P.inv_qnorm, (vqn[3] - dq * qz) * P.inv_qnorm};

        // (7) SH backward
        float vcp[3] = {0.f, 0.f, 0.f};
        if (color_mode == 0) {
            float dx = x - cam.campos[0], dy = y - cam.campos[1], dz = z - cam.campos[2];
            const float inorm = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
            dx *= inorm; dy *= inorm; dz *= inorm;
            constexpr int NB = (SH_DEG + 1) * (SH_DEG + 1);
            float b[16], bx[16], by[16], bz[16];
            sh_basis(SH_DEG, dx, dy, dz, b);
            sh_basis_grad(SH_DEG, dx, dy, dz, bx, by, bz);
            float acc[3] = {0.f, 0.f, 0.f};
            float cv[16][3];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const float* c = (sh_rest && k > 0) ? sh_rest + ((int64_t)g * (sh_K - 1) + (k - 1)) * 3
                                                    : colors_in + ((int64_t)g * (sh_rest ? 1 : sh_K) + k) * 3;
                cv[k][0] = c[0]; cv[k][1] = c[1]; cv[k][2] = c[2];
                acc[0] += b[k] * cv[k][0]; acc[1] += b[k] * cv[k][1]; acc[2] += b[k] * cv[k][2];
            }
            float vres[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) vres[ch] = (acc[ch] + 0.5f >= 0.f) ? v_col[ch] : 0.f;
            float vdn[3] = {0.f, 0.f, 0.f};
            float* o = v_colors ? v_colors + (int64_t)g * (sh_rest ? 1 : sh_K) * 3 : nullptr;
            float* orst = (v_colors && sh_rest) ? v_sh_rest + (int64_t)g * (sh_K - 1) * 3 : nullptr;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const float wsum = cv[k][0] * vres[0] + cv[k][1] * vres[1] + cv[k][2] * vres[2];
                vdn[0] += bx[k] * wsum; vdn[1] += by[k] * wsum; vdn[2] += bz[k] * wsum;
                if (FUSE_ADAM) { // applied by the coalesced phase at the end of the kernel
                } else {
                    float* dst = (orst && k > 0) ? orst + 3 * (k - 1) : (o ? o + 3 * k : nullptr);
                    if (dst) { dst[0] = b[k] * vres[0]; dst[1] = b[k] * vres[1]; dst[2] = b[k] * vres[2]; }
                }
            }
            if (FUSE_ADAM) { // hand (b_k, v_rgb) to the coalesced optimiser phase at the end of the kernel
#pragma unroll
                for (int k = 0; k < 16; ++k) adam_b[threadIdx.x][k] = k < NB ? b[k] : 0.f; // bands above the active degree: zero gradient
                adam_v[threadIdx.x][0] = vres[0]; adam_v[threadIdx.x][1] = vres[1]; adam_v[threadIdx.x][2] = vres[2];
            }
            else if (orst) { for (int i = (NB - 1) * 3; i < (sh_K - 1) * 3; ++i) orst[i] = 0.f; }
            else if (o) { for (int i = NB * 3; i < sh_K * 3; ++i) o[i] = 0.f; }
            const float dd = vdn[0] * dx + vdn[1] * dy + vdn[2] * dz;
            const float vd[3] = {(vdn[0] - dd * dx) * inorm, (vdn[1] - dd * dy) * inorm, (vdn[2] - dd * dz) * inorm};
#pragma unroll
            for (int i = 0; i < 3; ++i) { vp[i] += vd[i]; vcp[i] = -vd[i]; }
        } else if (color_mode == 1 && v_colors) {
            v_colors[3 * g] = v_col[0]; v_colors[3 * g + 1] = v_col[1]; v_colors[3 * g + 2] = v_col[2];
        }

        if (v_means) { v_means[3 * g] = vp[0]; v_means[3 * g + 1] = vp[1]; v_means[3 * g + 2] = vp[2]; }
        if (v_quats) reinterpret_cast<float4*>(v_quats)[g] = make_float4(vq[0], vq[1], vq[2], vq[3]);
        if (v_scales) { v_scales[3 * g] = vs[0]; v_scales[3 * g + 1] = vs[1]; v_scales[3 * g + 2] = vs[2]; }
        if (v_opacities) v_opacities[g] = v_op;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) cg[i * 3 + j] = vR[i][j];
            cg[9 + i] = vmc[i];
            cg[12 + i] = vcp[i];
        }
    }

    if (FUSE_ADAM) {
        // Sparse-Adam step of the block's SH coefficients, walked as the flat, 16 B-aligned slab they are in memory
        // (every access a full-width float4 per lane, like adam.hip).  A thread that walked its own Gaussian's row
        // (12 B accesses 180 B apart: 64 cache lines per wave instruction) made this 4x slower than the two separate
        // kernels; this way it is faster than them.  Gradient of element (row, k, ch) = b_k[row] * v_rgb[row][ch].
        __syncthreads();
        const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
        const int rows = (int)min((int64_t)blockDim.x, (int64_t)N - g0);
        const int RW = (sh_K - 1) * 3;
        const float lr_dc = opt.lr_dc[0], lr_rest = opt.lr_rest[0];
        const float omb1 = 1.0f - opt.b1, omb2 = 1.0f - opt.b2;
        auto step = [&](float& p, float& m, float& v, float gr, float lr) { // expression order of adam.hip:adam_elem
            m = opt.b1 * m + omb1 * gr;
            v = opt.b2 * v + omb2 * gr * gr;
            p += -lr * m / (sqrtf(v) + opt.eps);
        };
        {
            float* P = const_cast<float*>(sh_rest) + g0 * RW;
            float* M = opt.m_rest + g0 * RW;
            float* V = opt.v_rest + g0 * RW;
            const int total = rows * RW, n4 = total >> 2;
            for (int i = threadIdx.x; i < n4; i += blockDim.x) {
                const int e0 = 4 * i, r0 = e0 / RW, r1 = (e0 + 3) / RW;
                if (!adam_live[r0] && !adam_live[r1]) continue; // culled rows are not touched (not even read)
                float4 p4 = reinterpret_cast<float4*>(P)[i], m4 = reinterpret_cast<float4*>(M)[i], v4 = reinterpret_cast<float4*>(V)[i];
                float* pe = &p4.x; float* me = &m4.x; float* ve = &v4.x;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = e0 + j, row = e / RW, col = e - row * RW, k = col / 3 + 1, ch = col - (col / 3) * 3;
                    if (adam_live[row]) step(pe[j], me[j], ve[j], adam_b[row][k] * adam_v[row][ch], lr_rest);
                }
                reinterpret_cast<float4*>(P)[i] = p4; reinterpret_cast<float4*>(M)[i] = m4; reinterpret_cast<float4*>(V)[i] = v4;
            }
            for (int e = (n4 << 2) + threadIdx.x; e < total; e += blockDim.x) {
                const int row = e / RW, col = e - row * RW, k = col / 3 + 1, ch = col - (col / 3) * 3;
                if (adam_live[row]) step(P[e], M[e], V[e], adam_b[row][k] * adam_v[row][ch], lr_rest);
            }
        }
        {
            float* P = const_cast<float*>(colors_in) + g0 * 3;
            float* M = opt.m_dc + g0 * 3;
            float* V = opt.v_dc + g0 * 3;
            for (int e = threadIdx.x; e < rows * 3; e += blockDim.x) {
                const int row = e / 3, ch = e - row * 3;
                if (adam_live[row]) step(P[e], M[e], V[e], adam_b[row][0] * adam_v[row][ch], lr_dc);
            }
        }
    }
    if (cam_grad) { // uniform branch
        // Round 6: the 15 camera sums are carried in DOUBLE from the thread's own fp32 contribution to the accumulator (cam_grad is 16 doubles).
        // They are sums over every visible Gaussian of signed terms that cancel to a small fraction of their running magnitude: accumulated
        // with fp32 atomics over ~4 000 workgroups the pose gradient sat 3.7-4.7e-4 off the fp64 oracle at 1 M / 1080p (finding 40) -- the
        // largest deviation of any leaf, on the one that trains the camera.  64-bit shuffles + hardware fp64 atomics: 15 values per workgroup.
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            double d = (double)cg[i];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) d += __shfl_xor(d, o, 64);
            if (lane == 0) red[wv][i] = d;
        }
        __syncthreads();
        if (threadIdx.x < 15) {
            const double tot = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            if (tot != 0.0) unsafeAtomicAdd(reinterpret_cast<double*>(cam_grad) + threadIdx.x, tot);
        }
    }
}

// v_viewmat[4,4] += (v_R | v_t) + d(campos)/d(viewmat)^T v_campos, campos = -R^-1 t:
//   v_t' = -R^-T v_cp ;  v_R' = -R^-T (v_Ri) ... with Ri = R^-1:  d(Ri) = -Ri dR Ri,
//   campos = -Ri t  =>  v_Ri = -v_cp t^T ; v_R' = -Ri^T v_Ri Ri^T = Ri^T v_cp t^T Ri^T = (Ri^T v_cp)(Ri t)^T
// Leaves cam_grad (16 doubles) zeroed again: a caller can keep ONE accumulator per stream instead of clearing a fresh one every step.
__global__ void viewmat_grad_finalize_kernel(const float* __restrict__ viewmat, float* __restrict__ cam_grad,
                                             float* __restrict__ v_viewmat, const float* __restrict__ pose_r6, float* __restrict__ v_r6,
                                             float* __restrict__ v_t)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float R[3][3], t[3];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[i][j] = viewmat[i * 4 + j]; t[i] = viewmat[i * 4 + 3]; }
    float Ri[3][3];
    {
        const float c00 = R[1][1] * R[2][2] - R[1][2] * R[2][1], c01 = R[0][2] * R[2][1] - R[0][1] * R[2][2], c02 = R[0][1] * R[1][2] - R[0][2] * R[1][1];
        const float c10 = R[1][2] * R[2][0] - R[1][0] * R[2][2], c11 = R[0][0] * R[2][2] - R[0][2] * R[2][0], c12 = R[0][2] * R[1][0] - R[0][0] * R[1][2];
        const float c20 = R[1][0] * R[2][1] - R[1][1] * R[2][0], c21 = R[0][1] * R[2][0] - R[0][0] * R[2][1], c22 = R[0][0] * R[1][1] - R[0][1] * R[1][0];
        const float id = 1.0f / ((R[0][0] * c00 + R[0][1] * c10) + R[0][2] * c20);
        Ri[0][0] = c00 * id; Ri[0][1] = c01 * id; Ri[0][2] = c02 * id;
        Ri[1][0] = c10 * id; Ri[1][1] = c11 * id; Ri[1][2] = c12 * id;
        Ri[2][0] = c20 * id; Ri[2][1] = c21 * id; Ri[2][2] = c22 * id;
    }
    double* cgd = reinterpret_cast<double*>(cam_grad);   // 16 doubles (round 6): v_R (9) | v_t (3) | v_campos (3) | pad
    const float vcp[3] = {(float)cgd[12], (float)cgd[13], (float)cgd[14]};
    float a[3], bvec[3]; // a = Ri^T v_cp ; b = Ri t
    for (int i = 0; i < 3; ++i) {
        a[i] = Ri[0][i] * vcp[0] + Ri[1][i] * vcp[1] + Ri[2][i] * vcp[2];
        bvec[i] = Ri[i][0] * t[0] + Ri[i][1] * t[1] + Ri[i][2] * t[2];
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) v_viewmat[i * 4 + j] = (float)(cgd[i * 3 + j] + (double)a[i] * (double)bvec[j]);
        v_viewmat[i * 4 + 3] = (float)(cgd[9 + i] - (double)a[i]);
    }
    for (int j = 0; j < 4; ++j) v_viewmat[12 + j] = 0.f;
    for (int j = 0; j < 16; ++j) cgd[j] = 0.0;
    // the one-call step: Keyframe.get_Rt's backward on the matrix this thread has just written (else a launch of its own, adk_pose6d_bwd)
    if (pose_r6) pose6d_bwd_body(pose_r6, v_viewmat, v_r6, v_t);
}

} // namespace adk

#define ADK_DISPATCH_SH(DEG, ...)                                   \
    switch (DEG) {                                                  \
    case 0: { constexpr int SH_DEG = 0; __VA_ARGS__; } break;       \
    case 1: { constexpr int SH_DEG = 1; __VA_ARGS__; } break;       \
    case 2: { constexpr int SH_DEG = 2; __VA_ARGS__; } break;       \
    default: { constexpr int SH_DEG = 3; __VA_ARGS__; } break;      \
    }

extern "C" int adk_project_fwd(int N, const float* means, const float* quats, const float* scales,
                               const float* opacities, const float* colors_in, const float* sh_rest, int sh_K, int sh_degree,
                               int color_mode, const float* viewmat, const float* Kmat, int width, int height, float eps2d,
                               float near_plane, float far_plane, float radius_clip, int inv_depth, float* rec, int32_t* radii,
                               uint32_t* depth_keys, uint32_t* gauss_ids, int32_t* tiles_per_gauss, hipStream_t stream)
{
    return adk::project_fwd_launch(N, means, quats, scales, opacities, colors_in, sh_rest, sh_K, sh_degree, color_mode, viewmat, Kmat, width, height,
                                   eps2d, near_plane, far_plane, radius_clip, inv_depth, rec, radii, depth_keys, gauss_ids, tiles_per_gauss, nullptr,
                                   stream);
}

int adk::project_fwd_launch(int N, const float* means, const float* quats, const float* scales, const float* opacities, const float* colors_in,
                            const float* sh_rest, int sh_K, int sh_degree, int color_mode, const float* viewmat, const float* Kmat, int width,
                            int height, float eps2d, float near_plane, float far_plane, float radius_clip, int inv_depth, float* rec, int32_t* radii,
                            uint32_t* depth_keys, uint32_t* gauss_ids, int32_t* tiles_per_gauss, const ProjectMasks* masks_in, hipStream_t stream)
{
    if (N < 0 || width <= 0 || height <= 0) return ADK_EINVAL;
    if (N == 0) return 0;
    ProjectMasks masks = {nullptr, 0, nullptr, nullptr};
    if (masks_in) {
        masks = *masks_in;
        if (!masks.vis || (masks.gvis && (!masks.cls_id || masks.V <= 0))) return ADK_EINVAL;
    }
    if (!means || !quats || !scales || !opacities || !viewmat || !Kmat || !rec || !radii || !depth_keys || !gauss_ids || !tiles_per_gauss) return ADK_EINVAL;
    if (color_mode < 0 || color_mode > 2 || (color_mode != 2 && !colors_in)) return ADK_EINVAL;
    if (color_mode == 0 && (sh_degree < 0 || sh_degree > 3 || sh_K < (sh_degree + 1) * (sh_degree + 1))) return ADK_EINVAL;
    if (((uintptr_t)quats & 15) || ((uintptr_t)rec & 15) || (color_mode == 0 && !sh_rest && ((uintptr_t)colors_in & 15))) return ADK_EINVAL;
    if (sh_rest && (color_mode != 0 || sh_K < 2)) return ADK_EINVAL;
    const int tile_w = (width + 15) / 16, tile_h = (height + 15) / 16;
    const dim3 grid((unsigned)adk::ceil_div(N, 256)), block(256);
    const int deg = color_mode == 0 ? sh_degree : 0;
    ADK_DISPATCH_SH(deg, hipLaunchKernelGGL((adk::project_fwd_kernel<SH_DEG>), grid, block, 0, stream, N, means, quats,
                                            scales, opacities, colors_in, sh_rest, sh_K, color_mode, viewmat, Kmat, width, height,
                                            tile_w, tile_h, eps2d, near_plane, far_plane, radius_clip, inv_depth, rec, radii,
                                            depth_keys, gauss_ids, tiles_per_gauss, masks));
    ADK_RETURN_LAST_ERROR();
}

extern "C" int adk_lod_params_fwd(int N, const float* xyz, const float* opacity_raw, const float* scaling_raw,
                                  const float* rotation, const float* local_feat, const float* global_feat,
                                  const int64_t* cls_id, const float* d_max, int local_dim, int global_dim, int hidden_dim,
                                  const float* W1, const float* b1, const float* W2, const float* b2, const float* viewmat,
                                  float* opac_eff, float* scale_eff, float* quat_eff, uint8_t* selected, hipStream_t stream)
{
    if (N < 0) return ADK_EINVAL;
    if (local_dim != LOD_L || global_dim != LOD_G || hidden_dim != LOD_HID) return ADK_EUNSUPPORTED;
    if (N == 0) return 0;
    if (!xyz || !opacity_raw || !scaling_raw || !rotation || !local_feat || !global_feat || !cls_id || !d_max || !W1 || !b1 || !W2 || !b2 || !viewmat || !opac_eff || !scale_eff || !quat_eff || !selected) return ADK_EINVAL;
    if (((uintptr_t)rotation | (uintptr_t)local_feat | (uintptr_t)global_feat | (uintptr_t)quat_eff) & 15) return ADK_EINVAL;
    hipLaunchKernelGGL(adk::lod_params_fwd_kernel, dim3((unsigned)adk::ceil_div(N, 256)), dim3(256), 0, stream, N, xyz,
                       opacity_raw, scaling_raw, rotation, local_feat, global_feat, cls_id, d_max, W1, b1, W2, b2, viewmat,
                       opac_eff, scale_eff, quat_eff, selected);
    ADK_RETURN_LAST_ERROR();
}

int adk::lod_project_fwd_launch(int N, const float* xyz, const float* opacity_raw, const float* scaling_raw, const float* rotation,
                                const float* local_feat, const float* global_feat, const int64_t* cls_id, const float* d_max, const float* W1,
                                const float* b1, const float* W2, const float* b2, float* opac_eff, float* scale_eff, float* quat_eff,
                                uint8_t* selected, const float* f_dc, const float* f_rest, int sh_K, int sh_degree, const float* viewmat,
                                const float* Kmat, int width, int height, float eps2d, float near_plane, float far_plane, float radius_clip,
                                float* rec, int32_t* radii, uint32_t* depth_keys, uint32_t* gauss_ids, int32_t* tiles_per_gauss,
                                const ProjectMasks* masks_in, hipStream_t stream)
{
    if (N < 0 || width <= 0 || height <= 0) return ADK_EINVAL;
    if (N == 0) return 0;
    if (!xyz || !opacity_raw || !scaling_raw || !rotation || !local_feat || !global_feat || !cls_id || !d_max || !W1 || !b1 || !W2 || !b2 ||
        !opac_eff || !scale_eff || !quat_eff || !selected || !f_dc || !f_rest || !viewmat || !Kmat || !rec || !radii || !depth_keys ||
        !gauss_ids || !tiles_per_gauss) return ADK_EINVAL;
    if (sh_degree < 0 || sh_degree > 3 || sh_K < 2 || sh_K < (sh_degree + 1) * (sh_degree + 1)) return ADK_EINVAL;
    if (((uintptr_t)rotation | (uintptr_t)quat_eff | (uintptr_t)rec | (uintptr_t)local_feat | (uintptr_t)global_feat) & 15) return ADK_EINVAL;
    ProjectMasks masks = {nullptr, 0, nullptr, nullptr};
    if (masks_in) {
        masks = *masks_in;
        if (!masks.vis || (masks.gvis && (!masks.cls_id || masks.V <= 0))) return ADK_EINVAL;
    }
    const int tile_w = (width + 15) / 16, tile_h = (height + 15) / 16;
    const dim3 grid((unsigned)adk::ceil_div(N, 256)), block(256);
    ADK_DISPATCH_SH(sh_degree, hipLaunchKernelGGL((adk::lod_project_fwd_kernel<SH_DEG>), grid, block, 0, stream, N, xyz, opacity_raw, scaling_raw,
                                                  rotation, local_feat, global_feat, cls_id, d_max, W1, b1, W2, b2, opac_eff, scale_eff, quat_eff,
                                                  selected, f_dc, f_rest, sh_K, viewmat, Kmat, width, height, tile_w, tile_h, eps2d, near_plane,
                                                  far_plane, radius_clip, rec, radii, depth_keys, gauss_ids, tiles_per_gauss, masks));
    ADK_RETURN_LAST_ERROR();
}

int adk::project_bwd_launch(int N, const float* means, const float* quats, const float* scales,
                            const float* colors_in, const float* sh_rest, int sh_K, int sh_degree, int color_mode,
                            const float* viewmat, const float* Kmat, int width, int height, float eps2d,
                            float near_plane, float far_plane, int inv_depth, const int32_t* radii,
                            const float* v_rec, float* v_means, float* v_quats, float* v_scales,
                            float* v_opacities, float* v_colors, float* v_sh_rest, float* cam_grad,
                            float* v_viewmat, const adk::ColorAdam* opt, const float* pose_r6, float* v_r6, float* v_t, hipStream_t stream)
{
    if (N < 0 || width <= 0 || height <= 0) return ADK_EINVAL;
    if (!viewmat || !Kmat) return ADK_EINVAL;
    if ((v_viewmat != nullptr) != (cam_grad != nullptr)) return ADK_EINVAL;
    if (pose_r6 && (!v_viewmat || !v_r6 || !v_t)) return ADK_EINVAL;
    if (N > 0) {
        if (!means || !quats || !scales || !radii || !v_rec) return ADK_EINVAL;
        if (color_mode < 0 || color_mode > 2 || (color_mode == 0 && !colors_in)) return ADK_EINVAL;
        if (sh_rest && (color_mode != 0 || sh_K < 2 || (v_colors && !v_sh_rest))) return ADK_EINVAL;
        if (((uintptr_t)quats & 15) || ((uintptr_t)v_rec & 15) || (v_quats && ((uintptr_t)v_quats & 15))) return ADK_EINVAL;
        const dim3 grid((unsigned)adk::ceil_div(N, 256)), block(256);
        const int deg = color_mode == 0 ? sh_degree : 0;
        if (opt) {
            ADK_DISPATCH_SH(deg, hipLaunchKernelGGL((adk::project_bwd_kernel<SH_DEG, true>), grid, block, 0, stream, N, means,
                                                    quats, scales, colors_in, sh_rest, sh_K, color_mode, viewmat, Kmat, width, height,
                                                    eps2d, near_plane, far_plane, inv_depth, radii, v_rec, v_means, v_quats, v_scales,
                                                    v_opacities, nullptr, nullptr, cam_grad, *opt));
        } else {
            const adk::ColorAdam none = {};
            ADK_DISPATCH_SH(deg, hipLaunchKernelGGL((adk::project_bwd_kernel<SH_DEG, false>), grid, block, 0, stream, N, means,
                                                    quats, scales, colors_in, sh_rest, sh_K, color_mode, viewmat, Kmat, width, height,
                                                    eps2d, near_plane, far_plane, inv_depth, radii, v_rec, v_means, v_quats, v_scales,
                                                    v_opacities, v_colors, v_sh_rest, cam_grad, none));
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    if (v_viewmat) hipLaunchKernelGGL(adk::viewmat_grad_finalize_kernel, dim3(1), dim3(64), 0, stream, viewmat, cam_grad, v_viewmat, pose_r6, v_r6, v_t);
    ADK_RETURN_LAST_ERROR();
}

extern "C" int adk_project_bwd(int N, const float* means, const float* quats, const float* scales,
                               const float* colors_in, const float* sh_rest, int sh_K, int sh_degree, int color_mode,
                               const float* viewmat, const float* Kmat, int width, int height, float eps2d,
                               float near_plane, float far_plane, int inv_depth, const int32_t* radii,
                               const float* v_rec, float* v_means, float* v_quats, float* v_scales,
                               float* v_opacities, float* v_colors, float* v_sh_rest, float* cam_grad /*[16], zeroed*/,
                               float* v_viewmat /*[16] or NULL*/, hipStream_t stream)
{
    return adk::project_bwd_launch(N, means, quats, scales, colors_in, sh_rest, sh_K, sh_degree, color_mode, viewmat, Kmat, width, height,
                                   eps2d, near_plane, far_plane, inv_depth, radii, v_rec, v_means, v_quats, v_scales, v_opacities,
                                   v_colors, v_sh_rest, cam_grad, v_viewmat, nullptr, nullptr, nullptr, nullptr, stream);
}

// adk_project_bwd with the sparse-Adam step of the SH coefficients applied in the same pass: no v_colors / v_sh_rest
// are produced; f_dc [N,1,3] and f_rest [N,K-1,3] and their moments are updated IN PLACE for every Gaussian with
// radii > 0 -- exactly the rows adamUpdate(..., visible = radii > 0, ...) touches (optimizers.py:116-128), with the
// same arithmetic (bit-identical to adk_project_bwd followed by adk_adam_update).  lr_*: device scalars.
extern "C" int adk_project_bwd_adam(int N, const float* means, const float* quats, const float* scales,
                                    float* f_dc, float* f_rest, int sh_K, int sh_degree,
                                    const float* viewmat, const float* Kmat, int width, int height, float eps2d,
                                    float near_plane, float far_plane, int inv_depth, const int32_t* radii,
                                    const float* v_rec, float* v_means, float* v_quats, float* v_scales,
                                    float* v_opacities, float* cam_grad, float* v_viewmat,
                                    float* m_dc, float* v_dc, float* m_rest, float* v_rest, const float* lr_dc,
                                    const float* lr_rest, float b1, float b2, float eps, hipStream_t stream)
{
    return adk::project_bwd_adam_launch(N, means, quats, scales, f_dc, f_rest, sh_K, sh_degree, viewmat, Kmat, width, height, eps2d, near_plane,
                                        far_plane, inv_depth, radii, v_rec, v_means, v_quats, v_scales, v_opacities, cam_grad, v_viewmat, m_dc, v_dc,
                                        m_rest, v_rest, lr_dc, lr_rest, b1, b2, eps, nullptr, nullptr, nullptr, stream);
}

int adk::project_bwd_adam_launch(int N, const float* means, const float* quats, const float* scales, float* f_dc, float* f_rest, int sh_K,
                                 int sh_degree, const float* viewmat, const float* Kmat, int width, int height, float eps2d, float near_plane,
                                 float far_plane, int inv_depth, const int32_t* radii, const float* v_rec, float* v_means, float* v_quats,
                                 float* v_scales, float* v_opacities, float* cam_grad, float* v_viewmat, float* m_dc, float* v_dc, float* m_rest,
                                 float* v_rest, const float* lr_dc, const float* lr_rest, float b1, float b2, float eps, const float* pose_r6,
                                 float* v_r6, float* v_t, hipStream_t stream)
{
    if (!f_dc || !f_rest || sh_K < 2 || !m_dc || !v_dc || !m_rest || !v_rest || !lr_dc || !lr_rest) return ADK_EINVAL;
    if (sh_degree < 0 || sh_degree > 3 || sh_K < (sh_degree + 1) * (sh_degree + 1)) return ADK_EINVAL;
    if (sh_K > 16) return ADK_EUNSUPPORTED;
    if (((uintptr_t)f_rest | (uintptr_t)m_rest | (uintptr_t)v_rest) & 15) return ADK_EINVAL; // walked as float4
    adk::ColorAdam opt;
    opt.m_dc = m_dc; opt.v_dc = v_dc; opt.m_rest = m_rest; opt.v_rest = v_rest; opt.lr_dc = lr_dc; opt.lr_rest = lr_rest;
    opt.b1 = b1; opt.b2 = b2; opt.eps = eps;
    return adk::project_bwd_launch(N, means, quats, scales, f_dc, f_rest, sh_K, sh_degree, 0, viewmat, Kmat, width, height, eps2d,
                                   near_plane, far_plane, inv_depth, radii, v_rec, v_means, v_quats, v_scales, v_opacities,
                                   nullptr, nullptr, cam_grad, v_viewmat, &opt, pose_r6, v_r6, v_t, stream);
}
