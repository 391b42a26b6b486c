// The forward of the LoD / mlp_cov glue as device functions (h3dgsv3.py:626-662), shared by lod_params_fwd_kernel (lod_params.hip) and the
// fused LoD + projection forward of the one-call step (raster_project.hip).  `#pragma clang fp contract(fast)` pins the contraction mode of
// lod_params.hip inside every function, so that the file compiled with -ffp-contract=off (raster_project.hip: bit-exact integer decisions
// of the projection) computes the same opacity / scale / quaternion bits here as the stand-alone kernel does.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace adk {

#define LOD_G 16   // global_feat_dim (run.sh --global_feat_dim 16)
#define LOD_L 16   // local_feat_dim  (run.sh --local_feat_dim 16)
#define LOD_IN 32
#define LOD_HID 32
#define LOD_OUT 7
#define LOD_NW (LOD_HID * LOD_IN + LOD_HID + LOD_OUT * LOD_HID + LOD_OUT) // 1287 mlp parameters

struct CamCentre { float c[3]; };

__device__ __forceinline__ CamCentre cam_centre_of(const float* __restrict__ V) {
#pragma clang fp contract(fast)
    // -R^-1 t via the adjugate, same as raster_project.hip:load_cam
    float R[3][3], t[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[i][j] = V[i * 4 + j]; t[i] = V[i * 4 + 3]; }
    const float c00 = R[1][1] * R[2][2] - R[1][2] * R[2][1], c01 = R[0][2] * R[2][1] - R[0][1] * R[2][2], c02 = R[0][1] * R[1][2] - R[0][2] * R[1][1];
    const float c10 = R[1][2] * R[2][0] - R[1][0] * R[2][2], c11 = R[0][0] * R[2][2] - R[0][2] * R[2][0], c12 = R[0][2] * R[1][0] - R[0][0] * R[1][2];
    const float c20 = R[1][0] * R[2][1] - R[1][1] * R[2][0], c21 = R[0][1] * R[2][0] - R[0][0] * R[2][1], c22 = R[0][0] * R[1][1] - R[0][1] * R[1][0];
    const float id = 1.0f / ((R[0][0] * c00 + R[0][1] * c10) + R[0][2] * c20);
    CamCentre o;
    o.c[0] = -((c00 * t[0] + c01 * t[1] + c02 * t[2]) * id);
    o.c[1] = -((c10 * t[0] + c11 * t[1] + c12 * t[2]) * id);
    o.c[2] = -((c20 * t[0] + c21 * t[1] + c22 * t[2]) * id);
    return o;
}

__device__ __forceinline__ float sigmoidf(float x) {
#pragma clang fp contract(fast)
    return 1.0f / (1.0f + __expf(-x));
}

// x[32] = [global_feat[cls], local_feat[g]]
__device__ __forceinline__ void load_features(const float* __restrict__ global_feat, const float* __restrict__ local_feat,
                                              int64_t cls, int64_t g, float* x)
{
    const float4* gf = reinterpret_cast<const float4*>(global_feat + cls * LOD_G);
    const float4* lf = reinterpret_cast<const float4*>(local_feat + g * LOD_L);
#pragma unroll
    for (int i = 0; i < LOD_G / 4; ++i) { const float4 v = gf[i]; x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w; }
#pragma unroll
    for (int i = 0; i < LOD_L / 4; ++i) { const float4 v = lf[i]; x[LOD_G + 4 * i] = v.x; x[LOD_G + 4 * i + 1] = v.y; x[LOD_G + 4 * i + 2] = v.z; x[LOD_G + 4 * i + 3] = v.w; }
}

// h = relu(W1 x + b1) ; y = W2 h + b2.  Weight indices are wave-uniform => scalar loads.
__device__ __forceinline__ void mlp_forward(const float* __restrict__ W1, const float* __restrict__ b1,
                                            const float* __restrict__ W2, const float* __restrict__ b2,
                                            const float* x, float* h, float* y)
{
#pragma clang fp contract(fast)
#pragma unroll
    for (int i = 0; i < LOD_HID; ++i) {
        float a = b1[i];
#pragma unroll
        for (int j = 0; j < LOD_IN; ++j) a += W1[i * LOD_IN + j] * x[j];
        h[i] = fmaxf(a, 0.f);
    }
#pragma unroll
    for (int o = 0; o < LOD_OUT; ++o) {
        float a = b2[o];
#pragma unroll
        for (int i = 0; i < LOD_HID; ++i) a += W2[o * LOD_HID + i] * h[i];
        y[o] = a;
    }
}

struct LodGeom { float dist, alpha_ratio, inv_dmax; bool selected, fading; float dir[3]; };

__device__ __forceinline__ LodGeom lod_geometry_of(float px, float py, float pz, float dm, const CamCentre& cc);
__device__ __forceinline__ LodGeom lod_geometry(const float* __restrict__ xyz, const float* __restrict__ d_max, int64_t g, const CamCentre& cc) {
    return lod_geometry_of(xyz[3 * g], xyz[3 * g + 1], xyz[3 * g + 2], d_max[g], cc);
}
__device__ __forceinline__ LodGeom lod_geometry_of(float px, float py, float pz, float dm, const CamCentre& cc) {
#pragma clang fp contract(fast)
    LodGeom L;
    const float dx = px - cc.c[0], dy = py - cc.c[1], dz = pz - cc.c[2];
    L.dist = sqrtf(dx * dx + dy * dy + dz * dz);
    L.selected = L.dist < 2.f * dm;
    L.fading = (L.dist > dm) && (L.dist < 2.f * dm);
    L.inv_dmax = 1.0f / dm;
    L.alpha_ratio = L.fading ? (2.f * dm - L.dist) * L.inv_dmax : 1.0f;
    const float id = L.dist > 0.f ? 1.0f / L.dist : 0.f;
    L.dir[0] = dx * id; L.dir[1] = dy * id; L.dir[2] = dz * id;
    return L;
}

// One Gaussian's effective opacity, scale and (un-normalised) quaternion; unselected Gaussians get opacity 0 (culled by the projection).
// The stand-alone kernel and the fused LoD + projection kernel that inline this live in ONE translation unit (raster_project.hip): compiled in
// two files (one with -ffp-contract=off) the same source came out with different FMA choices although every function here pins
// contract(fast) -- the fade factor of some Gaussians moved by an ulp between the two (tools/lab/native_vs_stage_diag.py).  The fused kernel
// additionally hides its projection phase's inputs from the optimiser, so that no sub-expression is shared across the two phases.
struct LodOut { float opac; float scale[3]; float4 quat; float selected; };
__device__ __forceinline__ LodOut lod_forward_one(int64_t g, const float* __restrict__ xyz, const float* __restrict__ opacity_raw,
                                                  const float* __restrict__ scaling_raw, const float* __restrict__ rotation,
                                                  const float* __restrict__ local_feat, const float* __restrict__ global_feat,
                                                  const int64_t* __restrict__ cls_id, const float* __restrict__ d_max,
                                                  const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                                                  const float* __restrict__ b2, const float* __restrict__ viewmat)
{
#pragma clang fp contract(fast)
    LodOut o;
    const CamCentre cc = cam_centre_of(viewmat);
    const LodGeom L = lod_geometry(xyz, d_max, g, cc);
    const float fade = L.alpha_ratio;
    o.selected = L.selected ? 1.f : 0.f;
    if (!L.selected) { // never rendered: opacity 0 is culled by the projection (opacity < 1/255)
        o.opac = 0.f; o.scale[0] = o.scale[1] = o.scale[2] = 1.f; o.quat = make_float4(1.f, 0.f, 0.f, 0.f);
        return o;
    }
    float x[LOD_IN], h[LOD_HID], y[LOD_OUT];
    load_features(global_feat, local_feat, cls_id[g], g, x);
    mlp_forward(W1, b1, W2, b2, x, h, y);
    o.opac = sigmoidf(opacity_raw[g]) * fade;
#pragma unroll
    for (int k = 0; k < 3; ++k) o.scale[k] = __expf(scaling_raw[3 * g + k]) * sigmoidf(y[k]);
    const float4 q = reinterpret_cast<const float4*>(rotation)[g];
    // F.normalize(rotation * scale_rot[:,3:]) -- the projection normalises again (idempotent), so the
    // un-normalised product is handed over and the normalisation Jacobian lives in one place.
    o.quat = make_float4(q.x * y[3], q.y * y[4], q.z * y[5], q.w * y[6]);
    return o;
}

} // namespace adk
