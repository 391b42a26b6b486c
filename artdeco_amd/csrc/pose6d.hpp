// Keyframe.get_Rt's 6D rotation (Reconstruct/scene/keyframe.py:150-154, utils.py:223-229) and its backward as device functions: shared by the
// stand-alone kernels (photometric.hip) and by the camera-gradient finalisation of the projection backward (raster_project.hip), which runs
// the pose backward in the same single-thread launch when the one-call step asks for it.  `#pragma clang fp contract(fast)` pins the
// contraction mode of photometric.hip here, so that both users compute the same bits whatever their file's -ffp-contract says.
#pragma once
#include <hip/hip_runtime.h>

namespace adk {

struct Pose6 { float b1[3], b2[3], b3[3], a2[3], n1, nu, d; };

__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
#pragma clang fp contract(fast)
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ float dot3(const float* a, const float* b) {
#pragma clang fp contract(fast)
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

__device__ __forceinline__ Pose6 pose6_of(const float* __restrict__ r6 /* [3,2] row-major */) {
#pragma clang fp contract(fast)
    Pose6 q;
    float a1[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { a1[i] = r6[2 * i]; q.a2[i] = r6[2 * i + 1]; }
    q.n1 = sqrtf(dot3(a1, a1));
#pragma unroll
    for (int i = 0; i < 3; ++i) q.b1[i] = a1[i] / q.n1;
    q.d = dot3(q.b1, q.a2);
    float u[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) u[i] = q.a2[i] - q.d * q.b1[i];
    q.nu = sqrtf(dot3(u, u));
#pragma unroll
    for (int i = 0; i < 3; ++i) q.b2[i] = u[i] / q.nu;
    cross3(q.b1, q.b2, q.b3);
    return q;
}

// v_Rt [4,4] -> v_r6 [3,2], v_t [3]
__device__ __forceinline__ void pose6d_bwd_body(const float* __restrict__ r6, const float* v_Rt, float* __restrict__ v_r6, float* __restrict__ v_t) {
#pragma clang fp contract(fast)
    const Pose6 q = pose6_of(r6);
    float g1[3], g2[3], g3[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { g1[i] = v_Rt[4 * i]; g2[i] = v_Rt[4 * i + 1]; g3[i] = v_Rt[4 * i + 2]; v_t[i] = v_Rt[4 * i + 3]; }
    // b3 = b1 x b2
    float vb1[3], vb2[3], tmp[3];
    cross3(q.b2, g3, tmp);
#pragma unroll
    for (int i = 0; i < 3; ++i) vb1[i] = g1[i] + tmp[i];
    cross3(g3, q.b1, tmp);
#pragma unroll
    for (int i = 0; i < 3; ++i) vb2[i] = g2[i] + tmp[i];
    // b2 = u / |u|
    float vu[3];
    const float s2 = dot3(q.b2, vb2);
#pragma unroll
    for (int i = 0; i < 3; ++i) vu[i] = (vb2[i] - q.b2[i] * s2) / q.nu;
    // u = a2 - (b1 . a2) b1
    const float s1 = dot3(q.b1, vu);
    float va2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { va2[i] = vu[i] - q.b1[i] * s1; vb1[i] -= q.d * vu[i] + s1 * q.a2[i]; }
    // b1 = a1 / |a1|
    const float s0 = dot3(q.b1, vb1);
#pragma unroll
    for (int i = 0; i < 3; ++i) { v_r6[2 * i] = (vb1[i] - q.b1[i] * s0) / q.n1; v_r6[2 * i + 1] = va2[i]; }
}

} // namespace adk
