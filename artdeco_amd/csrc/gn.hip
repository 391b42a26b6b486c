// Sim(3) Gauss-Newton global optimiser (points / rays / calibrated-projection factors) for gfx950
// (SURVEY.md 8 f-3).
//
// Replaces gauss_newton_points / gauss_newton_rays / gauss_newton_calib of the mast3r_slam_backends extension:
// VSLAM/backend/src/gn.cpp:3-82 (bindings), gn_kernels.cu:455-811 (points), :813-1215 (rays), :1218-1637 (calib),
// called from VSLAM/mast3r_slam/global_opt.py:158-173 and :208-228.  Same problem, same residuals, Jacobians,
// Huber weights, fixed first pose, left-multiplicative Sim(3) retraction (expSim3 with the reference's series
// branches, gn_kernels.cu:322-413) and the same "stop when |dx| < delta_thresh" rule.
//
// What is different, and why (MI355X-first, not a translation):
//   * The reference builds, per factor, the full 14x14 Hessian of [pose i, pose j] per thread (105 + 14
//     accumulators) and tree-reduces 119 values through shared memory.  But every Jacobian row is
//     J_j = M_i J0 and J_i = -J_j, where J0 is the 7-vector written down in the reference and M_i the constant
//     7x7 inverse-adjoint of pose i (apply_Sim3_adj_inv is linear).  So only S = sum w J0 J0^T (28 numbers) and
//     v = sum w r J0 (7 numbers) are accumulated per point -- 4x less arithmetic and registers -- and
//     H_jj = M S M^T = H_ii = -H_ij, g_j = M v = -g_i are formed once per factor, in fp64.
//   * One factor is split over several workgroups (a factor has 196 608 points at 512x384 but a graph may have
//     only a few dozen factors: one workgroup per factor would light 20 of 256 CUs); fp32 partials are summed in
//     a fixed order in fp64.
//   * The normal equations (7 (P-1) unknowns) are assembled, factorised (dense fp64 blocked Cholesky over 32x32 tiles,
//     many workgroups, with the right-hand side carried as an extra block row) and solved ON THE DEVICE, followed by
//     the retraction and the step-norm test, which sets a device flag that turns the remaining pre-enqueued
//     iterations into no-ops.
//     The reference copies every factor block to the host, runs Eigen's sparse LLT there and copies dx back,
//     then reads |dx| on the host: 3 stream drains per iteration, 10 iterations per call.  Here: none.
#include "adk_common.hpp"

namespace adk {

#define GN_NS 28         // unique entries of the symmetric 7x7
#define GN_NACC 35       // + 7 gradient entries
#define GN_MAX_CHUNKS 64
#define GN_SOLVE_THREADS 1024

struct Sim3 { float t[3], q[4], s; };

__device__ __forceinline__ void quat_mul(const float* a, const float* b, float* o) { // gn_kernels.cu:178-184 (xyzw)
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

template <typename T>
__device__ __forceinline__ void rot(const T* q, const T* X, T* Y) { // actSO3, gn_kernels.cu:196-206
    const T u0 = T(2) * (q[1] * X[2] - q[2] * X[1]);
    const T u1 = T(2) * (q[2] * X[0] - q[0] * X[2]);
    const T u2 = T(2) * (q[0] * X[1] - q[1] * X[0]);
    const T y0 = X[0] + q[3] * u0 + (q[1] * u2 - q[2] * u1);
    const T y1 = X[1] + q[3] * u1 + (q[2] * u0 - q[0] * u2);
    const T y2 = X[2] + q[3] * u2 + (q[0] * u1 - q[1] * u0);
    Y[0] = y0; Y[1] = y1; Y[2] = y2;
}

__device__ __forceinline__ Sim3 load_pose(const float* __restrict__ Twc, int64_t p) {
    Sim3 T;
    const float* r = Twc + 8 * p;
    T.t[0] = r[0]; T.t[1] = r[1]; T.t[2] = r[2];
    T.q[0] = r[3]; T.q[1] = r[4]; T.q[2] = r[5]; T.q[3] = r[6];
    T.s = r[7];
    return T;
}

// T_i^-1 T_j, gn_kernels.cu:248-268
__device__ __forceinline__ Sim3 rel_sim3(const Sim3& Ti, const Sim3& Tj) {
    Sim3 R;
    const float si_inv = 1.0f / Ti.s;
    R.s = si_inv * Tj.s;
    const float qi_inv[4] = {-Ti.q[0], -Ti.q[1], -Ti.q[2], Ti.q[3]};
    quat_mul(qi_inv, Tj.q, R.q);
    float d[3] = {Tj.t[0] - Ti.t[0], Tj.t[1] - Ti.t[1], Tj.t[2] - Ti.t[2]};
    rot(qi_inv, d, d);
    R.t[0] = d[0] * si_inv; R.t[1] = d[1] * si_inv; R.t[2] = d[2] * si_inv;
    return R;
}

__device__ __forceinline__ float huber(float r) { // gn_kernels.cu:171-174
    const float a = fabsf(r);
    return a < 1.345f ? 1.0f : 1.345f / a;
}

// S += w J J^T (lower triangle, row-major packed), v += w e J
__device__ __forceinline__ void add_row(float* S, float* v, const float* J, float w, float e) {
    int l = 0;
#pragma unroll
    for (int n = 0; n < 7; ++n) {
        const float wj = w * J[n];
#pragma unroll
        for (int m = 0; m <= n; ++m) S[l++] += wj * J[m];
        v[n] += wj * e;
    }
}

struct GnArgs {
    const float* Twc; const float* Xs; const float* Cs; const float* K;
    const int64_t* ii; const int64_t* jj; const int64_t* idx; const uint8_t* valid; const float* Q;
    int num_points, chunk, num_chunks;
    int height, width, pixel_border;
    float z_eps, sigma_a, sigma_b, C_thresh, Q_thresh;
    float* partials;          // [E][num_chunks][GN_NACC]
    const int* done;
};

// KIND 0: 3-D point residual, 1: ray + distance residual, 2: pixel + log-depth residual
template <int KIND>
__global__ __launch_bounds__(256) void gn_accumulate_kernel(GnArgs a)
{
    __shared__ float red[4][GN_NACC];
    if (*a.done) return;
    const int e = blockIdx.x, ch = blockIdx.y;
    const int64_t ix = a.ii[e], jx = a.jj[e];
    const Sim3 Ti = load_pose(a.Twc, ix), Tj = load_pose(a.Twc, jx);
    const Sim3 Tij = rel_sim3(Ti, Tj);
    const int n = a.num_points;
    float fx = 0.f, fy = 0.f, cx = 0.f, cy = 0.f;
    if (KIND == 2) { fx = a.K[0]; fy = a.K[4]; cx = a.K[2]; cy = a.K[5]; }
    const float sa_inv = 1.0f / a.sigma_a, sb_inv = (KIND == 0) ? 0.f : 1.0f / a.sigma_b;

    float S[GN_NS], v[7];
#pragma unroll
    for (int l = 0; l < GN_NS; ++l) S[l] = 0.f;
#pragma unroll
    for (int l = 0; l < 7; ++l) v[l] = 0.f;

    const float* Xi_base = a.Xs + (int64_t)ix * n * 3;
    const float* Xj_base = a.Xs + (int64_t)jx * n * 3;
    const float* Ci_base = a.Cs + (int64_t)ix * n;
    const float* Cj_base = a.Cs + (int64_t)jx * n;
    const int k_end = min(n, (ch + 1) * a.chunk);
#pragma unroll 4
    for (int k = ch * a.chunk + (int)threadIdx.x; k < k_end; k += 256) {
        const int64_t ek = (int64_t)e * n + k;
        const bool vm = a.valid[ek] != 0;
        const int64_t ind = vm ? a.idx[ek] : 0;
        const float Xi[3] = {Xi_base[3 * ind], Xi_base[3 * ind + 1], Xi_base[3 * ind + 2]};
        const float Xj[3] = {Xj_base[3 * (int64_t)k], Xj_base[3 * (int64_t)k + 1], Xj_base[3 * (int64_t)k + 2]};
        float P[3];
        rot(Tij.q, Xj, P);
        P[0] = P[0] * Tij.s + Tij.t[0]; P[1] = P[1] * Tij.s + Tij.t[1]; P[2] = P[2] * Tij.s + Tij.t[2];
        const float q = a.Q[ek], ci = Ci_base[ind], cj = Cj_base[k];
        bool valid = vm && (q > a.Q_thresh) && (ci > a.C_thresh) && (cj > a.C_thresh);
        float J[7];
        if (KIND == 0) {                                   // gn_kernels.cu:560-680
            const float sw = valid ? sa_inv * sqrtf(q) : 0.f, wc = sw * sw;
            const float e0 = P[0] - Xi[0], e1 = P[1] - Xi[1], e2 = P[2] - Xi[2];
            J[0] = 1.f; J[1] = 0.f; J[2] = 0.f; J[3] = 0.f; J[4] = P[2]; J[5] = -P[1]; J[6] = P[0];
            add_row(S, v, J, huber(sw * e0) * wc, e0);
            J[0] = 0.f; J[1] = 1.f; J[2] = 0.f; J[3] = -P[2]; J[4] = 0.f; J[5] = P[0]; J[6] = P[1];
            add_row(S, v, J, huber(sw * e1) * wc, e1);
            J[0] = 0.f; J[1] = 0.f; J[2] = 1.f; J[3] = P[1]; J[4] = -P[0]; J[5] = 0.f; J[6] = P[2];
            add_row(S, v, J, huber(sw * e2) * wc, e2);
        } else if (KIND == 1) {                            // gn_kernels.cu:920-1090
            const float n2i = Xi[0] * Xi[0] + Xi[1] * Xi[1] + Xi[2] * Xi[2];
            const float n1i = sqrtf(n2i), n1i_inv = 1.0f / n1i;
            const float n2j = P[0] * P[0] + P[1] * P[1] + P[2] * P[2];
            const float n1j = sqrtf(n2j), n1j_inv = 1.0f / n1j;
            const float r[3] = {n1j_inv * P[0], n1j_inv * P[1], n1j_inv * P[2]};
            const float e0 = r[0] - n1i_inv * Xi[0], e1 = r[1] - n1i_inv * Xi[1], e2 = r[2] - n1i_inv * Xi[2];
            const float e3 = n1j - n1i;
            const float swr = valid ? sa_inv * sqrtf(q) : 0.f, swd = valid ? sb_inv * sqrtf(q) : 0.f;
            const float wr = swr * swr, wd = swd * swd;
            const float n3 = n1j_inv / n2j;
            const float dxx = n1j_inv - P[0] * P[0] * n3, dyy = n1j_inv - P[1] * P[1] * n3, dzz = n1j_inv - P[2] * P[2] * n3;
            const float dxy = -P[0] * P[1] * n3, dxz = -P[0] * P[2] * n3, dyz = -P[1] * P[2] * n3;
            J[0] = dxx; J[1] = dxy; J[2] = dxz; J[3] = 0.f; J[4] = r[2]; J[5] = -r[1]; J[6] = 0.f;
            add_row(S, v, J, huber(swr * e0) * wr, e0);
            J[0] = dxy; J[1] = dyy; J[2] = dyz; J[3] = -r[2]; J[4] = 0.f; J[5] = r[0]; J[6] = 0.f;
            add_row(S, v, J, huber(swr * e1) * wr, e1);
            J[0] = dxz; J[1] = dyz; J[2] = dzz; J[3] = r[1]; J[4] = -r[0]; J[5] = 0.f; J[6] = 0.f;
            add_row(S, v, J, huber(swr * e2) * wr, e2);
            J[0] = r[0]; J[1] = r[1]; J[2] = r[2]; J[3] = 0.f; J[4] = 0.f; J[5] = 0.f; J[6] = n1j;
            add_row(S, v, J, huber(swd * e3) * wd, e3);
        } else {                                           // gn_kernels.cu:1346-1480
            const int u_t = (int)(ind % a.width), v_t = (int)(ind / a.width);
            const bool vz = (P[2] > a.z_eps) && (Xi[2] > a.z_eps);
            const float zinv = vz ? 1.0f / P[2] : 0.f;
            const float zj_log = vz ? logf(P[2]) : 0.f, zi_log = vz ? logf(Xi[2]) : 0.f;
            const float xz = P[0] * zinv, yz = P[1] * zinv;
            const float u = fx * xz + cx, vv = fy * yz + cy;
            const bool vu = (u > (float)a.pixel_border) && (u < (float)(a.width - 1 - a.pixel_border));
            const bool vvv = (vv > (float)a.pixel_border) && (vv < (float)(a.height - 1 - a.pixel_border));
            valid = valid && vu && vvv && vz;
            const float e0 = u - (float)u_t, e1 = vv - (float)v_t, e2 = zj_log - zi_log;
            const float swp = valid ? sa_inv * sqrtf(q) : 0.f, swd = valid ? sb_inv * sqrtf(q) : 0.f;
            const float wp = swp * swp, wd = swd * swd;
            J[0] = fx * zinv; J[1] = 0.f; J[2] = -fx * xz * zinv; J[3] = -fx * xz * yz; J[4] = fx * (1.f + xz * xz); J[5] = -fx * yz; J[6] = 0.f;
            add_row(S, v, J, huber(swp * e0) * wp, e0);
            J[0] = 0.f; J[1] = fy * zinv; J[2] = -fy * yz * zinv; J[3] = -fy * (1.f + yz * yz); J[4] = fy * xz * yz; J[5] = fy * xz; J[6] = 0.f;
            add_row(S, v, J, huber(swp * e1) * wp, e1);
            J[0] = 0.f; J[1] = 0.f; J[2] = zinv; J[3] = yz; J[4] = -xz; J[5] = 0.f; J[6] = 1.f;
            add_row(S, v, J, huber(swd * e2) * wd, e2);
        }
    }
    // workgroup reduction: DPP within the wave, LDS across the four waves, fixed order
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int l = 0; l < GN_NS; ++l) { const float t = wave_sum_to_lane63(S[l]); if (lane == 63) red[wv][l] = t; }
#pragma unroll
    for (int l = 0; l < 7; ++l) { const float t = wave_sum_to_lane63(v[l]); if (lane == 63) red[wv][GN_NS + l] = t; }
    __syncthreads();
    if (threadIdx.x < GN_NACC)
        a.partials[((int64_t)e * a.num_chunks + ch) * GN_NACC + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- Sim(3) exponential and retraction, gn_kernels.cu:302-413 (series branches and the "C - ..." form kept) ----
#define GN_EPS 1e-6f
__device__ __forceinline__ void cross_inplace(const float* a, float* b) {
    const float x0 = a[1] * b[2] - a[2] * b[1], x1 = a[2] * b[0] - a[0] * b[2], x2 = a[0] * b[1] - a[1] * b[0];
    b[0] = x0; b[1] = x1; b[2] = x2;
}

__device__ inline void exp_sim3(const float* xi, float* t, float* q, float* s) {
    float tau[3] = {xi[0], xi[1], xi[2]};
    const float phi[3] = {xi[3], xi[4], xi[5]};
    const float sigma = xi[6];
    const float scale = expf(sigma);
    const float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
    {   // expSO3
        float imag, real;
        if (theta_sq < GN_EPS) {
            const float p4 = theta_sq * theta_sq;
            imag = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * p4;
            real = 1.0f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * p4;
        } else {
            const float theta = sqrtf(theta_sq);
            imag = sinf(0.5f * theta) / theta;
            real = cosf(0.5f * theta);
        }
        q[0] = imag * phi[0]; q[1] = imag * phi[1]; q[2] = imag * phi[2]; q[3] = real;
    }
    s[0] = scale;
    const float theta = sqrtf(theta_sq);
    float A, B, C;
    if (fabsf(sigma) < GN_EPS) {
        C = 1.0f;
        if (fabsf(theta) < GN_EPS) { A = 0.5f; B = 1.0f / 6.0f; }
        else { A = (1.0f - cosf(theta)) / theta_sq; B = (theta - sinf(theta)) / (theta_sq * theta); }
    } else {
        C = (scale - 1.0f) / sigma;
        if (fabsf(theta) < GN_EPS) {
            const float s2 = sigma * sigma;
            A = ((sigma - 1.0f) * scale + 1.0f) / s2;
            B = (scale * 0.5f * s2 + scale - 1.0f - sigma * scale) / (s2 * sigma);
        } else {
            const float a = scale * sinf(theta), b = scale * cosf(theta), c = theta_sq + sigma * sigma;
            A = (a * sigma + (1.0f - b) * theta) / (theta * c);
            B = (C - ((b - 1.0f) * sigma + a * theta) / c) / theta_sq;
        }
    }
    t[0] = C * tau[0]; t[1] = C * tau[1]; t[2] = C * tau[2];
    cross_inplace(phi, tau);
    t[0] += A * tau[0]; t[1] += A * tau[1]; t[2] += A * tau[2];
    cross_inplace(phi, tau);
    t[0] += B * tau[0]; t[1] += B * tau[1]; t[2] += B * tau[2];
}

__device__ inline void retract_pose(float* __restrict__ pose, const float* xi) { // retrSim3 + pose_retr_kernel
    float dt[3], dq[4], ds[1];
    exp_sim3(xi, dt, dq, ds);
    const float t[3] = {pose[0], pose[1], pose[2]}, q[4] = {pose[3], pose[4], pose[5], pose[6]};
    float q1[4], t1[3];
    quat_mul(dq, q, q1);
    rot(dq, t, t1);
    pose[0] = t1[0] * ds[0] + dt[0]; pose[1] = t1[1] * ds[0] + dt[1]; pose[2] = t1[2] * ds[0] + dt[2];
    pose[3] = q1[0]; pose[4] = q1[1]; pose[5] = q1[2]; pose[6] = q1[3];
    pose[7] = ds[0] * pose[7];
}

// 7x7 M of Y = apply_Sim3_adj_inv(t, q, s; X), gn_kernels.cu:273-299, in fp64
__device__ inline void adj_inv_matrix(const Sim3& T, double M[7][7]) {
    const double q[4] = {T.q[0], T.q[1], T.q[2], T.q[3]}, t[3] = {T.t[0], T.t[1], T.t[2]};
    const double s_inv = 1.0 / (double)T.s;
    for (int c = 0; c < 7; ++c) {
        double X[7] = {0, 0, 0, 0, 0, 0, 0};
        X[c] = 1.0;
        double Ra[3], Rb[3];
        rot(q, &X[0], Ra);
        rot(q, &X[3], Rb);
        M[0][c] = s_inv * Ra[0]; M[1][c] = s_inv * Ra[1]; M[2][c] = s_inv * Ra[2];
        M[3][c] = Rb[0] + s_inv * (t[1] * Ra[2] - t[2] * Ra[1]);
        M[4][c] = Rb[1] + s_inv * (t[2] * Ra[0] - t[0] * Ra[2]);
        M[5][c] = Rb[2] + s_inv * (t[0] * Ra[1] - t[1] * Ra[0]);
        M[6][c] = X[6] + s_inv * (t[0] * Ra[0] + t[1] * Ra[1] + t[2] * Ra[2]);
    }
}

// ---- normal equations: assembly + blocked fp64 Cholesky + solve ------------------------------------------------------
// A is (Dp + GN_TB) x Dp, row-major, lower triangle used: Dp = D rounded up to the tile size (the padding carries an
// identity diagonal), and ONE EXTRA BLOCK ROW whose first row is the right-hand side: running it through the panel and
// update kernels like any other block row performs the forward substitution for free (it ends up holding y = L^-1 b).
// Right-looking blocked factorisation, tile GN_TB = 32, two launches per block column:
//   gn_panel_kernel   every workgroup factors the 32x32 diagonal tile in LDS (11 k flops: cheaper than a dependency),
//                     workgroup 0 stores it, workgroup r > 0 solves its block row X L^T = A[r, kb];
//   gn_update_kernel  one workgroup per tile (i >= j > kb): A[i,j] -= A[i,kb] A[j,kb]^T from LDS.
// The column-by-column single-workgroup version this replaces took 0.26 ms at D = 105 and 13.7 ms at D = 665 (one
// CU, three barriers per column, trailing update straight from L2).
#define GN_TB 32

struct GnSys { double* A; int D, Dp; int* done; int* fail; };

// Zero-fill as a kernel: a hipMemsetAsync node captured into a hipGraph is not ordered before the kernel nodes that
// follow it (observed on ROCm 7.2 / MI355X), and this entry point should be capturable.
__global__ __launch_bounds__(256) void gn_clear_kernel(uint4* __restrict__ p, int64_t n16)
{
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

__global__ __launch_bounds__(1024) void gn_assemble_kernel(
    const float* __restrict__ Twc, int num_fix, const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, int num_edges,
    int num_chunks, const float* __restrict__ partials, GnSys sys, float* __restrict__ Hs_dbg /* [4][E][7][7] or null */,
    float* __restrict__ gs_dbg /* [2][E][7] or null */)
{
    if (*sys.done) return;
    const int D = sys.D, LD = sys.Dp;
    double* __restrict__ A = sys.A; // zero-filled by the host-side memset
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0) *sys.fail = 0;
    if (e < sys.Dp - D) A[(int64_t)(D + e) * LD + D + e] = 1.0; // identity on the padding
    if (e >= num_edges) return;
    // factor blocks: H_jj = M S M^T, g_j = M v; H_ii = H_jj, H_ij = H_ji = -H_jj, g_i = -g_j
    double S[7][7], v[7];
    {
        double acc[GN_NACC];
        for (int l = 0; l < GN_NACC; ++l) acc[l] = 0.0;
        const float* p = partials + (int64_t)e * num_chunks * GN_NACC;
        for (int c = 0; c < num_chunks; ++c)
            for (int l = 0; l < GN_NACC; ++l) acc[l] += (double)p[c * GN_NACC + l];
        int l = 0;
        for (int n = 0; n < 7; ++n) for (int m = 0; m <= n; ++m) { S[n][m] = acc[l]; S[m][n] = acc[l]; ++l; }
        for (int n = 0; n < 7; ++n) v[n] = acc[GN_NS + n];
    }
    const int64_t ix = ii[e], jx = jj[e];
    double M[7][7];
    adj_inv_matrix(load_pose(Twc, ix), M);
    double MS[7][7], H[7][7], g[7];
    for (int r = 0; r < 7; ++r) for (int c = 0; c < 7; ++c) { double t = 0; for (int k = 0; k < 7; ++k) t += M[r][k] * S[k][c]; MS[r][c] = t; }
    for (int r = 0; r < 7; ++r) for (int c = 0; c < 7; ++c) { double t = 0; for (int k = 0; k < 7; ++k) t += MS[r][k] * M[c][k]; H[r][c] = t; }
    for (int r = 0; r < 7; ++r) { double t = 0; for (int k = 0; k < 7; ++k) t += M[r][k] * v[k]; g[r] = t; }
    if (Hs_dbg) {
        for (int r = 0; r < 7; ++r) for (int c = 0; c < 7; ++c) {
            const float h = (float)H[r][c];
            Hs_dbg[((0 * (int64_t)num_edges + e) * 7 + r) * 7 + c] = h;
            Hs_dbg[((1 * (int64_t)num_edges + e) * 7 + r) * 7 + c] = -h;
            Hs_dbg[((2 * (int64_t)num_edges + e) * 7 + r) * 7 + c] = -h;
            Hs_dbg[((3 * (int64_t)num_edges + e) * 7 + r) * 7 + c] = h;
        }
        for (int r = 0; r < 7; ++r) { gs_dbg[(0 * (int64_t)num_edges + e) * 7 + r] = (float)-g[r]; gs_dbg[(1 * (int64_t)num_edges + e) * 7 + r] = (float)g[r]; }
    }
    const int64_t io = ix - num_fix, jo = jx - num_fix; // rows of the fixed poses are dropped (gn_kernels.cu:84)
    const int64_t rhs = (int64_t)sys.Dp * LD;           // first row of the extra block row
    for (int r = 0; r < 7; ++r) {
        for (int c = 0; c <= r; ++c) { // diagonal blocks (i,i) and (j,j): lower triangle only
            if (io >= 0) atomicAdd(&A[(io * 7 + r) * LD + io * 7 + c], H[r][c]);
            if (jo >= 0) atomicAdd(&A[(jo * 7 + r) * LD + jo * 7 + c], H[r][c]);
        }
        if (io >= 0) atomicAdd(&A[rhs + io * 7 + r], -g[r]);
        if (jo >= 0) atomicAdd(&A[rhs + jo * 7 + r], g[r]);
    }
    // H_ij (rows of i, columns of j) = -H and H_ji = -H^T = -H: both land in the lower triangle of the pair
    if (io >= 0 && jo >= 0 && io != jo) {
        const int64_t hi = io > jo ? io : jo, lo = io > jo ? jo : io;
        for (int r = 0; r < 7; ++r) for (int c = 0; c < 7; ++c) atomicAdd(&A[(hi * 7 + r) * LD + lo * 7 + c], -H[r][c]);
    } else if (io >= 0 && io == jo) { // self edge (not produced by the graph builder): H_ij + H_ji folded on the diagonal block
        for (int r = 0; r < 7; ++r) for (int c = 0; c <= r; ++c) atomicAdd(&A[(io * 7 + r) * LD + io * 7 + c], -2.0 * H[r][c]);
    }
}

// grid: 1 + (block rows below kb, the right-hand-side row included); 256 threads
__global__ __launch_bounds__(256) void gn_panel_kernel(GnSys sys, int kb)
{
    __shared__ double Lt[GN_TB][GN_TB + 1];
    __shared__ double Rt[GN_TB][GN_TB + 1];
    __shared__ int bad;
    if (*sys.done) return;
    const int LD = sys.Dp, tid = threadIdx.x;
    double* __restrict__ A = sys.A;
    const int64_t d0 = (int64_t)kb * GN_TB;
    for (int i = tid; i < GN_TB * GN_TB; i += 256) { const int r = i / GN_TB, c = i % GN_TB; Lt[r][c] = c <= r ? A[(d0 + r) * LD + d0 + c] : 0.0; }
    if (tid == 0) bad = 0;
    __syncthreads();
    // unblocked Cholesky of the diagonal tile in LDS (right-looking)
    for (int k = 0; k < GN_TB; ++k) {
        if (tid == 0) {
            const double d = Lt[k][k];
            if (!(d > 0.0)) { bad = 1; Lt[k][k] = 1.0; } else Lt[k][k] = sqrt(d);
        }
        __syncthreads();
        const double pinv = 1.0 / Lt[k][k];
        if (tid > k && tid < GN_TB) Lt[tid][k] *= pinv;
        __syncthreads();
        for (int i = tid; i < GN_TB * GN_TB; i += 256) {
            const int r = i / GN_TB, c = i % GN_TB;
            if (c > k && c <= r) Lt[r][c] -= Lt[r][k] * Lt[c][k];
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        for (int i = tid; i < GN_TB * GN_TB; i += 256) { const int r = i / GN_TB, c = i % GN_TB; if (c <= r) A[(d0 + r) * LD + d0 + c] = Lt[r][c]; }
        if (tid == 0 && bad) *sys.fail = 1;
        return;
    }
    // block row r0: X L^T = R  =>  X[:,c] = (R[:,c] - sum_{m<c} X[:,m] L[c][m]) / L[c][c]; one thread per row
    const int64_t r0 = d0 + (int64_t)blockIdx.x * GN_TB;
    for (int i = tid; i < GN_TB * GN_TB; i += 256) { const int r = i / GN_TB, c = i % GN_TB; Rt[r][c] = A[(r0 + r) * LD + d0 + c]; }
    __syncthreads();
    if (tid < GN_TB) {
        for (int c = 0; c < GN_TB; ++c) {
            double x = Rt[tid][c];
            for (int m = 0; m < c; ++m) x -= Rt[tid][m] * Lt[c][m];
            Rt[tid][c] = x / Lt[c][c];
        }
    }
    __syncthreads();
    for (int i = tid; i < GN_TB * GN_TB; i += 256) { const int r = i / GN_TB, c = i % GN_TB; A[(r0 + r) * LD + d0 + c] = Rt[r][c]; }
}

// grid (x: block row i - kb - 1 incl. the rhs row, y: block column j - kb - 1); tiles above the diagonal return
__global__ __launch_bounds__(256) void gn_update_kernel(GnSys sys, int kb)
{
    __shared__ double Pi[GN_TB][GN_TB + 1];
    __shared__ double Pj[GN_TB][GN_TB + 1];
    if (*sys.done) return;
    const int bi = kb + 1 + blockIdx.x, bj = kb + 1 + blockIdx.y;
    if (bj > bi || bj >= sys.Dp / GN_TB) return; // bi == Dp / GN_TB is the right-hand-side row: it has no column of its own
    const int LD = sys.Dp, tid = threadIdx.x;
    double* __restrict__ A = sys.A;
    const int64_t d0 = (int64_t)kb * GN_TB, ri = (int64_t)bi * GN_TB, rj = (int64_t)bj * GN_TB;
    for (int i = tid; i < GN_TB * GN_TB; i += 256) {
        const int r = i / GN_TB, c = i % GN_TB;
        Pi[r][c] = A[(ri + r) * LD + d0 + c];
        Pj[r][c] = A[(rj + r) * LD + d0 + c];
    }
    __syncthreads();
    const int r = tid >> 3, c0 = (tid & 7) * 4; // 32 rows x 8 groups of 4 columns
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
    for (int k = 0; k < GN_TB; ++k) {
        const double a = Pi[r][k];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += a * Pj[c0 + q][k];
    }
    double* out = A + (ri + r) * LD + rj + c0;
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] -= acc[q];
}

// One workgroup: back substitution L^T x = y (y = the right-hand-side row), dx = -x (zeros when a pivot failed,
// gn_kernels.cu:141-160), retraction, |dx| test.
__global__ __launch_bounds__(GN_SOLVE_THREADS) void gn_finish_kernel(float* __restrict__ Twc, int num_poses, int num_fix, GnSys sys,
                                                                     float* __restrict__ dx_out, float delta_thresh)
{
    extern __shared__ double xs[]; // Dp
    __shared__ float nrm[GN_SOLVE_THREADS / 64];
    if (*sys.done) return;
    const int tid = threadIdx.x, nthr = blockDim.x, D = sys.D, LD = sys.Dp;
    const double* __restrict__ A = sys.A;
    const bool fail = *sys.fail != 0;
    if (!fail) {
        for (int i = tid; i < sys.Dp; i += nthr) xs[i] = A[(int64_t)sys.Dp * LD + i];
        __syncthreads();
        for (int k = D - 1; k >= 0; --k) { // column-oriented: row k of L is contiguous
            if (tid == 0) xs[k] = xs[k] / A[(int64_t)k * LD + k];
            __syncthreads();
            const double xk = xs[k];
            const double* row = A + (int64_t)k * LD;
            for (int i = tid; i < k; i += nthr) xs[i] -= row[i] * xk;
            __syncthreads();
        }
    }
    const int lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6;
    float part = 0.f;
    for (int i = tid; i < D; i += nthr) {
        const float d = fail ? 0.f : (float)(-xs[i]);
        dx_out[i] = d;
        part += d * d;
    }
    part = wave_sum(part);
    if (lane == 0) nrm[wv] = part;
    __syncthreads();
    for (int p = num_fix + tid; p < num_poses; p += nthr) {
        float xi[7];
        for (int c = 0; c < 7; ++c) xi[c] = dx_out[(p - num_fix) * 7 + c];
        retract_pose(Twc + 8 * (int64_t)p, xi);
    }
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < nwv; ++w) s += nrm[w];
        if (sqrtf(s) < delta_thresh) *sys.done = 1;
    }
}

} // namespace adk

static inline int64_t gn_align(int64_t x) { return (x + 255) & ~(int64_t)255; }
static inline int gn_chunks(int num_edges, int num_points) {
    int ch = num_edges > 0 ? (1024 + num_edges - 1) / num_edges : 1;
    const int maxch = (num_points + 255) / 256;
    if (ch > maxch) ch = maxch;
    if (ch > GN_MAX_CHUNKS) ch = GN_MAX_CHUNKS;
    if (ch < 1) ch = 1;
    return ch;
}

static inline int64_t gn_padded(int D) { return ((int64_t)D + GN_TB - 1) / GN_TB * GN_TB; }

// workspace: done + fail flags (256 B) | partials [E][chunks][35] f32 | A [(Dp + 32) x Dp] f64
extern "C" int64_t adk_gn_workspace_bytes(int num_poses, int num_edges, int num_points)
{
    if (num_poses < 0 || num_edges < 0 || num_points < 0) return ADK_EINVAL;
    const int64_t Dp = gn_padded(7 * (num_poses > 1 ? num_poses - 1 : 0)) + (num_poses > 1 ? 0 : GN_TB);
    return 256 + gn_align((int64_t)num_edges * gn_chunks(num_edges, num_points) * GN_NACC * 4) + gn_align((Dp + GN_TB) * Dp * 8) + 256;
}

// kind: 0 = points (sigma_a = sigma_point), 1 = rays (sigma_a = sigma_ray, sigma_b = sigma_dist),
// 2 = calib (sigma_a = sigma_pixel, sigma_b = sigma_depth; K [3,3], height, width, pixel_border, z_eps).
// Twc [P,8] (t, q xyzw, s) is updated in place; ii/jj [E] index the pose arrays (position in the sorted unique
// keyframe list, gn_kernels.cu:163-169); the first num_fix poses are held fixed.  dx_out [P-num_fix,7] = last step.
// Hs_dbg [4,E,7,7] / gs_dbg [2,E,7] (optional, both or neither): the reference's per-factor blocks of the LAST
// executed iteration (for tests).  No host synchronisation; all max_iter iterations are enqueued.
extern "C" int adk_gauss_newton(int kind, int num_poses, int num_edges, int num_points, float* Twc, const float* Xs,
                                const float* Cs, const float* K, const int64_t* ii, const int64_t* jj,
                                const int64_t* idx_ii2jj, const uint8_t* valid_match, const float* Q, int height, int width,
                                int pixel_border, float z_eps, float sigma_a, float sigma_b, float C_thresh, float Q_thresh,
                                int max_iter, float delta_thresh, int num_fix, float* dx_out, float* Hs_dbg, float* gs_dbg,
                                void* workspace, int64_t workspace_bytes, hipStream_t stream)
{
    if (kind < 0 || kind > 2 || num_poses < 0 || num_edges < 0 || num_points < 0 || max_iter < 0 || num_fix < 0) return ADK_EINVAL;
    if ((Hs_dbg == nullptr) != (gs_dbg == nullptr)) return ADK_EINVAL;
    if (num_poses <= num_fix || max_iter == 0) return 0;
    if (!Twc || !dx_out || !workspace) return ADK_EINVAL;
    if (num_edges > 0 && (!Xs || !Cs || !ii || !jj || !idx_ii2jj || !valid_match || !Q)) return ADK_EINVAL;
    if (kind == 2 && (!K || width <= 0 || height <= 0)) return ADK_EINVAL;
    if (num_fix != 1) return ADK_EUNSUPPORTED; // workspace is sized for the reference's num_fix = 1
    if (workspace_bytes < adk_gn_workspace_bytes(num_poses, num_edges, num_points) || ((uintptr_t)workspace & 255)) return ADK_EWORKSPACE;
    const int D = 7 * (num_poses - num_fix);
    const int Dp = (int)gn_padded(D), nbk = Dp / GN_TB;
    if ((int64_t)Dp * 8 > 160 * 1024 - 4096) return ADK_EUNSUPPORTED; // the solution vector lives in LDS during the back substitution
    const int chunks = gn_chunks(num_edges, num_points);
    char* w = (char*)workspace;
    int* done = (int*)w;
    float* partials = (float*)(w + 256);
    double* A = (double*)(w + 256 + gn_align((int64_t)num_edges * chunks * GN_NACC * 4));
    const size_t a_bytes = (size_t)(Dp + GN_TB) * Dp * sizeof(double);
    hipLaunchKernelGGL(adk::gn_clear_kernel, dim3(1), dim3(256), 0, stream, (uint4*)done, (int64_t)16);
    adk::GnArgs a;
    a.Twc = Twc; a.Xs = Xs; a.Cs = Cs; a.K = K; a.ii = ii; a.jj = jj; a.idx = idx_ii2jj; a.valid = valid_match; a.Q = Q;
    a.num_points = num_points;
    a.num_chunks = chunks;
    a.chunk = (int)(((int64_t)(num_points + chunks - 1) / chunks + 255) / 256 * 256);
    a.height = height; a.width = width; a.pixel_border = pixel_border; a.z_eps = z_eps;
    a.sigma_a = sigma_a; a.sigma_b = sigma_b; a.C_thresh = C_thresh; a.Q_thresh = Q_thresh;
    a.partials = partials; a.done = done;
    adk::GnSys sys;
    sys.A = A; sys.D = D; sys.Dp = Dp; sys.done = done; sys.fail = done + 1;
    const size_t lds = (size_t)Dp * sizeof(double);
    (void)hipFuncSetAttribute((const void*)adk::gn_finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int asm_items = num_edges > Dp - D ? num_edges : Dp - D;
    for (int it = 0; it < max_iter; ++it) {
        if (num_edges > 0) {
            const dim3 grid((unsigned)num_edges, (unsigned)chunks);
            if (kind == 0) hipLaunchKernelGGL(adk::gn_accumulate_kernel<0>, grid, dim3(256), 0, stream, a);
            else if (kind == 1) hipLaunchKernelGGL(adk::gn_accumulate_kernel<1>, grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL(adk::gn_accumulate_kernel<2>, grid, dim3(256), 0, stream, a);
        }
        // unconditional (cheap); the kernels below are no-ops once `done`
        hipLaunchKernelGGL(adk::gn_clear_kernel, dim3(adk::stream_grid((int64_t)(a_bytes / 16), 256)), dim3(256), 0, stream, (uint4*)A, (int64_t)(a_bytes / 16));
        hipLaunchKernelGGL(adk::gn_assemble_kernel, dim3((unsigned)adk::ceil_div(asm_items > 0 ? asm_items : 1, 256)), dim3(256), 0, stream,
                           (const float*)Twc, num_fix, ii, jj, num_edges, chunks, (const float*)partials, sys, Hs_dbg, gs_dbg);
        for (int kb = 0; kb < nbk; ++kb) {
            const int below = nbk - kb; // block rows under the diagonal tile, the right-hand-side row included
            hipLaunchKernelGGL(adk::gn_panel_kernel, dim3(1 + below), dim3(256), 0, stream, sys, kb);
            if (kb + 1 < nbk) hipLaunchKernelGGL(adk::gn_update_kernel, dim3(below, below - 1), dim3(256), 0, stream, sys, kb);
        }
        hipLaunchKernelGGL(adk::gn_finish_kernel, dim3(1), dim3(GN_SOLVE_THREADS), lds, stream, Twc, num_poses, num_fix, sys, dx_out, delta_thresh);
    }
    ADK_RETURN_LAST_ERROR();
}
