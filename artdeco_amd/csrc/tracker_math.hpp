// Per-point and per-iteration arithmetic of the frontend Sim(3) tracker, shared between the HIP
// kernels (tracker.hip) and a host-compiled test harness (tests/host/tracker_host.cpp, g++): everything here is
// plain C++ over scalars, so the Jacobians, weights, selection and retraction can be checked against
// oracle/tracker_oracle.py on the CPU-only build container; the kernels add only the parallel plumbing.
//
// Restates VSLAM/CameraTracker.py:296-396 (opt_pose_calib_sim3), :223-238 (solve), VSLAM/mast3r_slam/geometry.py:47-54
// (act_Sim3), :66-113 (project_calib), nonlinear_optimizer.py:5-34 (check_convergence, huber) and the pypose Sim(3)
// algebra they call (Exp, mul, Inv, Act, quat2unit; pip dependency, not vendored).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define ADK_HD __host__ __device__ inline
#else
#define ADK_HD inline
#endif

namespace adk {
namespace trk {

#define TRK_NS 28   // lower triangle of the symmetric 7x7
#define TRK_NACC 36 // + 7 gradient entries + the cost
// --optimize_focal (CameraTracker.py:308-320,367-377): an 8th unknown, the focal length shared by fx and fy
#define TRK_NS8 36
#define TRK_NACC8 45

struct Pose { float t[3], q[4], s; }; // q = xyzw

struct Cam { float fx, fy, cx, cy; int H, W; };

struct Cfg {
    float sigma_pixel_inv, sigma_depth_inv, huber_k, z_eps, border;
    float C_conf, Q_conf, min_match_frac;
    double rel_error, delta_norm;
};

ADK_HD Pose load_pose(const float* p) {
    Pose T;
    T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2];
    T.q[0] = p[3]; T.q[1] = p[4]; T.q[2] = p[5]; T.q[3] = p[6];
    T.s = p[7];
    return T;
}
ADK_HD void store_pose(const Pose& T, float* p) {
    p[0] = T.t[0]; p[1] = T.t[1]; p[2] = T.t[2];
    p[3] = T.q[0]; p[4] = T.q[1]; p[5] = T.q[2]; p[6] = T.q[3];
    p[7] = T.s;
}

ADK_HD void quat_mul(const float* a, const float* b, float* o) {
    const float o0 = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    const float o1 = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    const float o2 = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    const float o3 = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
}

// Y = R(q) X   (X + w u + q x u, u = 2 q x X)
ADK_HD void rot(const float* q, const float* X, float* Y) {
    const float u0 = 2.0f * (q[1] * X[2] - q[2] * X[1]);
    const float u1 = 2.0f * (q[2] * X[0] - q[0] * X[2]);
    const float u2 = 2.0f * (q[0] * X[1] - q[1] * X[0]);
    const float y0 = X[0] + q[3] * u0 + (q[1] * u2 - q[2] * u1);
    const float y1 = X[1] + q[3] * u1 + (q[2] * u0 - q[0] * u2);
    const float y2 = X[2] + q[3] * u2 + (q[0] * u1 - q[1] * u0);
    Y[0] = y0; Y[1] = y1; Y[2] = y2;
}

// Sim3.Act: s R p + t
ADK_HD void act(const Pose& T, const float* X, float* P) {
    rot(T.q, X, P);
    P[0] = P[0] * T.s + T.t[0];
    P[1] = P[1] * T.s + T.t[1];
    P[2] = P[2] * T.s + T.t[2];
}

ADK_HD Pose quat2unit(Pose T) {
    const float n = sqrtf(T.q[0] * T.q[0] + T.q[1] * T.q[1] + T.q[2] * T.q[2] + T.q[3] * T.q[3]);
    T.q[0] /= n; T.q[1] /= n; T.q[2] /= n; T.q[3] /= n;
    return T;
}

ADK_HD Pose mul(const Pose& A, const Pose& B) {
    Pose R;
    act(A, B.t, R.t);
    quat_mul(A.q, B.q, R.q);
    R.s = A.s * B.s;
    return R;
}

ADK_HD Pose inv(const Pose& T) {
    Pose R;
    R.q[0] = -T.q[0]; R.q[1] = -T.q[1]; R.q[2] = -T.q[2]; R.q[3] = T.q[3];
    float r[3];
    rot(R.q, T.t, r);
    R.t[0] = -r[0] / T.s; R.t[1] = -r[1] / T.s; R.t[2] = -r[2] / T.s;
    R.s = 1.0f / T.s;
    return R;
}

ADK_HD void cross3(const float* a, const float* b, float* o) {
    const float x0 = a[1] * b[2] - a[2] * b[1], x1 = a[2] * b[0] - a[0] * b[2], x2 = a[0] * b[1] - a[1] * b[0];
    o[0] = x0; o[1] = x1; o[2] = x2;
}

// sim3 Exp: q = exp(phi), s = e^sigma, t = (C I + A [phi]x + B [phi]x^2) tau, with the small-angle / small-sigma limits
#define TRK_EPS 1e-6f
ADK_HD Pose exp_sim3(const float* xi) {
    Pose D;
    const float tau[3] = {xi[0], xi[1], xi[2]}, phi[3] = {xi[3], xi[4], xi[5]};
    const float sigma = xi[6];
    const float scale = expf(sigma);
    const float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
    const float theta = sqrtf(theta_sq);
    float imag, real;
    if (theta_sq < TRK_EPS) {
        const float p4 = theta_sq * theta_sq;
        imag = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * p4;
        real = 1.0f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * p4;
    } else {
        imag = sinf(0.5f * theta) / theta;
        real = cosf(0.5f * theta);
    }
    D.q[0] = imag * phi[0]; D.q[1] = imag * phi[1]; D.q[2] = imag * phi[2]; D.q[3] = real;
    D.s = scale;
    float A, B, C;
    if (fabsf(sigma) < TRK_EPS) {
        C = 1.0f;
        if (fabsf(theta) < TRK_EPS) { A = 0.5f; B = 1.0f / 6.0f; }
        else { A = (1.0f - cosf(theta)) / theta_sq; B = (theta - sinf(theta)) / (theta_sq * theta); }
    } else {
        C = (scale - 1.0f) / sigma;
        if (fabsf(theta) < TRK_EPS) {
            const float s2 = sigma * sigma;
            A = ((sigma - 1.0f) * scale + 1.0f) / s2;
            B = (scale * 0.5f * s2 + scale - 1.0f - sigma * scale) / (s2 * sigma);
        } else {
            const float a = scale * sinf(theta), b = scale * cosf(theta), c = theta_sq + sigma * sigma;
            A = (a * sigma + (1.0f - b) * theta) / (theta * c);
            B = (C - ((b - 1.0f) * sigma + a * theta) / c) / theta_sq;
        }
    }
    float pt[3], ppt[3];
    cross3(phi, tau, pt);
    cross3(phi, pt, ppt);
    for (int i = 0; i < 3; ++i) D.t[i] = C * tau[i] + A * pt[i] + B * ppt[i];
    return D;
}

// T <- quat2unit(Exp(tau) o T)   (CameraTracker.py:373-374)
ADK_HD Pose retract(const float* tau, const Pose& T) { return quat2unit(mul(exp_sim3(tau), T)); }

// Order-preserving map float -> uint32 (ascending), and back.
ADK_HD uint32_t float_key(float f) {
    union { float f; uint32_t u; } c;
    c.f = f;
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}
ADK_HD float key_float(uint32_t k) {
    union { float f; uint32_t u; } c;
    c.u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return c.f;
}

// torch.quantile's rank arithmetic (float32) and torch.lerp's two-sided form.
ADK_HD void quantile_rank(float q, int64_t n, int64_t* lo, int64_t* hi, float* w) {
    const float rank = q * (float)(n - 1);
    const float fl = floorf(rank);
    *lo = (int64_t)fl;
    *hi = (int64_t)ceilf(rank);
    *w = rank - fl;
}
ADK_HD float lerp_torch(float a, float b, float w) { return w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.0f - w); }

// Three-level radix select over the 32-bit keys, most significant digit first: 11 + 11 + 10 bits.
#define TRK_BINS 2048
ADK_HD int digit_shift(int pass) { return pass == 0 ? 21 : (pass == 1 ? 10 : 0); }
ADK_HD uint32_t digit_of(uint32_t key, int pass) { return pass == 0 ? (key >> 21) : (pass == 1 ? ((key >> 10) & 2047u) : (key & 1023u)); }
// the digits above `pass` (pass 1: the top 11 bits, pass 2: the top 22 bits); pass 0 has no prefix
ADK_HD uint32_t prefix_of(uint32_t key, int pass) { return pass == 0 ? 0u : (pass == 1 ? (key >> 21) : (key >> 10)); }
ADK_HD uint32_t extend_prefix(uint32_t prefix, uint32_t digit, int pass) { return pass == 0 ? digit : (pass == 1 ? ((prefix << 11) | digit) : ((prefix << 10) | digit)); }

// State of one selection: after pass p, pfx_* hold the digits fixed so far and rem_* the ranks inside them.
struct Sel {
    int64_t n, rem_lo, rem_hi;
    uint32_t pfx_lo, pfx_hi;
    float w;
    int same; // pfx_lo == pfx_hi (one histogram serves both ranks)
};

// rem_* start as the global ranks floor / ceil of q (n - 1) and are refined pass by pass.
ADK_HD void sel_begin(Sel& s, float q, int64_t n) {
    s.n = n; s.pfx_lo = s.pfx_hi = 0u; s.same = 1; s.w = 0.f; s.rem_lo = s.rem_hi = 0;
    if (n > 0) {
        int64_t lo, hi;
        quantile_rank(q, n, &lo, &hi, &s.w);
        if (lo < 0) lo = 0;
        if (hi > n - 1) hi = n - 1;
        if (lo > hi) lo = hi;
        s.rem_lo = lo; s.rem_hi = hi;
    }
}
ADK_HD void sel_advance(Sel& s, int pass, uint32_t d_lo, int64_t r_lo, uint32_t d_hi, int64_t r_hi) {
    s.pfx_lo = extend_prefix(s.pfx_lo, d_lo, pass);
    s.pfx_hi = extend_prefix(s.pfx_hi, d_hi, pass);
    s.rem_lo = r_lo; s.rem_hi = r_hi;
    s.same = s.pfx_lo == s.pfx_hi;
}
// after pass 2 the prefixes are the two order statistics' keys
ADK_HD float sel_value(const Sel& s) { return s.n > 0 ? lerp_torch(key_float(s.pfx_lo), key_float(s.pfx_hi), s.w) : 0.f; }

// Walk `nbins` histogram bins until the cumulative count passes `rank` (0-based): returns the bin, *rem = rank inside it.
ADK_HD int locate_rank(const uint32_t* hist, int nbins, int64_t rank, int64_t* rem) {
    int64_t acc = 0;
    for (int b = 0; b < nbins; ++b) {
        const int64_t c = hist[b];
        if (rank < acc + c) { *rem = rank - acc; return b; }
        acc += c;
    }
    *rem = 0;
    return nbins - 1;
}

// Determinant of the projected pixel covariance J (sR) diag(var) (sR)^T J^T, J = d(u, v, log z)/dP
// (CameraTracker.py:336-344) in closed form: (fx fy s^3 / Z^3)^2 vx vy vz.
ADK_HD float cov_det(const Pose& T, const Cam& c, const float* Xf, float varprod) {
    float P[3];
    act(T, Xf, P);
    const float s3 = T.s * T.s * T.s;
    const float jd = c.fx * c.fy / (P[2] * P[2] * P[2]) * s3;
    return jd * jd * varprod;
}

ADK_HD float huber_w(float r, float k) {
    const float a = fabsf(r);
    return a < k ? 1.0f : k / a;
}

// acc[0..27] += w J J^T (lower triangle, row-major packed), acc[28..34] += w e J, acc[35] += 0.5 w e^2
ADK_HD void add_row(float* acc, const float* J, float w, float e) {
    int l = 0;
    for (int n = 0; n < 7; ++n) {
        const float wj = w * J[n];
        for (int m = 0; m <= n; ++m) acc[l++] += wj * J[m];
        acc[TRK_NS + n] += wj * e;
    }
    acc[TRK_NACC - 1] += 0.5f * w * e * e;
}

// One matched point.  Xf: the frame's point (frame camera); w0 = valid ? sqrt(Qk) : 0 (valid = match & confidences &
// keyframe depth); (uk, vk, logzk): the keyframe's measurement; det_ok: the covariance filter's verdict.
// e = h(x) - z (the reference's -r), Jp = dh/dtau (the reference's -J): H = sum w Jp^T Jp, v = sum w Jp^T e and
// tau = -H^-1 v, identical to the reference's H = A^T A, g = -A^T b.
ADK_HD void point_rows(const Pose& T, const Cam& c, const Cfg& g, const float* Xf, float w0, float uk, float vk, float logzk,
                       bool det_ok, float* acc) {
    float P[3];
    act(T, Xf, P);
    const bool vz = P[2] > g.z_eps;
    const float zinv = 1.0f / P[2];
    const float xz = P[0] * zinv, yz = P[1] * zinv;
    const float u = (c.fx * P[0] + c.cx * P[2]) / P[2], v = (c.fy * P[1] + c.cy * P[2]) / P[2];
    const bool vu = (u > g.border) && (u < (float)(c.W - 1) - g.border);
    const bool vv = (v > g.border) && (v < (float)(c.H - 1) - g.border);
    if (!(vz && vu && vv && det_ok) || !(w0 > 0.0f)) return;
    const float swp = w0 * g.sigma_pixel_inv, swd = w0 * g.sigma_depth_inv;
    const float e0 = u - uk, e1 = v - vk, e2 = logf(P[2]) - logzk;
    float J[7];
    J[0] = c.fx * zinv; J[1] = 0.f; J[2] = -c.fx * xz * zinv; J[3] = -c.fx * xz * yz; J[4] = c.fx * (1.f + xz * xz); J[5] = -c.fx * yz; J[6] = 0.f;
    add_row(acc, J, huber_w(swp * e0, g.huber_k) * swp * swp, e0);
    J[0] = 0.f; J[1] = c.fy * zinv; J[2] = -c.fy * yz * zinv; J[3] = -c.fy * (1.f + yz * yz); J[4] = c.fy * xz * yz; J[5] = c.fy * xz; J[6] = 0.f;
    add_row(acc, J, huber_w(swp * e1, g.huber_k) * swp * swp, e1);
    J[0] = 0.f; J[1] = 0.f; J[2] = zinv; J[3] = yz; J[4] = -xz; J[5] = 0.f; J[6] = 1.f;
    add_row(acc, J, huber_w(swd * e2, g.huber_k) * swd * swd, e2);
}

// ---- --optimize_focal ----------------------------------------------------------------------------------------------------
// acc[0..35] += w J J^T (8x8 lower triangle), acc[36..43] += w e J, acc[44] += 0.5 w e^2
ADK_HD void add_row8(float* acc, const float* J, float w, float e) {
    int l = 0;
    for (int n = 0; n < 8; ++n) {
        const float wj = w * J[n];
        for (int m = 0; m <= n; ++m) acc[l++] += wj * J[m];
        acc[TRK_NS8 + n] += wj * e;
    }
    acc[TRK_NACC8 - 1] += 0.5f * w * e * e;
}

// The matched frame point re-backprojected with the CURRENT focal (backproject, geometry.py:116-124) and its derivative
// with respect to the focal as the reference writes it (CameraTracker.py:313-316): (uf, vf) = the frame pixel, z its depth.
ADK_HD void frame_point_focal(const Cam& c, float uf, float vf, float z, float* X, float* dX) {
    X[0] = (uf - c.cx) / c.fx * z;
    X[1] = (vf - c.cy) / c.fy * z;
    X[2] = z;
    dX[0] = -(uf - c.cx) / (c.fx * c.fx) * z;
    dX[1] = -(vf - c.cy) / (c.fy * c.fy) * z;
    dX[2] = 0.f;
}

// point_rows with the focal column of project_calib (geometry.py:110-112, kept term for term -- including the division by
// z_inv^2): d = (s R) dXf/df, col = (x/z + fx (d0 z - d2 x) / z_inv^2, y/z + fy (d1 z - d2 y) / z_inv^2, d2 / z).
ADK_HD void point_rows_focal(const Pose& T, const Cam& c, const Cfg& g, float uf, float vf, float zf, float w0, float uk, float vk,
                             float logzk, bool det_ok, float* acc) {
    float Xf[3], dXf[3], P[3], d[3];
    frame_point_focal(c, uf, vf, zf, Xf, dXf);
    act(T, Xf, P);
    rot(T.q, dXf, d);
    d[0] *= T.s; d[1] *= T.s; d[2] *= T.s;
    const bool vz = P[2] > g.z_eps;
    const float zinv = 1.0f / P[2];
    const float xz = P[0] * zinv, yz = P[1] * zinv;
    const float u = (c.fx * P[0] + c.cx * P[2]) / P[2], v = (c.fy * P[1] + c.cy * P[2]) / P[2];
    const bool vu = (u > g.border) && (u < (float)(c.W - 1) - g.border);
    const bool vv = (v > g.border) && (v < (float)(c.H - 1) - g.border);
    if (!(vz && vu && vv && det_ok) || !(w0 > 0.0f)) return;
    const float swp = w0 * g.sigma_pixel_inv, swd = w0 * g.sigma_depth_inv;
    const float e0 = u - uk, e1 = v - vk, e2 = logf(P[2]) - logzk;
    const float zi2 = zinv * zinv;
    float J[8];
    J[0] = c.fx * zinv; J[1] = 0.f; J[2] = -c.fx * xz * zinv; J[3] = -c.fx * xz * yz; J[4] = c.fx * (1.f + xz * xz); J[5] = -c.fx * yz; J[6] = 0.f;
    J[7] = xz + c.fx * (d[0] * P[2] - d[2] * P[0]) / zi2;
    add_row8(acc, J, huber_w(swp * e0, g.huber_k) * swp * swp, e0);
    J[0] = 0.f; J[1] = c.fy * zinv; J[2] = -c.fy * yz * zinv; J[3] = -c.fy * (1.f + yz * yz); J[4] = c.fy * xz * yz; J[5] = c.fy * xz; J[6] = 0.f;
    J[7] = yz + c.fy * (d[1] * P[2] - d[2] * P[1]) / zi2;
    add_row8(acc, J, huber_w(swp * e1, g.huber_k) * swp * swp, e1);
    J[0] = 0.f; J[1] = 0.f; J[2] = zinv; J[3] = yz; J[4] = -xz; J[5] = 0.f; J[6] = 1.f;
    J[7] = zinv * d[2];
    add_row8(acc, J, huber_w(swd * e2, g.huber_k) * swd * swd, e2);
}

// Device-resident state of one tracking call.
struct State {
    float T[8];        // T_CkCf, the variable
    float Tk[8];       // quat2unit(T_WCk)
    double old_cost;   // +inf before the first iteration
    double cost;       // cost of the last linearisation
    int iters, done, fail, lost;
    float thr;         // covariance-filter threshold of the current iteration
    float tau[7];      // last step
    float dist_q;      // displacement quantile (check_keyframe_map)
    unsigned n_opt, n_kf, n_unique;
    float fx, fy;      // the focal lengths the iterations use (optimize_focal updates them; K itself is never written)
};

// H tau = -v by Cholesky in double (NV unknowns: acc = lower triangle, then the NV gradient entries); false when a pivot is
// not positive (torch.linalg.cholesky would raise).
template <int NV>
ADK_HD bool solve_chol(const double* acc, double* tau) {
    double L[NV][NV];
    int l = 0;
    for (int n = 0; n < NV; ++n) for (int m = 0; m <= n; ++m) L[n][m] = acc[l++];
    for (int k = 0; k < NV; ++k) {
        double d = L[k][k];
        for (int m = 0; m < k; ++m) d -= L[k][m] * L[k][m];
        if (!(d > 0.0) || !(d < 1e300)) return false;
        const double p = sqrt(d);
        L[k][k] = p;
        for (int r = k + 1; r < NV; ++r) {
            double x = L[r][k];
            for (int m = 0; m < k; ++m) x -= L[r][m] * L[k][m];
            L[r][k] = x / p;
        }
    }
    double y[NV];
    for (int r = 0; r < NV; ++r) {
        double x = -acc[NV * (NV + 1) / 2 + r];
        for (int m = 0; m < r; ++m) x -= L[r][m] * y[m];
        y[r] = x / L[r][r];
    }
    for (int r = NV - 1; r >= 0; --r) {
        double x = y[r];
        for (int m = r + 1; m < NV; ++m) x -= L[m][r] * tau[m];
        tau[r] = x / L[r][r];
    }
    for (int r = 0; r < NV; ++r) if (!(tau[r] == tau[r])) return false;
    return true;
}
ADK_HD bool solve7(const double* acc, double* tau) { return solve_chol<7>(acc, tau); }

// One Gauss-Newton step from the summed accumulators (CameraTracker.py:372-389): solve, retract, convergence test.
ADK_HD void gn_step(State& s, const double* acc, const Cfg& g) {
    double tau[7];
    const double cost = acc[TRK_NACC - 1];
    s.cost = cost;
    if (!solve7(acc, tau)) { s.fail = 1; s.done = 1; return; }
    float tf[7];
    double n2 = 0.0;
    for (int r = 0; r < 7; ++r) { tf[r] = (float)tau[r]; s.tau[r] = tf[r]; n2 += (double)tf[r] * (double)tf[r]; }
    store_pose(retract(tf, load_pose(s.T)), s.T);
    s.iters += 1;
    // check_convergence: |(old - new) / old| < rel_error or |tau| < delta_norm; old = inf gives nan -> false
    const double rel = fabs((s.old_cost - cost) / s.old_cost);
    if (rel < g.rel_error || sqrt(n2) < g.delta_norm) s.done = 1;
    s.old_cost = cost;
}

// The same step with the focal as 8th unknown (CameraTracker.py:372-389): the pose takes tau[:7], both focal lengths take
// tau[7], and the convergence test looks at tau[:7] only.
ADK_HD void gn_step_focal(State& s, const double* acc, const Cfg& g) {
    double tau[8];
    const double cost = acc[TRK_NACC8 - 1];
    s.cost = cost;
    if (!solve_chol<8>(acc, tau)) { s.fail = 1; s.done = 1; return; }
    float tf[8];
    double n2 = 0.0;
    for (int r = 0; r < 8; ++r) tf[r] = (float)tau[r];
    for (int r = 0; r < 7; ++r) { s.tau[r] = tf[r]; n2 += (double)tf[r] * (double)tf[r]; }
    store_pose(retract(tf, load_pose(s.T)), s.T);
    s.fx = s.fx + tf[7];
    s.fy = s.fy + tf[7];
    s.iters += 1;
    const double rel = fabs((s.old_cost - cost) / s.old_cost);
    if (rel < g.rel_error || sqrt(n2) < g.delta_norm) s.done = 1;
    s.old_cost = cost;
}

// 5x5 local variance product of the ray-constrained pointmap at pixel (px, py) (utils_uncertainty.py:5-53 on top of
// constrain_points_to_ray): z [H*W] is the only input that matters (x, y follow from z and the pixel);
// weights = finite & z > 0, reflect padding, variance floored at 1e-12.  var3 receives (vx, vy, vz).
ADK_HD void local_var(const float* Xcanon /* [n,3] */, const Cam& c, int px, int py, float* var3) {
    float sw = 0.f, s1[3] = {0.f, 0.f, 0.f}, s2[3] = {0.f, 0.f, 0.f};
    for (int dy = -2; dy <= 2; ++dy) {
        int y = py + dy;
        y = y < 0 ? -y : (y >= c.H ? 2 * (c.H - 1) - y : y);
        for (int dx = -2; dx <= 2; ++dx) {
            int x = px + dx;
            x = x < 0 ? -x : (x >= c.W ? 2 * (c.W - 1) - x : x);
            const float z = Xcanon[3 * ((int64_t)y * c.W + x) + 2];
            const float X[3] = {((float)x - c.cx) / c.fx * z, ((float)y - c.cy) / c.fy * z, z};
            const bool ok = (z > 0.f) && (fabsf(z) <= 3.402823466e38f) && (fabsf(X[0]) <= 3.402823466e38f) && (fabsf(X[1]) <= 3.402823466e38f);
            if (ok) {
                sw += 1.f;
                for (int a = 0; a < 3; ++a) { s1[a] += X[a]; s2[a] += X[a] * X[a]; }
            }
        }
    }
    const float denom = fmaxf(sw / 25.f, 1e-9f);
    for (int a = 0; a < 3; ++a) {
        const float mean = (s1[a] / 25.f) / denom, ex2 = (s2[a] / 25.f) / denom;
        var3[a] = fmaxf(ex2 - mean * mean, 1e-12f);
    }
}

} // namespace trk
} // namespace adk
