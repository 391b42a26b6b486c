// TEST INFRASTRUCTURE ONLY (never loaded by artdeco_amd/): the tracker's launch sequence (artdeco_amd/csrc/tracker.hip,
// adk_track_frame) replayed sequentially on the host over the SAME arithmetic header the kernels compile
// (artdeco_amd/csrc/tracker_math.hpp), so the CPU-only build container can check the Jacobians, weights, radix select,
// Cholesky step and retraction against oracle/tracker_oracle.py and the reference-generated goldens.  What it cannot
// check is the parallel plumbing (atomics, wave reductions, launches): that is what the -m gpu tests are for.
// Built by tests/test_tracker.py with:  g++ -O2 -ffp-contract=off -shared -fPIC -I artdeco_amd/csrc
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <vector>

#include "tracker_math.hpp"

using namespace adk::trk;

namespace {

struct Select {
    std::vector<uint32_t> hA, hB;
    Select() : hA(TRK_BINS), hB(TRK_BINS) {}
    // keys: the skip key 0xFFFFFFFF is ignored; n = population size.  Same three passes as trk_select_kernel.
    float run(const std::vector<uint32_t>& keys, int64_t n, float q) {
        Sel sel;
        sel_begin(sel, q, n);
        for (int pass = 0; pass < 3; ++pass) {
            std::fill(hA.begin(), hA.end(), 0u);
            std::fill(hB.begin(), hB.end(), 0u);
            const bool two = pass > 0 && !sel.same;
            for (uint32_t k : keys) {
                if (k == 0xFFFFFFFFu) continue;
                const uint32_t d = digit_of(k, pass);
                if (pass == 0) { hA[d]++; continue; }
                const uint32_t pf = prefix_of(k, pass);
                if (pf == sel.pfx_lo) hA[d]++;
                if (two && pf == sel.pfx_hi) hB[d]++;
            }
            int64_t r_lo, r_hi;
            const int b_lo = locate_rank(hA.data(), TRK_BINS, sel.rem_lo, &r_lo);
            const int b_hi = locate_rank(two ? hB.data() : hA.data(), TRK_BINS, sel.rem_hi, &r_hi);
            if (sel.n > 0) sel_advance(sel, pass, (uint32_t)b_lo, r_lo, (uint32_t)b_hi, r_hi);
        }
        return sel_value(sel);
    }
};

} // namespace

extern "C" float th_quantile(const float* x, int64_t n, float q)
{
    std::vector<uint32_t> keys(n);
    for (int64_t i = 0; i < n; ++i) keys[i] = float_key(x[i]);
    Select s;
    return s.run(keys, n, q);
}

extern "C" void th_exp_retract(const float* tau, const float* T_in, float* T_out)
{
    store_pose(retract(tau, load_pose(T_in)), T_out);
}

// Same arguments as adk_track_frame (minus workspace / stream), host pointers.
extern "C" int th_track_frame(int height, int width, const float* K, const float* Xf_canon, const float* Cf, float inv_Nf,
                              const float* Qf, const float* Xk_canon, const float* Ck, float inv_Nk, const float* Qk,
                              const int64_t* idx_f2k, const uint8_t* valid_match, const float* T_WCf, const float* T_WCk,
                              float sigma_pixel, float sigma_depth, float huber_k, float C_conf, float Q_conf,
                              float min_match_frac, int pixel_border, float depth_eps, float rel_error, float delta_norm,
                              int max_iters, int covariance_filter, int optimize_focal, float dist_quantile_q, float* result /* [32] */,
                              float* dbg_Xc,
                              float* dbg_var, uint8_t* dbg_valid_opt, float* dbg_acc0, float* dbg_thr /* [max_iters] */)
{
    const int n = height * width, H = height, W = width;
    Cam c;
    c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5]; c.H = H; c.W = W;
    Cfg g;
    g.sigma_pixel_inv = 1.0f / sigma_pixel; g.sigma_depth_inv = 1.0f / sigma_depth; g.huber_k = huber_k; g.z_eps = depth_eps;
    g.border = (float)pixel_border; g.C_conf = C_conf; g.Q_conf = Q_conf; g.min_match_frac = min_match_frac;
    g.rel_error = (double)rel_error; g.delta_norm = (double)delta_norm;
    // prepare
    std::vector<float> Xfc(4 * (size_t)n);
    for (int p = 0; p < n; ++p) {
        const int px = p % W, py = p / W;
        const float z = Xf_canon[3 * (size_t)p + 2];
        float var[3];
        local_var(Xf_canon, c, px, py, var);
        Xfc[4 * (size_t)p] = ((float)px - c.cx) / c.fx * z;
        Xfc[4 * (size_t)p + 1] = ((float)py - c.cy) / c.fy * z;
        Xfc[4 * (size_t)p + 2] = z;
        Xfc[4 * (size_t)p + 3] = var[0] * var[1] * var[2];
        if (dbg_Xc) for (int a = 0; a < 3; ++a) dbg_Xc[3 * (size_t)p + a] = Xfc[4 * (size_t)p + a];
        if (dbg_var) for (int a = 0; a < 3; ++a) dbg_var[3 * (size_t)p + a] = var[a];
    }
    // gather
    std::vector<float> recA(4 * (size_t)n), recB(2 * (size_t)n);
    std::vector<uint32_t> keys(n), seen(n, 0u);
    unsigned counts[3] = {0, 0, 0};
    for (int k = 0; k < n; ++k) {
        int64_t ix = idx_f2k[k];
        ix = ix < 0 ? 0 : (ix >= n ? n - 1 : ix);
        const bool vm = valid_match[k] != 0;
        const float q = sqrtf(Qf[ix] * Qk[k]);
        const bool kf = vm && (q > g.Q_conf);
        const bool opt = kf && (Cf[ix] * inv_Nf > g.C_conf) && (Ck[k] * inv_Nk > g.C_conf);
        const float zk = Xk_canon[3 * (size_t)k + 2];
        const bool vmeas = zk > g.z_eps;
        for (int a = 0; a < 4; ++a) recA[4 * (size_t)k + a] = Xfc[4 * (size_t)ix + a];
        if (optimize_focal) { recA[4 * (size_t)k] = (float)(int)(ix % W); recA[4 * (size_t)k + 1] = (float)(int)(ix / W); }
        recB[2 * (size_t)k] = (opt && vmeas) ? sqrtf(q) : 0.f;
        recB[2 * (size_t)k + 1] = vmeas ? logf(zk) : 0.f;
        if (vm && !seen[ix]) { seen[ix] = 1u; counts[2]++; }
        uint32_t key = 0xFFFFFFFFu;
        if (opt) {
            const float du = (float)((int)(ix % W) - (k % W)), dv = (float)((int)(ix / W) - (k / W));
            key = float_key(sqrtf(du * du + dv * dv));
        }
        keys[k] = key;
        counts[0] += opt; counts[1] += kf;
        if (dbg_valid_opt) dbg_valid_opt[k] = opt ? 1 : 0;
    }
    Select sel;
    result[22] = fmaxf(sel.run(keys, counts[0], dist_quantile_q), -INFINITY);
    // init
    State s;
    memset(&s, 0, sizeof(s));
    const Pose Tf = quat2unit(load_pose(T_WCf)), Tk = quat2unit(load_pose(T_WCk));
    store_pose(mul(inv(Tk), Tf), s.T);
    store_pose(Tk, s.Tk);
    s.old_cost = (double)INFINITY;
    s.thr = INFINITY;
    s.fx = K[0]; s.fy = K[4];
    s.lost = ((float)counts[0] / (float)n < g.min_match_frac) ? 1 : 0;
    s.done = s.lost;
    for (int it = 0; it < max_iters && !s.done; ++it) {
        const Pose T = load_pose(s.T);
        c.fx = s.fx; c.fy = s.fy;
        if (covariance_filter) {
            for (int k = 0; k < n; ++k) {
                float X[3] = {recA[4 * (size_t)k], recA[4 * (size_t)k + 1], recA[4 * (size_t)k + 2]}, dX[3];
                if (optimize_focal) frame_point_focal(c, recA[4 * (size_t)k], recA[4 * (size_t)k + 1], recA[4 * (size_t)k + 2], X, dX);
                uint32_t key = float_key(cov_det(T, c, X, recA[4 * (size_t)k + 3]));
                if (key == 0xFFFFFFFFu) key = 0xFFFFFFFEu;
                keys[k] = key;
            }
            s.thr = fmaxf(sel.run(keys, n, 0.9f), 1.0f);
            if (dbg_thr) dbg_thr[it] = s.thr;
        }
        // accumulate: one float accumulator set per "workgroup slot" of 256 threads x grid-stride, summed in double
        const int nacc = optimize_focal ? TRK_NACC8 : TRK_NACC;
        double acc[TRK_NACC8];
        for (int l = 0; l < nacc; ++l) acc[l] = 0.0;
        const int chunk = 4096;
        for (int k0 = 0; k0 < n; k0 += chunk) {
            float part[TRK_NACC8];
            for (int l = 0; l < nacc; ++l) part[l] = 0.f;
            for (int k = k0; k < n && k < k0 + chunk; ++k) {
                const bool det_ok = covariance_filter ? (key_float(keys[k]) < s.thr) : true;
                const float* a = &recA[4 * (size_t)k];
                if (optimize_focal)
                    point_rows_focal(T, c, g, a[0], a[1], a[2], recB[2 * (size_t)k], (float)(k % W), (float)(k / W), recB[2 * (size_t)k + 1], det_ok, part);
                else
                    point_rows(T, c, g, a, recB[2 * (size_t)k], (float)(k % W), (float)(k / W), recB[2 * (size_t)k + 1], det_ok, part);
            }
            for (int l = 0; l < nacc; ++l) acc[l] += (double)part[l];
        }
        if (dbg_acc0 && s.iters == 0) for (int l = 0; l < nacc; ++l) dbg_acc0[l] = (float)acc[l];
        if (optimize_focal) gn_step_focal(s, acc, g);
        else gn_step(s, acc, g);
    }
    const bool ok = !s.lost && !s.fail;
    if (ok) store_pose(quat2unit(mul(load_pose(s.Tk), load_pose(s.T))), result);
    else for (int i = 0; i < 8; ++i) result[i] = T_WCf[i];
    for (int i = 0; i < 8; ++i) result[8 + i] = s.T[i];
    result[16] = (float)s.lost; result[17] = (float)s.fail; result[18] = (float)s.iters;
    result[19] = (float)counts[0]; result[20] = (float)counts[1]; result[21] = (float)counts[2];
    result[23] = (float)s.cost; result[24] = (float)s.done; result[25] = s.thr;
    result[26] = s.fx; result[27] = s.fy;
    for (int i = 28; i < 32; ++i) result[i] = 0.f;
    return 0;
}

extern "C" void th_fuse_pointmap(int64_t n, const float* result, const float* Xkf, const float* Ckf, float* X_canon, float* C)
{
    if (result[16] != 0.f || result[17] != 0.f) return;
    const Pose T = load_pose(result + 8);
    for (int64_t k = 0; k < n; ++k) {
        float P[3];
        act(T, Xkf + 3 * k, P);
        const float c0 = C[k], c1 = Ckf[k], cs = c0 + c1;
        for (int a = 0; a < 3; ++a) X_canon[3 * k + a] = (c0 * X_canon[3 * k + a] + c1 * P[a]) / cs;
        C[k] = cs;
    }
}
