// Frontend Sim(3) tracker for gfx950: everything CameraTracker.track does between the MASt3R match
// and the keyframe decision, on the device.
//
// Replaces VSLAM/CameraTracker.py:62-153 -- get_points_poses (:189-219: constrain_points_to_ray, local_diag_cov_from_X1,
// keyframe measurements), the validity masks (:83-87), the insufficient-match test (:90-91), opt_pose_calib_sim3
// (:296-396: up to 50 Gauss-Newton iterations, each with act_Sim3, project_calib, the covariance filter's
// torch.det + torch.quantile, Huber weights, a 7x7 Cholesky, the pypose retraction and a cost-based stop), the counts
// behind check_keyframe (:159-167: sum, torch.unique) and check_keyframe_map (:170-186: torch.quantile of the match
// displacement), and the point fusion (:136-141).  In the reference that is ~100 torch launches and three blocking
// host reads (.item(), quantile, cholesky) PER ITERATION.
//
// Here (MI355X-first):
//   * one gather pass builds a 24-byte record per keyframe pixel (matched frame point, variance product, weight,
//     keyframe log-depth), so an iteration streams 5.5 MB at 512x384 instead of re-gathering through idx_f2k;
//   * the covariance determinant is evaluated in closed form, (fx fy s^3 / Z^3)^2 vx vy vz, instead of an LU of J S J^T;
//   * torch.quantile's two order statistics come from a three-level (11 + 11 + 10 bit) radix SELECT over order-preserving
//     keys: exact, no sort.  Histograms are built in LDS per workgroup and written out as per-workgroup partials; NO
//     global atomics on the data path -- measured on MI355X, the first version's 196 608 atomicAdds per pass into a
//     global histogram (most of them on a few hot bins, ~2.7 ns each) took 0.5 ms per kernel;
//   * "last workgroup to arrive finishes the job": the workgroup that draws the last ticket of a launch reduces the
//     partials (histogram totals + rank search, or the 36 normal-equation sums + 7x7 Cholesky + retraction +
//     convergence test), so an iteration is 4 launches (3 select passes + accumulate) and a device flag turns the
//     remaining enqueued iterations into no-ops;
//   * a call enqueues a CHUNK of iterations (no-op launches still cost ~4 us each on the GPU); the host reads one
//     32-float result and only calls again (resume = 1) in the rare case the chunk did not converge.
// The arithmetic lives in tracker_math.hpp (also compiled on the host by the tests); this file is the parallel plumbing.
#include "adk_common.hpp"
#include "tracker_math.hpp"

namespace adk {
using namespace trk;

#define TRK_BLOCK 256
#define TRK_SEL_BLOCKS 16
#define TRK_SEL_THREADS 1024
#define TRK_ACC_BLOCKS 64
#define TRK_SKIP_KEY 0xFFFFFFFFu

// per keyframe pixel: recA = (x, y, z, varprod) of the matched frame point, recB = (w0, logz_k): 24 bytes
struct TrkWs {
    State* state;
    Sel* sel;
    unsigned* arrive;      // [4] arrival tickets: 1 select passes, 2 accumulate
    unsigned* blk_counts;  // [pt_blocks][2] valid_opt / valid_kf per gather workgroup
    unsigned* blk_seen;    // [TRK_SEL_BLOCKS]
    unsigned* hist;        // [TRK_SEL_BLOCKS][2][TRK_BINS]
    uint8_t* seen;         // [n] frame pixels hit by a valid match (unique count)
    float4* Xfc;           // [n] constrained frame point + variance product
    float4* recA;          // [n]
    float2* recB;          // [n]
    unsigned* keys;        // [n] selection keys (displacement, then determinants)
    float* partials;       // [TRK_ACC_BLOCKS][TRK_NACC]
};

__device__ __forceinline__ Cam load_cam(const float* __restrict__ K, int H, int W) {
    Cam c;
    c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5]; c.H = H; c.W = W;
    return c;
}

// the camera the iterations use: K's principal point with the state's focal lengths (= K's own unless optimize_focal moved them)
__device__ __forceinline__ Cam state_cam(const float* __restrict__ K, int H, int W, const State* s) {
    Cam c = load_cam(K, H, W);
    c.fx = s->fx; c.fy = s->fy;
    return c;
}

__device__ __forceinline__ unsigned wave_sum_u(unsigned v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Every workgroup calls this after its global writes; exactly one (the last to arrive) gets `true` and may read what
// all the others wrote.  Release: fence, then the ticket; acquire: the ticket, then fence.  The counter is left at 0.
__device__ __forceinline__ bool arrive_last(unsigned* counter, int* flag_lds) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(counter, 1u);
        const int last = (t == gridDim.x - 1) ? 1 : 0;
        if (last) atomicExch(counter, 0u);
        *flag_lds = last;
    }
    __syncthreads();
    const bool last = *flag_lds != 0;
    if (last) __threadfence();
    return last;
}

// Clears the header of the workspace (state, tickets, `seen`).  A kernel rather than hipMemsetAsync: inside a captured
// hipGraph the memset node was observed (ROCm 7.2, MI355X) not to be ordered before the kernels that follow it.
__global__ __launch_bounds__(TRK_BLOCK) void trk_clear_kernel(uint4* __restrict__ p, int64_t n16)
{
    for (int64_t i = blockIdx.x * (int64_t)TRK_BLOCK + threadIdx.x; i < n16; i += (int64_t)gridDim.x * TRK_BLOCK) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ---- 1. frame side: constrain_points_to_ray + local_diag_cov_from_X1 (diagonal only) ------------------------------
__global__ __launch_bounds__(TRK_BLOCK) void trk_prepare_kernel(int n, int H, int W, const float* __restrict__ K,
                                                                const float* __restrict__ Xf_canon, float4* __restrict__ Xfc,
                                                                float* __restrict__ dbg_Xc, float* __restrict__ dbg_var)
{
    const int p = blockIdx.x * TRK_BLOCK + threadIdx.x;
    if (p >= n) return;
    const Cam c = load_cam(K, H, W);
    const int px = p % W, py = p / W;
    const float z = Xf_canon[3 * (int64_t)p + 2];
    const float x = ((float)px - c.cx) / c.fx * z, y = ((float)py - c.cy) / c.fy * z;
    float var[3];
    local_var(Xf_canon, c, px, py, var);
    Xfc[p] = make_float4(x, y, z, var[0] * var[1] * var[2]);
    if (dbg_Xc) { dbg_Xc[3 * (int64_t)p] = x; dbg_Xc[3 * (int64_t)p + 1] = y; dbg_Xc[3 * (int64_t)p + 2] = z; }
    if (dbg_var) { dbg_var[3 * (int64_t)p] = var[0]; dbg_var[3 * (int64_t)p + 1] = var[1]; dbg_var[3 * (int64_t)p + 2] = var[2]; }
}

// ---- 2. keyframe order: gather, masks, counts, displacement keys ---------------------------------------------------
__global__ __launch_bounds__(TRK_BLOCK) void trk_gather_kernel(
    int n, int W, const int64_t* __restrict__ idx_f2k, const uint8_t* __restrict__ valid_match, const float* __restrict__ Cf,
    float inv_Nf, const float* __restrict__ Ck, float inv_Nk, const float* __restrict__ Qf, const float* __restrict__ Qk,
    const float* __restrict__ Xk_canon, Cfg g, int focal, TrkWs ws, uint8_t* __restrict__ dbg_valid_opt)
{
    __shared__ unsigned wcnt[TRK_BLOCK / 64][2];
    const int k = blockIdx.x * TRK_BLOCK + threadIdx.x;
    bool opt = false, kf = false;
    if (k < n) {
        int64_t ix = idx_f2k[k];
        ix = ix < 0 ? 0 : (ix >= n ? n - 1 : ix);
        const bool vm = valid_match[k] != 0;
        const float q = sqrtf(Qf[ix] * Qk[k]);
        kf = vm && (q > g.Q_conf);
        opt = kf && (Cf[ix] * inv_Nf > g.C_conf) && (Ck[k] * inv_Nk > g.C_conf);
        const float zk = Xk_canon[3 * (int64_t)k + 2];
        const bool vmeas = zk > g.z_eps;
        float4 a = ws.Xfc[ix];
        if (focal) { a.x = (float)(int)(ix % W); a.y = (float)(int)(ix / W); } // the frame PIXEL: the point is rebuilt with the current focal
        ws.recA[k] = a;
        ws.recB[k] = make_float2((opt && vmeas) ? sqrtf(q) : 0.f, vmeas ? logf(zk) : 0.f);
        if (vm) ws.seen[ix] = 1; // benign race: every writer stores the same byte
        unsigned key = TRK_SKIP_KEY;
        if (opt) {
            const float du = (float)((int)(ix % W) - (k % W)), dv = (float)((int)(ix / W) - (k / W));
            key = float_key(sqrtf(du * du + dv * dv));
        }
        ws.keys[k] = key;
        if (dbg_valid_opt) dbg_valid_opt[k] = opt ? 1 : 0;
    }
    const unsigned c0 = __popcll(__ballot(opt)), c1 = __popcll(__ballot(kf));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { wcnt[wv][0] = c0; wcnt[wv][1] = c1; }
    __syncthreads();
    if (threadIdx.x < 2)
        ws.blk_counts[blockIdx.x * 2 + threadIdx.x] = (wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x]) + (wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x]);
}

// ---- radix select: torch.quantile without a sort ---------------------------------------------------------------------
// Which bin of the (LDS) histogram h[TRK_BINS] holds 0-based rank `rank`, and the rank inside that bin.
__device__ __forceinline__ void block_locate(const unsigned* h, int64_t rank, unsigned* wsum, int* out_bin, long long* out_rem)
{
    const int t = threadIdx.x;
    const unsigned a = h[2 * t], b = h[2 * t + 1], s = a + b;
    unsigned inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(inc, o, 64); if ((t & 63) >= o) inc += v; }
    if ((t & 63) == 63) wsum[t >> 6] = inc;
    if (t == 0) { *out_bin = TRK_BINS - 1; *out_rem = 0; }
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < (t >> 6); ++w) base += wsum[w];
    const int64_t excl = (int64_t)base + inc - s;
    if (s > 0 && rank >= excl && rank < excl + (int64_t)s) {
        const int64_t r = rank - excl;
        if (r < (int64_t)a) { *out_bin = 2 * t; *out_rem = r; } else { *out_bin = 2 * t + 1; *out_rem = r - a; }
    }
    __syncthreads();
}

// One pass (0, 1, 2) of the select.  MODE 0: the keys are the match displacements written by the gather kernel (population
// = the valid_opt matches; pass 0 also counts the distinct matched frame pixels).  MODE 1: the keys are the covariance
// determinants of the current pose, computed and stored by pass 0 (population = all n points, as the reference).
// The last workgroup to arrive totals the per-workgroup histograms, finds the bins of the two ranks and advances *ws.sel;
// after pass 2 it writes max(quantile, floor_value) to *out_value.
template <int MODE>
__global__ __launch_bounds__(TRK_SEL_THREADS) void trk_select_kernel(int n, int pass, int num_gather_blocks, int H, int W,
                                                                     const float* __restrict__ K, int focal, TrkWs ws, float q, float floor_value,
                                                                     float* __restrict__ out_value)
{
    __shared__ unsigned hA[TRK_BINS], hB[TRK_BINS];
    __shared__ unsigned wsum[TRK_SEL_THREADS / 64];
    __shared__ int lastf, bin_lo, bin_hi;
    __shared__ long long rem_lo, rem_hi;
    __shared__ unsigned popn;
    if (MODE == 1 && ws.state->done) return;
    if (pass > 0 && ws.sel->n < 0) return; // pass 0 already knew the answer (quantile below the floor)
    const int tid = threadIdx.x;
    Sel sel;
    if (pass > 0) sel = *ws.sel;
    else sel_begin(sel, q, 0);
    const bool two = pass > 0 && !sel.same;
    hA[tid] = 0u; hA[tid + TRK_SEL_THREADS] = 0u; hB[tid] = 0u; hB[tid + TRK_SEL_THREADS] = 0u;
    __syncthreads();
    Cam c;
    Pose T;
    if (MODE == 1 && pass == 0) { c = state_cam(K, H, W, ws.state); T = load_pose(ws.state->T); }
    unsigned seen_cnt = 0;
    for (int k = blockIdx.x * TRK_SEL_THREADS + tid; k < n; k += TRK_SEL_BLOCKS * TRK_SEL_THREADS) {
        unsigned key;
        if (MODE == 1 && pass == 0) {
            const float4 a = ws.recA[k];
            float X[3] = {a.x, a.y, a.z};
            if (focal) { float dX[3]; frame_point_focal(c, a.x, a.y, a.z, X, dX); }
            key = float_key(cov_det(T, c, X, a.w));
            if (key == TRK_SKIP_KEY) key = TRK_SKIP_KEY - 1u; // keep every point in the population
            ws.keys[k] = key;
        } else {
            key = ws.keys[k];
        }
        if (MODE == 0 && pass == 0) seen_cnt += ws.seen[k];
        if (key == TRK_SKIP_KEY) continue;
        const unsigned d = digit_of(key, pass);
        if (pass == 0) {
            atomicAdd(&hA[d], 1u);
        } else {
            const unsigned pf = prefix_of(key, pass);
            if (pf == sel.pfx_lo) atomicAdd(&hA[d], 1u);
            if (two && pf == sel.pfx_hi) atomicAdd(&hB[d], 1u);
        }
    }
    __syncthreads();
    unsigned* out = ws.hist + (size_t)blockIdx.x * 2 * TRK_BINS;
    out[tid] = hA[tid]; out[tid + TRK_SEL_THREADS] = hA[tid + TRK_SEL_THREADS];
    if (two) { out[TRK_BINS + tid] = hB[tid]; out[TRK_BINS + tid + TRK_SEL_THREADS] = hB[tid + TRK_SEL_THREADS]; }
    if (MODE == 0 && pass == 0) {
        seen_cnt = wave_sum_u(seen_cnt);
        if ((tid & 63) == 0) wsum[tid >> 6] = seen_cnt;
        __syncthreads();
        if (tid == 0) { unsigned s = 0; for (int w = 0; w < TRK_SEL_THREADS / 64; ++w) s += wsum[w]; ws.blk_seen[blockIdx.x] = s; }
    }
    if (!arrive_last(&ws.arrive[1], &lastf)) return;
    // ---- the last workgroup: totals, rank search, next prefix
    if (MODE == 0 && pass == 0) { // valid_opt / valid_kf counts of the gather workgroups (fixed order)
        unsigned s0 = 0, s1 = 0;
        for (int b = tid; b < num_gather_blocks; b += TRK_SEL_THREADS) { s0 += ws.blk_counts[2 * b]; s1 += ws.blk_counts[2 * b + 1]; }
        s0 = wave_sum_u(s0); s1 = wave_sum_u(s1);
        __syncthreads();
        if ((tid & 63) == 0) { hA[tid >> 6] = s0; hB[tid >> 6] = s1; }
        __syncthreads();
        if (tid == 0) {
            unsigned t0 = 0, t1 = 0;
            for (int w = 0; w < TRK_SEL_THREADS / 64; ++w) { t0 += hA[w]; t1 += hB[w]; }
            ws.state->n_opt = t0; ws.state->n_kf = t1;
            popn = t0;
        }
        __syncthreads();
    }
    if (pass == 0) sel_begin(sel, q, MODE == 0 ? (int64_t)popn : (int64_t)n);
    __syncthreads(); // everybody is done with hA / hB / wsum of the counting phase
    for (int i = tid; i < TRK_BINS; i += TRK_SEL_THREADS) {
        unsigned a = 0, b = 0;
        for (int blk = 0; blk < TRK_SEL_BLOCKS; ++blk) {
            a += ws.hist[((size_t)blk * 2) * TRK_BINS + i];
            if (two) b += ws.hist[((size_t)blk * 2 + 1) * TRK_BINS + i];
        }
        hA[i] = a; hB[i] = b;
    }
    __syncthreads();
    block_locate(hA, sel.rem_lo, wsum, &bin_lo, &rem_lo);
    block_locate(two ? hB : hA, sel.rem_hi, wsum, &bin_hi, &rem_hi);
    if (tid == 0) {
        if (sel.n > 0) sel_advance(sel, pass, (unsigned)bin_lo, rem_lo, (unsigned)bin_hi, rem_hi);
        // every key whose leading digit is below the floor's is below the floor: max(quantile, floor) = floor, passes 1-2 idle
        if (pass == 0 && sel.n > 0 && floor_value > -INFINITY && sel.pfx_hi < digit_of(float_key(floor_value), 0)) {
            sel.n = -1;
            *out_value = floor_value;
        }
        *ws.sel = sel;
        if (pass == 2) *out_value = fmaxf(sel_value(sel), floor_value);
        if (MODE == 0 && pass == 0) {
            unsigned s = 0;
            for (int blk = 0; blk < TRK_SEL_BLOCKS; ++blk) s += ws.blk_seen[blk];
            ws.state->n_unique = s;
        }
    }
}

// ---- 3. optimisation -----------------------------------------------------------------------------------------------------
__global__ void trk_init_kernel(int n, const float* __restrict__ K, const float* __restrict__ T_WCf, const float* __restrict__ T_WCk, Cfg g,
                                TrkWs ws)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    State& s = *ws.state;
    s.fx = K[0]; s.fy = K[4];
    const Pose Tf = quat2unit(load_pose(T_WCf)), Tk = quat2unit(load_pose(T_WCk)); // get_points_poses :194-195
    store_pose(mul(inv(Tk), Tf), s.T);                                              // :303
    store_pose(Tk, s.Tk);
    s.old_cost = (double)INFINITY;
    s.cost = 0.0;
    s.iters = 0; s.fail = 0;
    s.thr = INFINITY;
    for (int r = 0; r < 7; ++r) s.tau[r] = 0.f;
    // CameraTracker.py:90-91: valid_opt.sum() / numel < min_match_frac -> lost, no optimisation
    s.lost = ((float)s.n_opt / (float)n < g.min_match_frac) ? 1 : 0;
    s.done = s.lost;
}

// normal equations of the current linearisation; the last workgroup to arrive sums the partials in a fixed order (fp64),
// solves the 7x7 system, retracts and tests for convergence (gn_step)
template <bool FOCAL>
__global__ __launch_bounds__(TRK_BLOCK) void trk_accumulate_kernel(int n, int H, int W, const float* __restrict__ K, Cfg g,
                                                                   int use_cov, TrkWs ws, float* __restrict__ dbg_acc0)
{
    constexpr int NACC = FOCAL ? TRK_NACC8 : TRK_NACC;
    __shared__ float red[TRK_BLOCK / 64][NACC];
    __shared__ double sum[NACC];
    __shared__ int lastf;
    if (ws.state->done) return;
    const Cam c = state_cam(K, H, W, ws.state);
    const Pose T = load_pose(ws.state->T);
    const float thr = use_cov ? ws.state->thr : 0.f;
    float acc[NACC];
#pragma unroll
    for (int l = 0; l < NACC; ++l) acc[l] = 0.f;
    for (int k = blockIdx.x * TRK_BLOCK + threadIdx.x; k < n; k += TRK_ACC_BLOCKS * TRK_BLOCK) {
        const float4 a = ws.recA[k];
        const float2 b = ws.recB[k];
        const bool det_ok = use_cov ? (key_float(ws.keys[k]) < thr) : true;
        if (FOCAL) {
            point_rows_focal(T, c, g, a.x, a.y, a.z, b.x, (float)(k % W), (float)(k / W), b.y, det_ok, acc);
        } else {
            const float X[3] = {a.x, a.y, a.z};
            point_rows(T, c, g, X, b.x, (float)(k % W), (float)(k / W), b.y, det_ok, acc);
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int l = 0; l < NACC; ++l) { const float t = wave_sum_to_lane63(acc[l]); if (lane == 63) red[wv][l] = t; }
    __syncthreads();
    if (threadIdx.x < NACC)
        ws.partials[blockIdx.x * NACC + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (!arrive_last(&ws.arrive[2], &lastf)) return;
    if (threadIdx.x < NACC) {
        double s = 0.0;
        for (int b = 0; b < TRK_ACC_BLOCKS; ++b) s += (double)ws.partials[b * NACC + threadIdx.x];
        sum[threadIdx.x] = s;
        if (dbg_acc0 && ws.state->iters == 0) dbg_acc0[threadIdx.x] = (float)s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FOCAL) gn_step_focal(*ws.state, sum, g);
        else gn_step(*ws.state, sum, g);
    }
}

// result[0..7] = new T_WCf (the input pose when lost / failed), [8..15] = T_CkCf, [16] lost, [17] failed, [18] iterations,
// [19] n_opt, [20] n_kf, [21] n_unique, [22] displacement quantile, [23] last cost, [24] finished (converged, lost or
// failed: nothing left to iterate), [25] last covariance threshold, [26] fx, [27] fy as the iterations left them, [28..31] 0
__global__ void trk_finish_kernel(const float* __restrict__ T_WCf, TrkWs ws, float* __restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const State& s = *ws.state;
    const bool ok = !s.lost && !s.fail;
    if (ok) store_pose(quat2unit(mul(load_pose(s.Tk), load_pose(s.T))), out); // :395 + :134
    else for (int i = 0; i < 8; ++i) out[i] = T_WCf[i];
    for (int i = 0; i < 8; ++i) out[8 + i] = s.T[i];
    out[16] = (float)s.lost; out[17] = (float)s.fail; out[18] = (float)s.iters;
    out[19] = (float)s.n_opt; out[20] = (float)s.n_kf; out[21] = (float)s.n_unique;
    out[22] = s.dist_q; out[23] = (float)s.cost; out[24] = (float)s.done; out[25] = s.thr;
    out[26] = s.fx; out[27] = s.fy;
    for (int i = 28; i < 32; ++i) out[i] = 0.f;
}

// Point fusion (CameraTracker.py:136-141 + ImageFrame.update_pointmap :30-48), gated on the device-side success flags:
// X_canon = (C X_canon + Ckf (T_CkCf Xkf)) / (C + Ckf);  C += Ckf.
__global__ __launch_bounds__(TRK_BLOCK) void trk_fuse_kernel(int n, const float* __restrict__ result, const float* __restrict__ Xkf,
                                                             const float* __restrict__ Ckf, float* __restrict__ X_canon,
                                                             float* __restrict__ C)
{
    const int k = blockIdx.x * TRK_BLOCK + threadIdx.x;
    if (k >= n || result[16] != 0.f || result[17] != 0.f) return;
    const Pose T = load_pose(result + 8);
    const float X[3] = {Xkf[3 * (int64_t)k], Xkf[3 * (int64_t)k + 1], Xkf[3 * (int64_t)k + 2]};
    float P[3];
    act(T, X, P);
    const float c0 = C[k], c1 = Ckf[k], cs = c0 + c1;
    for (int a = 0; a < 3; ++a) X_canon[3 * (int64_t)k + a] = (c0 * X_canon[3 * (int64_t)k + a] + c1 * P[a]) / cs;
    C[k] = cs;
}

} // namespace adk

static inline int64_t trk_align(int64_t x) { return (x + 255) & ~(int64_t)255; }

struct TrkLayout { int64_t state, sel, arrive, seen, zero_bytes, blk_counts, blk_seen, hist, xfc, recA, recB, keys, partials, total; };

static TrkLayout trk_layout(int64_t n)
{
    TrkLayout L;
    int64_t o = 0;
    L.state = o; o += trk_align(sizeof(adk::trk::State));
    L.sel = o; o += trk_align(sizeof(adk::trk::Sel));
    L.arrive = o; o += 256;
    L.seen = o; o += trk_align(n);
    L.zero_bytes = o; // everything above is cleared at the start of a (non-resumed) call
    L.blk_counts = o; o += trk_align(adk::ceil_div(n, TRK_BLOCK) * 2 * 4);
    L.blk_seen = o; o += trk_align(TRK_SEL_BLOCKS * 4);
    L.hist = o; o += trk_align((int64_t)TRK_SEL_BLOCKS * 2 * TRK_BINS * 4);
    L.xfc = o; o += trk_align(n * 16);
    L.recA = o; o += trk_align(n * 16);
    L.recB = o; o += trk_align(n * 8);
    L.keys = o; o += trk_align(n * 4);
    L.partials = o; o += trk_align((int64_t)TRK_ACC_BLOCKS * TRK_NACC8 * 4);
    L.total = o;
    return L;
}

extern "C" int64_t adk_track_workspace_bytes(int height, int width)
{
    if (height <= 0 || width <= 0) return ADK_EINVAL;
    return trk_layout((int64_t)height * width).total;
}

extern "C" int adk_track_frame(int height, int width, const float* K, const float* Xf_canon, const float* Cf, float inv_Nf,
                               const float* Qf, const float* Xk_canon, const float* Ck, float inv_Nk, const float* Qk,
                               const int64_t* idx_f2k, const uint8_t* valid_match, const float* T_WCf, const float* T_WCk,
                               float sigma_pixel, float sigma_depth, float huber_k, float C_conf, float Q_conf,
                               float min_match_frac, int pixel_border, float depth_eps, float rel_error, float delta_norm,
                               int num_iters, int covariance_filter, int optimize_focal, float dist_quantile_q, int resume, float* result,
                               float* dbg_Xc, float* dbg_var, uint8_t* dbg_valid_opt, float* dbg_acc0, void* workspace,
                               int64_t workspace_bytes, hipStream_t stream)
{
    using namespace adk;
    if (height < 3 || width < 3 || num_iters < 0) return ADK_EINVAL; // the 5x5 reflect window needs >= 3 rows / columns
    if (!K || !Xf_canon || !Cf || !Qf || !Xk_canon || !Ck || !Qk || !idx_f2k || !valid_match || !T_WCf || !T_WCk || !result || !workspace)
        return ADK_EINVAL;
    if (!(sigma_pixel > 0.f) || !(sigma_depth > 0.f) || !(dist_quantile_q >= 0.f && dist_quantile_q <= 1.f)) return ADK_EINVAL;
    const int64_t n64 = (int64_t)height * width;
    if (n64 >= (1ll << 31)) return ADK_EUNSUPPORTED;
    const int n = (int)n64;
    const TrkLayout L = trk_layout(n);
    if (workspace_bytes < L.total || ((uintptr_t)workspace & 255)) return ADK_EWORKSPACE;
    char* w = (char*)workspace;
    TrkWs ws;
    ws.state = (State*)(w + L.state); ws.sel = (Sel*)(w + L.sel); ws.arrive = (unsigned*)(w + L.arrive);
    ws.blk_counts = (unsigned*)(w + L.blk_counts); ws.blk_seen = (unsigned*)(w + L.blk_seen); ws.hist = (unsigned*)(w + L.hist);
    ws.seen = (uint8_t*)(w + L.seen); ws.Xfc = (float4*)(w + L.xfc); ws.recA = (float4*)(w + L.recA); ws.recB = (float2*)(w + L.recB);
    ws.keys = (unsigned*)(w + L.keys); ws.partials = (float*)(w + L.partials);
    Cfg g;
    g.sigma_pixel_inv = 1.0f / sigma_pixel; g.sigma_depth_inv = 1.0f / sigma_depth; g.huber_k = huber_k; g.z_eps = depth_eps;
    g.border = (float)pixel_border; g.C_conf = C_conf; g.Q_conf = Q_conf; g.min_match_frac = min_match_frac;
    g.rel_error = (double)rel_error; g.delta_norm = (double)delta_norm;

    const int pt_blocks = (int)ceil_div(n, TRK_BLOCK);
    if (!resume) {
        hipLaunchKernelGGL(trk_clear_kernel, dim3(stream_grid(L.zero_bytes / 16, TRK_BLOCK)), dim3(TRK_BLOCK), 0, stream, (uint4*)w, L.zero_bytes / 16);
        hipLaunchKernelGGL(trk_prepare_kernel, dim3(pt_blocks), dim3(TRK_BLOCK), 0, stream, n, height, width, K, Xf_canon, ws.Xfc, dbg_Xc, dbg_var);
        hipLaunchKernelGGL(trk_gather_kernel, dim3(pt_blocks), dim3(TRK_BLOCK), 0, stream, n, width, idx_f2k, valid_match, Cf, inv_Nf, Ck, inv_Nk,
                           Qf, Qk, Xk_canon, g, optimize_focal, ws, dbg_valid_opt);
        // displacement quantile over the valid_opt matches (check_keyframe_map :181-183) + distinct matched frame pixels
        for (int pass = 0; pass < 3; ++pass)
            hipLaunchKernelGGL(trk_select_kernel<0>, dim3(TRK_SEL_BLOCKS), dim3(TRK_SEL_THREADS), 0, stream, n, pass, pt_blocks, height, width, K,
                               0, ws, dist_quantile_q, -INFINITY, &ws.state->dist_q);
        hipLaunchKernelGGL(trk_init_kernel, dim3(1), dim3(64), 0, stream, n, K, T_WCf, T_WCk, g, ws);
    }
    for (int it = 0; it < num_iters; ++it) {
        if (covariance_filter)
            for (int pass = 0; pass < 3; ++pass)
                hipLaunchKernelGGL(trk_select_kernel<1>, dim3(TRK_SEL_BLOCKS), dim3(TRK_SEL_THREADS), 0, stream, n, pass, pt_blocks, height, width, K,
                                   optimize_focal, ws, 0.9f, 1.0f, &ws.state->thr);
        if (optimize_focal)
            hipLaunchKernelGGL(trk_accumulate_kernel<true>, dim3(TRK_ACC_BLOCKS), dim3(TRK_BLOCK), 0, stream, n, height, width, K, g,
                               covariance_filter, ws, dbg_acc0);
        else
            hipLaunchKernelGGL(trk_accumulate_kernel<false>, dim3(TRK_ACC_BLOCKS), dim3(TRK_BLOCK), 0, stream, n, height, width, K, g,
                               covariance_filter, ws, dbg_acc0);
    }
    hipLaunchKernelGGL(trk_finish_kernel, dim3(1), dim3(64), 0, stream, T_WCf, ws, result);
    ADK_RETURN_LAST_ERROR();
}

extern "C" int adk_track_fuse_pointmap(int64_t n, const float* result, const float* Xkf, const float* Ckf, float* X_canon, float* C,
                                       hipStream_t stream)
{
    if (n < 0) return ADK_EINVAL;
    if (n == 0) return 0;
    if (n >= (1ll << 31)) return ADK_EUNSUPPORTED;
    if (!result || !Xkf || !Ckf || !X_canon || !C) return ADK_EINVAL;
    hipLaunchKernelGGL(adk::trk_fuse_kernel, dim3((unsigned)adk::ceil_div(n, TRK_BLOCK)), dim3(TRK_BLOCK), 0, stream, (int)n, result, Xkf,
                       Ckf, X_canon, C);
    ADK_RETURN_LAST_ERROR();
}
