// Fused multi-head attention forward for the MASt3R frontend: softmax(q k^T * scale) v, head dim 64,
// fp16 operands, fp32 scores / softmax / accumulation -- the "TF32-class" arithmetic of artdeco_amd/mast3r_model.py.
//
// Replaces the attention of croco/models/blocks.py:97-111 (self) and :138-157 (cross) -- `attn = (q @ k.transpose(-2,-1)) *
// scale; attn = attn.softmax(-1); x = (attn @ v).transpose(1, 2).reshape(B, N, C)` -- for the shapes the frontend runs:
// 768 tokens (512x384 / 16^2), 16 heads (encoder) or 12 (decoder), B = 1.  At this size a library flash kernel launches 96
// workgroups for 256 CUs and takes 34 us per call, 72 calls per tracked frame (profiles/r02_frontend_kernel_stats.csv).
//
// Design (gfx950, wave64, v_mfma_f32_16x16x32_f16):
//   * workgroup = 64 query rows of one (batch, head), wave w of each 4-wave group owns 16 query rows; 12 x 16 = 192
//     workgroups at 768 tokens / 16 heads -- one per CU; up to 3 wave groups per workgroup split the key tiles (below);
//   * K / V are walked in tiles of 64 keys, staged once per workgroup in LDS (row pitch 144 B for K: the 16 rows a
//     ds_read_b128 fragment load touches land on 16 different 4-bank slots; 160 B for V: conflict-free for the
//     transposing reads), the next tile's global loads in flight in registers while the current one is consumed;
//   * SWAPPED first product, S^T = K Q^T: in the accumulator layout lane (g, c) = (lane >> 4, lane & 15) then holds, for
//     ONE query c, the scores of keys 16 kb + 4 g + r -- the softmax statistics of a row are 16 lane-local values plus two
//     permlane swaps (lanes c, c+16, c+32, c+48), and the probabilities are ALREADY the B operand of the second product:
//     k-slot (g, e) of the MFMA is bound to key 32 s + 4 g + e (e < 4) / 32 s + 16 + 4 g + e - 4 (e >= 4), an order the
//     sum over keys does not care about, as long as the V fragment uses the same one;
//   * second product also transposed, O^T = V^T P^T, so that the accumulator column is again the query of lane & 15 and the
//     online-softmax rescale exp2(m_old - m_new) is a lane-local multiply (no redistribution of per-row factors);
//   * the V fragment wants V[key(g, e)][d = lane & 15]: 8 different ROWS per lane.  ds_read_b64_tr_b16 (gfx950's LDS
//     transpose read: inside each 16-lane group, lane c element j receives element c & 3 of the 8 bytes addressed by lane
//     4 j + (c >> 2)) delivers it from the row-major tile in two instructions per fragment;
//   * exp2 with the scale and log2(e) folded into one fma per score.
#include "adk_common.hpp"

namespace adk {

typedef _Float16 att_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 att_f16x4 __attribute__((ext_vector_type(4)));
typedef float att_f32x4 __attribute__((ext_vector_type(4)));
typedef short att_s16x4 __attribute__((ext_vector_type(4)));

#define ATT_D 64
#define ATT_QB 64   // query rows per workgroup
#define ATT_KB 64   // keys per tile
#define ATT_KP 72   // K tile row pitch in halves (144 B)
#define ATT_VP 80   // V tile row pitch in halves (160 B)

struct AttnArgs {
    const _Float16 *q, *k, *v;
    _Float16* out;
    int H, Nq, Nk;
    int64_t q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn; // element strides: batch, head, token
    float scale_log2e;
};

__device__ __forceinline__ float att_max3(float x, float y, float z) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}
__device__ __forceinline__ float att_xor_max(float v) { // max over lanes c, c+16, c+32, c+48
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(sw.x), __uint_as_float(sw.y));
    sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(sw.x), __uint_as_float(sw.y));
}
__device__ __forceinline__ float att_xor_sum(float v) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(sw.x) + __uint_as_float(sw.y);
    sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(sw.x) + __uint_as_float(sw.y);
}

// NSPLIT wave groups of 4 waves share the 64 query rows and split the KEY tiles between them (group s takes tiles s, s +
// NSPLIT, ...).  Measured on MI355X at 768 x 768 x 16 heads (tools/lab/att_bench.py, hipGraph of 200 calls): one group
// 15.5 us -- ~0.9 us per tile whatever the instruction count, i.e. the L2 latency of the one K/V tile a group has in flight;
// two groups 11.6, three 11.0 (a fourth changes nothing: 3 waves per SIMD already saturate VALU + MFMA issue), 2.4 us of
// which are launch + first-tile latency + the merge.  torch's scaled_dot_product_attention + the layout copy: 30.5 us.
// The groups' (m, l, O) triples are merged through LDS at the end.
template <int NSPLIT>
__global__ __launch_bounds__(256 * NSPLIT) void attention_fwd_f16_kernel(AttnArgs a)
{
    __shared__ __attribute__((aligned(16))) _Float16 smem[NSPLIT * ATT_KB * (ATT_KP + ATT_VP)];
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    _Float16* const sk = smem + grp * ATT_KB * (ATT_KP + ATT_VP);
    _Float16* const sv = sk + ATT_KB * ATT_KP;
    const int g = lane >> 4, c = lane & 15;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * ATT_QB + wave * 16;

    const _Float16* kbase = a.k + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh;
    const _Float16* vbase = a.v + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh;

    // Q fragment (B operand of S^T = K Q^T): lane (g, c) holds Q[q0 + c][32 ks + 8 g .. + 7]
    att_f16x8 qf[2];
    {
        const int qrow = min(q0 + c, a.Nq - 1);
        const _Float16* qp = a.q + (int64_t)b * a.q_sb + (int64_t)h * a.q_sh + (int64_t)qrow * a.q_sn + 8 * g;
        qf[0] = *reinterpret_cast<const att_f16x8*>(qp);
        qf[1] = *reinterpret_cast<const att_f16x8*>(qp + 32);
    }

    // staging: the tile is 64 rows x 8 chunks of 16 B; thread t moves chunks t and t + 256 of K and of V
    const int srow = tid >> 3, scol = (tid & 7) * 8; // rows srow and srow + 32
    att_f16x8 pk[2], pv[2];
    auto fetch = [&](int tile0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = min(tile0 + srow + 32 * i, a.Nk - 1); // rows past the end are masked in the scores
            pk[i] = *reinterpret_cast<const att_f16x8*>(kbase + (int64_t)row * a.k_sn + scol);
            pv[i] = *reinterpret_cast<const att_f16x8*>(vbase + (int64_t)row * a.v_sn + scol);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<att_f16x8*>(&sk[(srow + 32 * i) * ATT_KP + scol]) = pk[i];
            *reinterpret_cast<att_f16x8*>(&sv[(srow + 32 * i) * ATT_VP + scol]) = pv[i];
        }
    };

    // LDS addresses of this lane's fragment reads
    const _Float16* kfrag = &sk[c * ATT_KP + 8 * g];                             // + (16 kb) rows, + 32 ks halves
    typedef __attribute__((address_space(3))) att_s16x4 lds_s16x4;
    const _Float16* vfrag = &sv[(4 * g + (c >> 2)) * ATT_VP + 4 * (c & 3)];      // + (32 s + 16 half) rows, + 16 db halves

    att_f32x4 ot[4]; // O^T: ot[db][r] = O[query c][d = 16 db + 4 g + r]
#pragma unroll
    for (int db = 0; db < 4; ++db) ot[db] = att_f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f; // running max (raw score units) and this lane's share of the running sum

    const int n_tiles = (a.Nk + ATT_KB - 1) / ATT_KB;
    const int n_rounds = (n_tiles + NSPLIT - 1) / NSPLIT; // every group runs every round's barriers; a group without a tile idles
    if (grp < n_tiles) { fetch(grp * ATT_KB); stash(); }
    __syncthreads();
    for (int round = 0; round < n_rounds; ++round) {
        const int t = round * NSPLIT + grp;
        const bool have = t < n_tiles, have_next = t + NSPLIT < n_tiles;
        if (have_next) fetch((t + NSPLIT) * ATT_KB);
        if (have) {

        // ---- S^T = K Q^T: st[kb][r] = <K[16 kb + 4 g + r], Q[c]>.  All 8 K fragments are requested before the first MFMA and
        // all 16 V fragments right behind the MFMAs (the sched_barriers keep hipcc from sinking each read in front of its
        // consumer, which serialised every MFMA behind a full LDS latency): the V reads land while the softmax runs.
        att_f16x8 kf[4][2];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kf[kb][ks] = *reinterpret_cast<const att_f16x8*>(kfrag + 16 * kb * ATT_KP + 32 * ks);
        __builtin_amdgcn_sched_barrier(0);
        att_f32x4 st[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            st[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][0], qf[0], att_f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            st[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][1], qf[1], st[kb], 0, 0, 0);
        }
        att_s16x4 vr[4][2][2]; // [db][s][half]: V[key 32 s + 16 half + 4 g + j][d = 16 db + c], j = 0..3
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    vr[db][s][hf] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vfrag + (32 * s + 16 * hf) * ATT_VP + 16 * db));
        __builtin_amdgcn_sched_barrier(0);
        if ((t + 1) * ATT_KB > a.Nk) { // ragged last tile: keys past the end get -inf
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (t * ATT_KB + 16 * kb + 4 * g + r >= a.Nk) st[kb][r] = -INFINITY;
        }

        // ---- online softmax (per query = per lane & 15; the 4 lane groups share the row)
        // 16 scores + the running max through v_max3_f32 (fmaxf() would canonicalise every MFMA result with a v_max x, x first)
        float tmax = att_max3(att_max3(st[0][0], st[0][1], st[0][2]), att_max3(st[0][3], st[1][0], st[1][1]),
                              att_max3(st[1][2], st[1][3], st[2][0]));
        tmax = att_max3(tmax, att_max3(st[2][1], st[2][2], st[2][3]), att_max3(st[3][0], st[3][1], st[3][2]));
        tmax = att_xor_max(att_max3(tmax, st[3][3], m_run));
        const float m_new = tmax; // >= m_run
        const float neg_m = -m_new * a.scale_log2e;
        // the rescale of O and l is skipped while no row of the wave raised its maximum (alpha == 1 exactly)
        const bool grew = __builtin_amdgcn_ballot_w64(m_new > m_run) != 0ull;
        const float alpha = grew ? __builtin_amdgcn_exp2f((m_run - m_new) * a.scale_log2e) : 1.0f;
        m_run = m_new;
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
        att_f16x8 pf[2]; // P^T fragments: pf[s][e] = P[c][key 32 s + 4 g + e] (e < 4), P[c][key 32 s + 16 + 4 g + e - 4] (e >= 4)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(st[kb][r], a.scale_log2e, neg_m));
                ps[r] += p;
                pf[kb >> 1][(kb & 1) * 4 + r] = (_Float16)p;
            }
        if (grew) {
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 4; ++db) ot[db] *= alpha;
        }
        l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);

        // ---- O^T += V^T P^T
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const att_f16x4 lo_h = __builtin_bit_cast(att_f16x4, vr[db][s][0]), hi_h = __builtin_bit_cast(att_f16x4, vr[db][s][1]);
                const att_f16x8 vf = __builtin_shufflevector(lo_h, hi_h, 0, 1, 2, 3, 4, 5, 6, 7);
                ot[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[s], ot[db], 0, 0, 0);
            }

        } // have
        __syncthreads(); // every wave is done with this round's tiles
        if (round + 1 < n_rounds) {
            if (have_next) stash();
            __syncthreads();
        }
    }

    // ---- merge the groups: groups 1.. publish (m, row sum, O^T), group 0 folds them in (the tiles are dead: smem is reused)
    float l_tot = att_xor_sum(l_run);
    if (NSPLIT > 1) {
        float* const mbuf = reinterpret_cast<float*>(smem); // [group - 1][18][256]
        if (grp > 0) {
            float* mb = mbuf + (grp - 1) * 18 * 256 + tid;
            mb[0] = m_run; mb[256] = l_tot;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r = 0; r < 4; ++r) mb[(2 + 4 * db + r) * 256] = ot[db][r];
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int s2 = 1; s2 < NSPLIT; ++s2) {
            const float* mb = mbuf + (s2 - 1) * 18 * 256 + tid;
            const float m2 = mb[0], l2 = mb[256];
            const float m_new = fmaxf(m_run, m2);
            const float a1 = __builtin_amdgcn_exp2f((m_run - m_new) * a.scale_log2e);
            const float a2 = __builtin_amdgcn_exp2f((m2 - m_new) * a.scale_log2e); // a group that never had a tile: m2 = -inf -> 0
            m_run = m_new;
            l_tot = l_tot * a1 + l2 * a2;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r = 0; r < 4; ++r) ot[db][r] = ot[db][r] * a1 + mb[(2 + 4 * db + r) * 256] * a2;
        }
    }

    // ---- normalise and store: out [B, Nq, H, 64]; this lane holds d = 16 db + 4 g + 0..3 of query q0 + c
    const float inv = __builtin_amdgcn_rcpf(l_tot);
    const int qrow = q0 + c;
    if (qrow < a.Nq) {
        _Float16* op = a.out + (((int64_t)b * a.Nq + qrow) * a.H + h) * ATT_D + 4 * g;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            att_f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (_Float16)(ot[db][r] * inv);
            *reinterpret_cast<att_f16x4*>(op + 16 * db) = o;
        }
    }
}

} // namespace adk

// q [B,H,Nq,64], k / v [B,H,Nk,64] given as base pointers + element strides (batch, head, token), last dim dense;
// out [B,Nq,H,64] contiguous (= [B,Nq,H*64], the layout the output projection consumes).  fp16; every stride a multiple
// of 8 elements and every pointer 16-byte aligned (128-bit loads).  scale multiplies q k^T (1/sqrt(64) in the model).
extern "C" int adk_attention_fwd_f16(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk,
                                     const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                     float scale, hipStream_t stream)
{
    if (!q || !k || !v || !out || !q_strides || !k_strides || !v_strides) return ADK_EINVAL;
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || B > 65535 || H > 65535) return ADK_EINVAL;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) return ADK_EINVAL;
    for (int i = 0; i < 3; ++i)
        if ((q_strides[i] | k_strides[i] | v_strides[i]) & 7) return ADK_EINVAL;
    adk::AttnArgs a;
    a.q = static_cast<const _Float16*>(q); a.k = static_cast<const _Float16*>(k); a.v = static_cast<const _Float16*>(v);
    a.out = static_cast<_Float16*>(out);
    a.H = H; a.Nq = Nq; a.Nk = Nk;
    a.q_sb = q_strides[0]; a.q_sh = q_strides[1]; a.q_sn = q_strides[2];
    a.k_sb = k_strides[0]; a.k_sh = k_strides[1]; a.k_sn = k_strides[2];
    a.v_sb = v_strides[0]; a.v_sh = v_strides[1]; a.v_sn = v_strides[2];
    a.scale_log2e = scale * 1.4426950408889634f;
    const dim3 grid((Nq + ATT_QB - 1) / ATT_QB, H, B);
    const int n_tiles = (Nk + ATT_KB - 1) / ATT_KB;
    if (n_tiles >= 6)
        hipLaunchKernelGGL(adk::attention_fwd_f16_kernel<3>, grid, dim3(768), 0, stream, a);
    else if (n_tiles >= 2)
        hipLaunchKernelGGL(adk::attention_fwd_f16_kernel<2>, grid, dim3(512), 0, stream, a);
    else
        hipLaunchKernelGGL(adk::attention_fwd_f16_kernel<1>, grid, dim3(256), 0, stream, a);
    ADK_RETURN_LAST_ERROR();
}
