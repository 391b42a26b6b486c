// Fused level-of-detail cull + covariance MLP + activations for gfx950.
//
// Replaces the torch glue of SceneModel.render between the parameter dictionary and the rasteriser,
// Reconstruct/scene/scene_models/h3dgsv3.py:626-662:
//     cam_centre = inverse(view)[:3,3] ; dist = |xyz - cam_centre|
//     selection  = dist < 2 d_max ; opacity = sigmoid(o) * ((2 d_max - dist)/d_max on (d_max, 2 d_max), else 1)
//     scale_rot  = mlp_cov([global_feat[cls_id], local_feat])     (Linear 32->32, ReLU, Linear 32->7)
//     scaling    = exp(s) * sigmoid(scale_rot[:3]) ; rotation = normalize(rotation * scale_rot[3:])
// which at 1 M Gaussians costs ~12 ms per step on MI355X in stock torch (seven boolean-mask gathers and
// their index_put backwards, hipBLASLt fp32 GEMMs of shape [1M,32]x[32,32] at <2 % of roofline).
//
// Here: ONE streaming kernel per direction, one thread per Gaussian.
//   * No compaction: a Gaussian outside its LoD range gets opacity 0, which the projection kernel
//     culls (opacity < 1/255) -- the rendered result is identical to rendering the masked subset,
//     and the visibility mask is simply radii > 0 over the full set.
//   * The 32->32->7 MLP is evaluated per thread with wave-uniform (scalar-cache) weight reads.
//   * Backward: per-thread matvecs for the feature gradients; the WEIGHT gradients
//     dW1 = sum_g dz_g x_g^T (32x32) and dW2 = sum_g dy_g h_g^T (7x32) are the one true dense
//     contraction of the mapper (K = number of Gaussians) and run on the matrix cores with
//     v_mfma_f32_32x32x2_f32 (exact fp32), each wavefront contracting its 64 Gaussians from LDS;
//     per-workgroup partials go to a scratch slab and a second tiny kernel reduces them, so the
//     result is deterministic and needs no atomics.  global_feat gradients (gather by voxel id in
//     the forward) are scattered with hardware fp32 atomics.
//   * Only Gaussians that received a gradient (visible ones) do any backward work.
#include "adk_common.hpp"
#include "lod_core.hpp"

namespace adk {

// ---- sparse Adam of the five per-Gaussian tensors, applied INSIDE the backward (round 4) ---------------------------------------------
// SparseGaussianAdam.step (Reconstruct/scene/optimizers.py:106-161) runs adamUpdate on xyz / opacity / scaling / rotation / local_feat
// (27 of a Gaussian's 75 floats; the 48 SH colours already take their step inside the projection backward) on the rows with
// visibility = radii > 0.  lod_params_bwd is the LAST kernel that touches those five gradients, and each is produced by exactly
// the thread (or accumulator lane) that can also update the parameter: the gradients are then never written (108 B per Gaussian)
// nor re-read by a separate Adam kernel together with p, m, v (another 108 B + a launch), and adk_adam_update_multi shrinks to the
// voxel features, the mlp and the keyframe's pose.  Same arithmetic as adam.hip:adam_elem, IEEE-unfused (this file is compiled with
// contraction on, hence the pragma): bit-identical to the two-kernel path (tests/test_fused_glue.py).
struct LodAdam {
    const uint8_t* visible;                       // [N] rows SparseGaussianAdam.step would touch (radii > 0)
    float *m_xyz, *v_xyz, *lr_xyz;                // lr_xyz: per-element [N,3], decayed in place on visible rows (optimizers.py:158-161)
    float *m_opacity, *v_opacity, *m_scaling, *v_scaling, *m_rotation, *v_rotation, *m_local, *v_local;
    const float *lr_opacity, *lr_scaling, *lr_rotation, *lr_local;   // 0-dim device tensors
    float lr_decay_xyz, lr_min_xyz, b1, b2, eps;
};

__device__ __forceinline__ void lod_adam_elem(float& p, float g, float& m, float& v, float lr, float b1, float b2, float omb1, float omb2, float eps)
{
#pragma clang fp contract(off)
    m = b1 * m + omb1 * g;
    v = b2 * v + omb2 * g * g;
    const float step = -lr * m / (sqrtf(v) + eps);
    p += step;
}

// (the forward kernel lives in raster_project.hip, next to the fused LoD + projection forward that shares its body: lod_core.hpp)

// ---- backward -------------------------------------------------------------------------------------
// One wavefront per workgroup walks 64-Gaussian chunks.  Every dense contraction of the chunk runs on
// the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32):
//     H  = relu(X W1^T + b1)      [64x32] . [32x32]      (recomputed, not stored by the forward)
//     Y  = H W2^T + b2            [64x32] . [32x7]
//     VH = VY W2 ; VZ = VH * (H>0)[64x7]  . [7x32]
//     VX = VZ W1                  [64x32] . [32x32]      -> v_local_feat rows, v_global_feat atomics
//     dW1 += VZ^T X ; dW2 += VY^T H                      (K = Gaussians; accumulators live across chunks)
// The four weight operands are loaded ONCE per wavefront into VGPR fragments in the B-operand layout;
// activations move between the "lane = Gaussian" elementwise stages and the MFMA operand layout through
// wave-private LDS tiles with a 33-float row pitch (conflict-free for both access patterns).
// MFMA 32x32x2 layout: a = A[i = lane&31][k = lane>>5], b = B[k = lane>>5][j = lane&31],
//                      d[r] = D[i = (r&3) + 8 (r>>2) + 4 (lane>>5)][j = lane&31].
#define LOD_LDW 33
#define LOD_YW 9

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): this wave's LDS traffic has landed (tiles are wave-private)
    __builtin_amdgcn_wave_barrier();
}

// One visible Gaussian's xyz (per-element lr, decayed in place), opacity, scaling and rotation step (lane = Gaussian).  WITH_LOCAL:
// also its 16 local features with a zero gradient (chunks without any active Gaussian skip the matrix stages altogether).
template <bool WITH_LOCAL>
__device__ __forceinline__ void lod_adam_rows(const LodAdam& A, int64_t g, float lr_o, float lr_s, float lr_r, float lr_l, float omb1, float omb2,
                                              float* xyz, float* opacity_raw, float* scaling_raw, float* rotation, float* local_feat,
                                              float go, float gs0, float gs1, float gs2, float4 gq, float gx0, float gx1, float gx2)
{
    const float b1 = A.b1, b2 = A.b2, eps = A.eps;
    {   // opacity [N,1]
        float p = opacity_raw[g], m = A.m_opacity[g], v = A.v_opacity[g];
        lod_adam_elem(p, go, m, v, lr_o, b1, b2, omb1, omb2, eps);
        opacity_raw[g] = p; A.m_opacity[g] = m; A.v_opacity[g] = v;
    }
    const float gs[3] = {gs0, gs1, gs2}, gx[3] = {gx0, gx1, gx2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // scaling [N,3], xyz [N,3] with its per-element learning rate
        float p = scaling_raw[3 * g + k], m = A.m_scaling[3 * g + k], v = A.v_scaling[3 * g + k];
        lod_adam_elem(p, gs[k], m, v, lr_s, b1, b2, omb1, omb2, eps);
        scaling_raw[3 * g + k] = p; A.m_scaling[3 * g + k] = m; A.v_scaling[3 * g + k] = v;
        float px = xyz[3 * g + k], mx = A.m_xyz[3 * g + k], vx = A.v_xyz[3 * g + k];
        const float lr = A.lr_xyz[3 * g + k];
        lod_adam_elem(px, gx[k], mx, vx, lr, b1, b2, omb1, omb2, eps);
        xyz[3 * g + k] = px; A.m_xyz[3 * g + k] = mx; A.v_xyz[3 * g + k] = vx;
        if (A.lr_decay_xyz != 1.0f) A.lr_xyz[3 * g + k] = fmaxf(lr * A.lr_decay_xyz, A.lr_min_xyz);
    }
    {   // rotation [N,4]
        float4 p = reinterpret_cast<float4*>(rotation)[g], m = reinterpret_cast<float4*>(A.m_rotation)[g], v = reinterpret_cast<float4*>(A.v_rotation)[g];
        lod_adam_elem(p.x, gq.x, m.x, v.x, lr_r, b1, b2, omb1, omb2, eps);
        lod_adam_elem(p.y, gq.y, m.y, v.y, lr_r, b1, b2, omb1, omb2, eps);
        lod_adam_elem(p.z, gq.z, m.z, v.z, lr_r, b1, b2, omb1, omb2, eps);
        lod_adam_elem(p.w, gq.w, m.w, v.w, lr_r, b1, b2, omb1, omb2, eps);
        reinterpret_cast<float4*>(rotation)[g] = p; reinterpret_cast<float4*>(A.m_rotation)[g] = m; reinterpret_cast<float4*>(A.v_rotation)[g] = v;
    }
    if (WITH_LOCAL) {
#pragma unroll
        for (int i = 0; i < LOD_L / 4; ++i) {
            float4 p = reinterpret_cast<float4*>(local_feat + g * LOD_L)[i], m = reinterpret_cast<float4*>(A.m_local + g * LOD_L)[i],
                   v = reinterpret_cast<float4*>(A.v_local + g * LOD_L)[i];
            lod_adam_elem(p.x, 0.f, m.x, v.x, lr_l, b1, b2, omb1, omb2, eps);
            lod_adam_elem(p.y, 0.f, m.y, v.y, lr_l, b1, b2, omb1, omb2, eps);
            lod_adam_elem(p.z, 0.f, m.z, v.z, lr_l, b1, b2, omb1, omb2, eps);
            lod_adam_elem(p.w, 0.f, m.w, v.w, lr_l, b1, b2, omb1, omb2, eps);
            reinterpret_cast<float4*>(local_feat + g * LOD_L)[i] = p; reinterpret_cast<float4*>(A.m_local + g * LOD_L)[i] = m;
            reinterpret_cast<float4*>(A.v_local + g * LOD_L)[i] = v;
        }
    }
}

// ADAM: the five per-Gaussian parameter tensors are updated in place (see LodAdam) and their gradients are not written; every read of
// a parameter precedes the write of the same element in the same thread, so the pointers simply lose their __restrict__.
// Tried on the one-wave form in round 4 and REMOVED (evidence under profiles/): requesting the next chunk's stage-0 inputs during this chunk's
// matrix stages (neutral, r04_ab_lod_prefetch.txt); a full software pipeline that lands every input of chunk i + 1 before chunk i's first store
// (60 more live registers push the weight fragments into scratch, whose reloads are vector-memory loads themselves: 16 % slower,
// r04_ab_lod_reorder.txt).  What stayed: the 7-wide products on the 16x16x4 MFMA (SMALL, -9 %) and VX + atomics before dW1 (-1.5 %).
#ifndef ADK_LOD_SMALL
#define ADK_LOD_SMALL 1
#endif
// LATE: the next chunk's stage-0 inputs (incoming gradients, position, d_max, voxel id) are requested right after this chunk's atomics, in
// front of dW1's 32 MFMAs -- behind the atomics in the memory counter's order, so the wait at the top of the next chunk covers both, but
// 2 048 matrix cycles later; two of stage 0's three dependent round trips leave the critical path.
#ifndef ADK_LOD_LATE
#define ADK_LOD_LATE 1
#endif
#ifndef ADK_LOD_BWD_MINWAVES
#define ADK_LOD_BWD_MINWAVES 3   // two-wave form: 3 waves per SIMD = 168 VGPRs, no scratch; 4 (128 VGPRs) spills 143 dwords
#endif
// WAVES (round 4): 1 = one wave walks both 32-row blocks of a 64-Gaussian chunk (the round-2 form: 2 waves per SIMD); 2 = two waves per
// workgroup share the chunk's LDS tiles, each takes ONE 32-row block of every matrix stage and half of the K = 64 contraction of the
// weight gradients: the same 19.5 KB of LDS then carries 4 waves per SIMD and every wave's serial chain of stages is half as long.
template <bool ADAM, int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES == 2 ? ADK_LOD_BWD_MINWAVES : 2, 2 * WAVES))) void lod_params_bwd_kernel(
    int N, const float* xyz, const float* opacity_raw, const float* scaling_raw,
    const float* rotation, const float* local_feat, const float* __restrict__ global_feat,
    const int64_t* __restrict__ cls_id, const float* __restrict__ d_max, const float* __restrict__ W1,
    const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ viewmat, const float* __restrict__ v_opac_eff, const float* __restrict__ v_scale_eff,
    const float* __restrict__ v_quat_eff,
    float* __restrict__ v_xyz_add /* [N,3] += (fade term) */, float* __restrict__ v_opacity_raw,
    float* __restrict__ v_scaling_raw, float* __restrict__ v_rotation, float* __restrict__ v_local_feat,
    float* __restrict__ v_global_feat /* [V,G], zeroed, atomics */, float* __restrict__ partials /* [gridDim.x][LOD_NW] */,
    const LodAdam A)
{
    // Two 64x33 tiles + a 64x9 one = 19.5 KB per workgroup => 8 workgroups per CU: 2 waves per SIMD with one wave per workgroup (which is
    // also what its 256-VGPR budget allows), 4 with two (128 VGPRs each).  TH holds H, later VZ: H is dead once dW2 has been accumulated, and
    // the stage order below is chosen so that this alias is legal.
    __shared__ float TX[64 * LOD_LDW], TH[64 * LOD_LDW], TY[64 * LOD_YW];
    __shared__ int TC[64];             // voxel id per row
    __shared__ unsigned char TV[64];   // ADAM: the chunk's visibility flags, for the accumulator-layout local_feat update
    __shared__ int TAny;               // WAVES == 2: does the chunk have an active Gaussian? (decided by wave 0, read by both)
    float* const TZ = TH;
    const int wave = WAVES == 2 ? (int)(threadIdx.x >> 6) : 0;
    auto tile_sync = [&]() { if (WAVES == 2) __syncthreads(); else lds_fence(); };   // the tiles are wave-private with one wave
    const float omb1 = 1.0f - A.b1, omb2 = 1.0f - A.b2;
    float lr_o = 0.f, lr_s = 0.f, lr_r = 0.f, lr_l = 0.f;
    if (ADAM) { lr_o = A.lr_opacity[0]; lr_s = A.lr_scaling[0]; lr_r = A.lr_rotation[0]; lr_l = A.lr_local[0]; }
    const int lane = threadIdx.x & 63, kk = lane >> 5, rc = lane & 31;

    // weight fragments (B operands), resident for the whole kernel
    float w1t[16], w1n[16], w2t[16], w2n[4];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        w1t[s] = W1[rc * LOD_IN + 2 * s + kk];                              // B[k][j] = W1[j][k]
        w1n[s] = W1[(2 * s + kk) * LOD_IN + rc];                            // B[k][j] = W1[k][j]
        w2t[s] = rc < LOD_OUT ? W2[rc * LOD_HID + 2 * s + kk] : 0.f;        // B[k][o] = W2[o][k]
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int o = 2 * s + kk;
        w2n[s] = o < LOD_OUT ? W2[o * LOD_HID + rc] : 0.f;                  // B[o][i] = W2[o][i]
    }
    const float bias1 = b1[rc], bias2 = rc < LOD_OUT ? b2[rc] : 0.f;
    // Round 4 (one-wave form): the two 7-wide products (Y = H W2^T, dW2 = VY^T H) on v_mfma_f32_16x16x4_f32: padding 7 to 16 instead of to
    // 32 halves their matrix-pipe time (2 x 2048 -> 2 x 1024 cycles of 10 752 per chunk) and their weight / accumulator registers.
    // Layout: a = A[i = lane & 15][k = lane >> 4], b = B[k = lane >> 4][j = lane & 15], d[r] = D[i = 4 (lane >> 4) + r][j = lane & 15].
    constexpr bool SMALL = WAVES == 1 && ADK_LOD_SMALL;
    const int c16 = lane & 15, q16 = lane >> 4;
    float w2y[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) w2y[s] = c16 < LOD_OUT ? W2[c16 * LOD_HID + 4 * s + q16] : 0.f;   // B[k = hid 4 s + q][j = o] = W2[o][hid]
    const float bias2y = c16 < LOD_OUT ? b2[c16] : 0.f;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc2a = {0.f, 0.f, 0.f, 0.f}, acc2b = {0.f, 0.f, 0.f, 0.f};   // dW2[o = 4 q + r][hid = c] and [hid = 16 + c]

    const CamCentre cc = cam_centre_of(viewmat);
    f32x16 acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc2[r] = 0.f; }
    float bs1 = 0.f, bs2 = 0.f;

    const int n_chunks = (N + 63) / 64;
    constexpr bool LATE = WAVES == 1 && !ADAM && ADK_LOD_LATE;
    struct NextIn { float vo, vs0, vs1, vs2, px, py, pz, dm; float4 vq; int64_t cls; };
    auto request = [&](int ch) -> NextIn {
        NextIn in;
        in.vo = in.vs0 = in.vs1 = in.vs2 = in.px = in.py = in.pz = 0.f; in.dm = 1.f; in.vq = make_float4(0.f, 0.f, 0.f, 0.f); in.cls = 0;
        const int64_t gg = (int64_t)ch * 64 + lane;
        if (ch < n_chunks && gg < N) {
            in.vo = v_opac_eff[gg];
            in.vs0 = v_scale_eff[3 * gg]; in.vs1 = v_scale_eff[3 * gg + 1]; in.vs2 = v_scale_eff[3 * gg + 2];
            in.vq = reinterpret_cast<const float4*>(v_quat_eff)[gg];
            in.px = xyz[3 * gg]; in.py = xyz[3 * gg + 1]; in.pz = xyz[3 * gg + 2];
            in.dm = d_max[gg];
            in.cls = cls_id[gg];
        }
        return in;
    };
    NextIn nxt;
    if (LATE) nxt = request(blockIdx.x);
    for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int64_t g = (int64_t)chunk * 64 + lane;
        // ---- stage 0 (lane = Gaussian; wave 0): incoming gradients, LoD geometry, feature gather
        bool active = false;
        float vo = 0.f, vs[3] = {0.f, 0.f, 0.f};
        float4 vq = make_float4(0.f, 0.f, 0.f, 0.f);
        LodGeom L;
        L.alpha_ratio = 1.f; L.inv_dmax = 0.f; L.fading = false; L.selected = false; L.dist = 0.f;
        L.dir[0] = L.dir[1] = L.dir[2] = 0.f;
        bool visible = false;
        bool any_active = false;
        if (wave == 0) {
            if (ADAM) { visible = g < N && A.visible[g] != 0; TV[lane] = visible ? 1 : 0; }
            int64_t cls_next = 0;
            if (LATE) {
                if (g < N) {
                    vo = nxt.vo; vs[0] = nxt.vs0; vs[1] = nxt.vs1; vs[2] = nxt.vs2; vq = nxt.vq; cls_next = nxt.cls;
                    L = lod_geometry_of(nxt.px, nxt.py, nxt.pz, nxt.dm, cc);
                    active = L.selected && (vo != 0.f || vs[0] != 0.f || vs[1] != 0.f || vs[2] != 0.f || vq.x != 0.f || vq.y != 0.f || vq.z != 0.f || vq.w != 0.f);
                }
            } else if (g < N) {
                vo = v_opac_eff[g];
                vs[0] = v_scale_eff[3 * g]; vs[1] = v_scale_eff[3 * g + 1]; vs[2] = v_scale_eff[3 * g + 2];
                vq = reinterpret_cast<const float4*>(v_quat_eff)[g];
                L = lod_geometry(xyz, d_max, g, cc);
                active = L.selected && (vo != 0.f || vs[0] != 0.f || vs[1] != 0.f || vs[2] != 0.f || vq.x != 0.f || vq.y != 0.f || vq.z != 0.f || vq.w != 0.f);
            }
            if (WAVES == 2) asm volatile("" : "=v"(vo), "=v"(vs[0]), "=v"(vs[1]), "=v"(vs[2]), "=v"(vq.x), "=v"(vq.y), "=v"(vq.z), "=v"(vq.w));   // dead until re-read below
            any_active = __ballot(active) != 0ull;
            if (WAVES == 2 && lane == 0) TAny = any_active ? 1 : 0;
            if (any_active) {
                // features straight into the X tile, one float4 at a time (holding all 32 in registers next to the weight fragments spills at 128 VGPRs)
                int c32 = -1;
                if (active) {
                    const int64_t cls = LATE ? cls_next : cls_id[g];
                    c32 = (int)cls;
                    const float4* gf = reinterpret_cast<const float4*>(global_feat + cls * LOD_G);
                    const float4* lf = reinterpret_cast<const float4*>(local_feat + g * LOD_L);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float4 v = i < 4 ? gf[i] : lf[i - 4];
                        float* t = TX + lane * LOD_LDW + 4 * i;
                        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < LOD_IN; ++i) TX[lane * LOD_LDW + i] = 0.f;
                }
                TC[lane] = c32;
            }
        }
        if (WAVES == 2) { __syncthreads(); any_active = TAny != 0; }
        if (!any_active) { // nothing visible in this chunk: zero gradients, no matrix work
            if (wave == 0) {
                if (ADAM) {
                    // visible rows still take their Adam step (zero gradient: the moments decay, the parameter moves by -lr m / (sqrt(v) + eps))
                    if (visible) {
                        lod_adam_rows<true>(A, g, lr_o, lr_s, lr_r, lr_l, omb1, omb2, const_cast<float*>(xyz), const_cast<float*>(opacity_raw),
                                            const_cast<float*>(scaling_raw), const_cast<float*>(rotation), const_cast<float*>(local_feat), 0.f,
                                            0.f, 0.f, 0.f, make_float4(0.f, 0.f, 0.f, 0.f), v_xyz_add[3 * g], v_xyz_add[3 * g + 1], v_xyz_add[3 * g + 2]);
                    }
                } else if (g < N) {
                    v_opacity_raw[g] = 0.f;
                    v_scaling_raw[3 * g] = 0.f; v_scaling_raw[3 * g + 1] = 0.f; v_scaling_raw[3 * g + 2] = 0.f;
                    reinterpret_cast<float4*>(v_rotation)[g] = make_float4(0.f, 0.f, 0.f, 0.f);
                    float4* vl = reinterpret_cast<float4*>(v_local_feat + g * LOD_L);
#pragma unroll
                    for (int i = 0; i < LOD_L / 4; ++i) vl[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (WAVES == 2) __syncthreads();   // TAny / TV are rewritten by the next chunk
            if (LATE) nxt = request(chunk + (int)gridDim.x);
            continue;
        }
        tile_sync();

        // ---- H = relu(X W1^T + b1): two 32-row blocks; keep only the sign mask in registers
        unsigned hmask[2] = {0u, 0u};
#pragma unroll
        for (int rbi = 0; rbi < 2 / WAVES; ++rbi) {
            const int rb = WAVES == 2 ? wave : rbi;
            // the wave's row block enters through BASE pointers and every per-row offset below is a compile-time constant: with rb = wave a
            // run-time value, offsets written as (rb * 32 + row) * pitch became 16 loop-invariant 64-bit registers per use (spilled)
            const float* xa = TX + (rb * 32 + rc) * LOD_LDW + kk;
            float* hw = TH + (rb * 32 + 4 * kk) * LOD_LDW + rc;
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s)
                d = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[2 * s], w1t[s], d, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float h = fmaxf(d[r] + bias1, 0.f);
                hw[((r & 3) + 8 * (r >> 2)) * LOD_LDW] = h;
                hmask[rbi] |= (h > 0.f ? 1u : 0u) << r;
            }
        }
        tile_sync();
        // ---- Y = H W2^T + b2 (columns 0..6 of the 32-wide tile)
        if (SMALL) {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {   // 16 Gaussians per block
                const float* ha = TH + (16 * ib + c16) * LOD_LDW + q16;
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 8; ++s) d = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[4 * s], w2y[s], d, 0, 0, 0);
                if (c16 < 8) {
                    float* yw = TY + (16 * ib + 4 * q16) * LOD_YW + c16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) yw[r * LOD_YW] = d[r] + bias2y;
                }
            }
        } else
#pragma unroll
        for (int rbi = 0; rbi < 2 / WAVES; ++rbi) {
            const int rb = WAVES == 2 ? wave : rbi;
            const float* ha = TH + (rb * 32 + rc) * LOD_LDW + kk;
            float* yw = TY + (rb * 32 + 4 * kk) * LOD_YW + rc;
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s)
                d = __builtin_amdgcn_mfma_f32_32x32x2f32(ha[2 * s], w2t[s], d, 0, 0, 0);
            if (rc < 8) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    yw[((r & 3) + 8 * (r >> 2)) * LOD_YW] = d[r] + bias2;
                }
            }
        }
        tile_sync();
        // ---- elementwise stage (lane = Gaussian; wave 0): activation gradients, vy
        if (wave == 0) {
            float y[LOD_OUT], vy[LOD_OUT];
#pragma unroll
            for (int o = 0; o < LOD_OUT; ++o) { y[o] = TY[lane * LOD_YW + o]; vy[o] = 0.f; }
            float go = 0.f, gs[3] = {0.f, 0.f, 0.f}, gx[3] = {0.f, 0.f, 0.f};
            float4 gq = make_float4(0.f, 0.f, 0.f, 0.f);
            if (WAVES == 2 && active) {
                // the incoming gradients and the LoD geometry are read AGAIN here instead of being carried through the two matrix stages
                // (18 registers that pushed the 168-VGPR budget of the two-wave form into scratch); they are L2-resident
                vo = v_opac_eff[g];
                vs[0] = v_scale_eff[3 * g]; vs[1] = v_scale_eff[3 * g + 1]; vs[2] = v_scale_eff[3 * g + 2];
                vq = reinterpret_cast<const float4*>(v_quat_eff)[g];
                L = lod_geometry(xyz, d_max, g, cc);
            }
            if (active) {
                // opacity = sigmoid(o) * alpha_ratio
                const float so = sigmoidf(opacity_raw[g]);
                go = vo * L.alpha_ratio * so * (1.f - so);
                if (L.fading) { // d(alpha_ratio)/d(xyz) = -dir / d_max
                    const float c = -vo * so * L.inv_dmax;
                    gx[0] = c * L.dir[0]; gx[1] = c * L.dir[1]; gx[2] = c * L.dir[2];
                }
                // scaling = exp(s) * sigmoid(y)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float sraw = scaling_raw[3 * g + k];
                    const float e = __expf(sraw), sy = sigmoidf(y[k]);
                    gs[k] = vs[k] * e * sy;
                    vy[k] = vs[k] * e * sy * (1.f - sy);
                }
                // quat = rotation * y[3:7]
                const float4 q = reinterpret_cast<const float4*>(rotation)[g];
                gq = make_float4(vq.x * y[3], vq.y * y[4], vq.z * y[5], vq.w * y[6]);
                vy[3] = vq.x * q.x; vy[4] = vq.y * q.y; vy[5] = vq.z * q.z; vy[6] = vq.w * q.w;
            }
            if (ADAM) {
                if (visible)
                    lod_adam_rows<false>(A, g, lr_o, lr_s, lr_r, lr_l, omb1, omb2, const_cast<float*>(xyz), const_cast<float*>(opacity_raw),
                                         const_cast<float*>(scaling_raw), const_cast<float*>(rotation), const_cast<float*>(local_feat), go,
                                         gs[0], gs[1], gs[2], gq, v_xyz_add[3 * g] + gx[0], v_xyz_add[3 * g + 1] + gx[1], v_xyz_add[3 * g + 2] + gx[2]);
            } else if (g < N) {
                v_opacity_raw[g] = go;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    v_scaling_raw[3 * g + k] = gs[k];
                    if (gx[k] != 0.f) v_xyz_add[3 * g + k] += gx[k];
                }
                reinterpret_cast<float4*>(v_rotation)[g] = gq;
            }
#pragma unroll
            for (int o = 0; o < LOD_OUT; ++o) TY[lane * LOD_YW + o] = vy[o];
            TY[lane * LOD_YW + 7] = 0.f;
        }
        tile_sync();
        // ---- dW2 += VY^T H (K = the 64 Gaussians of the chunk); last use of H
        const float* ky = TY + ((WAVES == 2 ? 32 * wave : 0) + kk) * LOD_YW + rc;
        const float* kh = TH + ((WAVES == 2 ? 32 * wave : 0) + kk) * LOD_LDW + rc;   // TH = H here, VZ below
        if (SMALL) {
            const float* ky16 = TY + q16 * LOD_YW + c16;
            const float* kh16 = TH + q16 * LOD_LDW + c16;
#pragma unroll 8
            for (int s = 0; s < 16; ++s) {   // k = Gaussian 4 s + q
                const float a = c16 < 8 ? ky16[4 * s * LOD_YW] : 0.f;                      // A[i = o][k = g] = VY[g][o]
                acc2a = __builtin_amdgcn_mfma_f32_16x16x4f32(a, kh16[4 * s * LOD_LDW], acc2a, 0, 0, 0);        // B[k = g][j = hid c]
                acc2b = __builtin_amdgcn_mfma_f32_16x16x4f32(a, kh16[4 * s * LOD_LDW + 16], acc2b, 0, 0, 0);   //           hid 16 + c
                bs2 += a;
            }
        } else
#pragma unroll 8
        for (int si = 0; si < 32 / WAVES; ++si) {
            const float a = rc < 8 ? ky[2 * si * LOD_YW] : 0.f;   // each wave contracts over its own 32 Gaussians
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, kh[2 * si * LOD_LDW], acc2, 0, 0, 0);
            bs2 += a;
        }
        // ---- VZ = (VY W2) * (H > 0), written over H
        f32x16 dz[2];
#pragma unroll
        for (int rbi = 0; rbi < 2 / WAVES; ++rbi) {
            const int rb = WAVES == 2 ? wave : rbi;
#pragma unroll
            for (int r = 0; r < 16; ++r) dz[rbi][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                dz[rbi] = __builtin_amdgcn_mfma_f32_32x32x2f32(TY[(rb * 32 + rc) * LOD_YW + kk + 2 * s], w2n[s], dz[rbi], 0, 0, 0);
        }
        tile_sync(); // every read of H (dW2) has completed
#pragma unroll
        for (int rbi = 0; rbi < 2 / WAVES; ++rbi) {
            const int rb = WAVES == 2 ? wave : rbi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                TZ[(rb * 32 + 4 * kk) * LOD_LDW + rc + ((r & 3) + 8 * (r >> 2)) * LOD_LDW] = ((hmask[rbi] >> r) & 1u) ? dz[rbi][r] : 0.f;
            }
        }
        tile_sync();
        // Round 4: VX and its stores / atomics come BEFORE dW1 (they only share read-only tiles).  Loads and stores share one in-order counter
        // (vmcnt) on this family, and with both kinds pending the compiler has to wait for vmcnt(0): the next chunk's stage-0 loads waited for
        // every global atomic of this chunk's output stage (SQ_WAIT_INST_ANY: 43 % of a wave's life, profiles/r04_pmc_lod_bwd.txt).  With dW1's
        // 32 MFMAs (2 048 cycles) between the atomics and those loads, most of that round trip is spent computing.
        // ---- VX = VZ W1, stored straight from the accumulator layout (lane = feature column, register = Gaussian
        //      row): each store/atomic instruction then covers whole 64 B lines (16 consecutive floats of one
        //      Gaussian per half-wave).  Measured: re-laying VX out one Gaussian per lane (16 B per lane, 64 lines
        //      per instruction) is 3x slower.  Columns 16..31 = local-feature gradients (plain stores, every row),
        //      columns 0..15 scatter into the voxel's global feature (hardware fp32 atomics).
#pragma unroll
        for (int rbi = 0; rbi < 2 / WAVES; ++rbi) {
            const int rb = WAVES == 2 ? wave : rbi;
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s)
                d = __builtin_amdgcn_mfma_f32_32x32x2f32(TZ[(rb * 32 + rc) * LOD_LDW + kk + 2 * s], w1n[s], d, 0, 0, 0);
            if (ADAM && rc >= LOD_G) {
                // the gradient d[r] of local_feat[row][rc - 16] meets its parameter and moments here: same 64 B-line-per-quarter-wave
                // pattern as the gradient store it replaces
                const int64_t off0 = ((int64_t)chunk * 64 + rb * 32 + 4 * kk) * LOD_L + (rc - LOD_G);
                float* pb = const_cast<float*>(local_feat) + off0;
                float* mb = A.m_local + off0;
                float* vb = A.v_local + off0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row0 = (r & 3) + 8 * (r >> 2);
                    if (TV[rb * 32 + 4 * kk + row0]) {   // visible implies row < N
                        float p = pb[row0 * LOD_L], m = mb[row0 * LOD_L], v = vb[row0 * LOD_L];
                        lod_adam_elem(p, d[r], m, v, lr_l, A.b1, A.b2, omb1, omb2, A.eps);
                        pb[row0 * LOD_L] = p; mb[row0 * LOD_L] = m; vb[row0 * LOD_L] = v;
                    }
                }
            } else if (rc >= LOD_G) {
                // one base address per lane, compile-time row offsets (immediate-offset stores)
                float* lbase = v_local_feat + ((int64_t)chunk * 64 + rb * 32 + 4 * kk) * LOD_L + (rc - LOD_G);
                if ((int64_t)chunk * 64 + 64 <= N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) lbase[((r & 3) + 8 * (r >> 2)) * LOD_L] = d[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row0 = (r & 3) + 8 * (r >> 2);
                        if ((int64_t)chunk * 64 + rb * 32 + row0 + 4 * kk < N) lbase[row0 * LOD_L] = d[r];
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // four rows at a time: left alone the scheduler forms all 16 addresses (32 registers) and voxel ids up front, which does
                    // not fit next to the resident weight fragments and accumulators in the two-wave form's 168 VGPRs
                    if (WAVES == 2 && (r & 3) == 0) __builtin_amdgcn_sched_barrier(0);
                    const int c = TC[rb * 32 + 4 * kk + (r & 3) + 8 * (r >> 2)];
                    if (c >= 0 && d[r] != 0.f) unsafeAtomicAdd(v_global_feat + (int64_t)c * LOD_G + rc, d[r]);
                }
            }
        }
        if (LATE) nxt = request(chunk + (int)gridDim.x);   // lands while dW1 runs
        // ---- dW1 += VZ^T X; last use of X
        const float* kx = TX + ((WAVES == 2 ? 32 * wave : 0) + kk) * LOD_LDW + rc;
#pragma unroll 8
        for (int si = 0; si < 32 / WAVES; ++si) {
            const float a = kh[2 * si * LOD_LDW];
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, kx[2 * si * LOD_LDW], acc1, 0, 0, 0);
            bs1 += a;
        }
        tile_sync(); // tiles are rewritten by the next chunk
    }

    // ---- one partial row per workgroup: dW1 | db1 | dW2 | db2
    float* out = partials + ((size_t)blockIdx.x * WAVES + wave) * LOD_NW;   // one partial row per WAVE
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
        out[row * LOD_IN + rc] = acc1[r];
        if (!SMALL && row < LOD_OUT) out[LOD_HID * LOD_IN + LOD_HID + row * LOD_HID + rc] = acc2[r];
    }
    if (SMALL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 4 * q16 + r;
            if (o < LOD_OUT) {
                out[LOD_HID * LOD_IN + LOD_HID + o * LOD_HID + c16] = acc2a[r];
                out[LOD_HID * LOD_IN + LOD_HID + o * LOD_HID + 16 + c16] = acc2b[r];
            }
        }
    }
    const float o1 = __shfl(bs1, (lane + 32) & 63);
    if (lane < 32) out[LOD_HID * LOD_IN + lane] = bs1 + o1;
    if (SMALL) {   // db2[o]: lane (q, c = o) summed its own Gaussians (g = 4 s + q): the four q groups are added here
        float t = bs2 + __shfl(bs2, (lane + 32) & 63);
        t += __shfl(t, (lane + 16) & 63);
        if (lane < LOD_OUT) out[LOD_HID * LOD_IN + LOD_HID + LOD_OUT * LOD_HID + lane] = t;
    } else {
        const float o2 = __shfl(bs2, (lane + 32) & 63);
        if (lane < LOD_OUT) out[LOD_HID * LOD_IN + LOD_HID + LOD_OUT * LOD_HID + lane] = bs2 + o2;
    }
}

// v_w[i] = sum_b partials[b][i]; 32 row-slices per column summed in a fixed order => deterministic
#define LOD_RED_PARTS 32
__global__ __launch_bounds__(1024) void lod_reduce_partials_kernel(const float* __restrict__ partials, int nblocks,
                                                                   float* __restrict__ v_w)
{
    __shared__ float red[LOD_RED_PARTS][32];
    const int col = blockIdx.x * 32 + (threadIdx.x & 31), part = threadIdx.x >> 5;
    float s0 = 0.f, s1 = 0.f;
    if (col < LOD_NW) {
        int b = part;
        for (; b + LOD_RED_PARTS < nblocks; b += 2 * LOD_RED_PARTS) { // two independent chains
            s0 += partials[(size_t)b * LOD_NW + col];
            s1 += partials[(size_t)(b + LOD_RED_PARTS) * LOD_NW + col];
        }
        if (b < nblocks) s0 += partials[(size_t)b * LOD_NW + col];
    }
    red[part][threadIdx.x & 31] = s0 + s1;
    __syncthreads();
    if (threadIdx.x < 32 && col < LOD_NW) {
        float t = 0.f;
#pragma unroll
        for (int p = 0; p < LOD_RED_PARTS; ++p) t += red[p][threadIdx.x];
        v_w[col] = t;
    }
}

// ---- weed_out_gaussians (h3dgsv3.py:942-953): in how many keyframes is each Gaussian inside its LoD range? -------
// The reference loops over every keyframe of the map in Python: a 4x4 torch.inverse, a [N,3] subtraction, a norm, a
// compare and an integer add per keyframe -- five full passes over the Gaussians (and a small LU) times the number
// of keyframes, on every important frame.  Here: one kernel; a workgroup turns up to 256 keyframe poses (6D rotation
// + translation, scene/keyframe.py:150-154) into camera centres in LDS and every thread tests its Gaussian against
// them, so the Gaussians are read once per 256 keyframes.
#define LOD_KF_CHUNK 256
__global__ __launch_bounds__(256) void lod_visible_count_kernel(int N, const float* __restrict__ xyz, const float* __restrict__ d_max,
                                                                int n_kf, const float* __restrict__ r6 /* [n_kf,3,2] */,
                                                                const float* __restrict__ t /* [n_kf,3] */, int* __restrict__ counts)
{
    __shared__ float cen[LOD_KF_CHUNK][3];
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float x = 0.f, y = 0.f, z = 0.f, lim = -1.f;
    if (g < N) { x = xyz[3 * g]; y = xyz[3 * g + 1]; z = xyz[3 * g + 2]; lim = 2.f * d_max[g]; }
    int cnt = 0;
    for (int k0 = 0; k0 < n_kf; k0 += LOD_KF_CHUNK) {
        const int nk = min(LOD_KF_CHUNK, n_kf - k0);
        __syncthreads();
        if ((int)threadIdx.x < nk) { // camera centre -R^T t with R = sixD2mtx(rW2C) (utils.py:223-229)
            const float* r = r6 + (int64_t)(k0 + threadIdx.x) * 6;
            const float* tt = t + (int64_t)(k0 + threadIdx.x) * 3;
            float a1[3] = {r[0], r[2], r[4]}, a2[3] = {r[1], r[3], r[5]};
            const float n1 = sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
            float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
            const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
            float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
            const float nu = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
            float b2[3] = {u[0] / nu, u[1] / nu, u[2] / nu};
            float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
            // R = [b1 b2 b3] (columns); centre = -R^T t: component c = -(column c of R) . t
            cen[threadIdx.x][0] = -(b1[0] * tt[0] + b1[1] * tt[1] + b1[2] * tt[2]);
            cen[threadIdx.x][1] = -(b2[0] * tt[0] + b2[1] * tt[1] + b2[2] * tt[2]);
            cen[threadIdx.x][2] = -(b3[0] * tt[0] + b3[1] * tt[1] + b3[2] * tt[2]);
        }
        __syncthreads();
        for (int k = 0; k < nk; ++k) {
            const float dx = x - cen[k][0], dy = y - cen[k][1], dz = z - cen[k][2];
            cnt += (sqrtf(dx * dx + dy * dy + dz * dz) < lim) ? 1 : 0;
        }
    }
    if (g < N) counts[g] = cnt;
}

} // namespace adk

#define LOD_BWD_MAX_BLOCKS 2048 // 8 resident single-wave workgroups (19.2 KB LDS each) per CU x 256 CUs
extern "C" int64_t adk_lod_params_bwd_workspace_bytes(int N)
{
    if (N < 0) return ADK_EINVAL;
    return (int64_t)LOD_BWD_MAX_BLOCKS * 2 * LOD_NW * (int64_t)sizeof(float);   // one partial row per wave, up to two waves per workgroup
}

// v_mlp: [1287] = dW1 (32x32 row-major) | db1 (32) | dW2 (7x32) | db2 (7).  v_xyz_add is accumulated
// into (pass the xyz gradient of the rasteriser); v_global_feat must be zero-filled by the caller.
extern "C" int adk_lod_params_bwd(int N, const float* xyz, const float* opacity_raw, const float* scaling_raw,
                                  const float* rotation, const float* local_feat, const float* global_feat,
                                  const int64_t* cls_id, const float* d_max, int local_dim, int global_dim, int hidden_dim,
                                  const float* W1, const float* b1, const float* W2, const float* b2, const float* viewmat,
                                  const float* v_opac_eff, const float* v_scale_eff, const float* v_quat_eff,
                                  float* v_xyz_add, float* v_opacity_raw, float* v_scaling_raw, float* v_rotation,
                                  float* v_local_feat, float* v_global_feat, float* v_mlp, void* workspace,
                                  int64_t workspace_bytes, hipStream_t stream)
{
    if (N < 0) return ADK_EINVAL;
    if (local_dim != LOD_L || global_dim != LOD_G || hidden_dim != LOD_HID) return ADK_EUNSUPPORTED;
    if (!v_mlp) return ADK_EINVAL;
    if (N == 0) return adk::clear_bytes(v_mlp, LOD_NW * sizeof(float), stream);
    if (!xyz || !opacity_raw || !scaling_raw || !rotation || !local_feat || !global_feat || !cls_id || !d_max || !W1 || !b1 || !W2 || !b2 || !viewmat) return ADK_EINVAL;
    if (!v_opac_eff || !v_scale_eff || !v_quat_eff || !v_xyz_add || !v_opacity_raw || !v_scaling_raw || !v_rotation || !v_local_feat || !v_global_feat || !workspace) return ADK_EINVAL;
    if (workspace_bytes < adk_lod_params_bwd_workspace_bytes(N)) return ADK_EWORKSPACE;
    if (((uintptr_t)rotation | (uintptr_t)local_feat | (uintptr_t)global_feat | (uintptr_t)v_quat_eff | (uintptr_t)v_rotation | (uintptr_t)v_local_feat) & 15) return ADK_EINVAL;
    int nb = (int)adk::ceil_div(N, 64);
    if (nb > LOD_BWD_MAX_BLOCKS) nb = LOD_BWD_MAX_BLOCKS;
    adk::LodAdam none{};
    // ADK_LOD_BWD_WAVES = 1 (default) | 2 (read per launch: the lab flips it in-process): waves per 64-Gaussian chunk, see the kernel
    const char* we = getenv("ADK_LOD_BWD_WAVES");
    const int waves = (we && we[0] == '2') ? 2 : 1;   // default: one wave per chunk (the two-wave form measured 60 % slower, profiles/r04_ab_lod_waves.txt)
#define ADK_LOD_ARGS N, xyz, opacity_raw, scaling_raw, rotation, local_feat, global_feat, cls_id, d_max, W1, b1, W2, b2, viewmat, v_opac_eff, v_scale_eff, \
                     v_quat_eff, v_xyz_add, v_opacity_raw, v_scaling_raw, v_rotation, v_local_feat, v_global_feat, (float*)workspace, none
    if (waves == 2) hipLaunchKernelGGL((adk::lod_params_bwd_kernel<false, 2>), dim3(nb), dim3(128), 0, stream, ADK_LOD_ARGS);
    else hipLaunchKernelGGL((adk::lod_params_bwd_kernel<false, 1>), dim3(nb), dim3(64), 0, stream, ADK_LOD_ARGS);
#undef ADK_LOD_ARGS
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(adk::lod_reduce_partials_kernel, dim3((LOD_NW + 31) / 32), dim3(1024), 0, stream, (const float*)workspace, nb * waves, v_mlp);
    ADK_RETURN_LAST_ERROR();
}

// adk_lod_params_bwd with the sparse-Adam step of xyz / opacity / scaling / rotation / local_feat applied INSIDE the kernel
// (SparseGaussianAdam.step, Reconstruct/scene/optimizers.py:136-161, for those five keys) on the rows with visible[g] != 0:
// the five parameter tensors (xyz .. local_feat, the same pointers the gradient is evaluated at) and their moments are updated in
// place, their gradients are NOT written (there are no v_opacity_raw / v_scaling_raw / v_rotation / v_local_feat arguments), xyz's
// per-element learning rate lr_xyz [N,3] is decayed in place on the visible rows: lr = max(lr * lr_decay_xyz, lr_min_xyz).
// v_xyz (the rasteriser's gradient of the means) is only read.  lr_opacity .. lr_local: 0-dim device tensors.  v_global_feat
// (zero-filled by the caller, atomics) and v_mlp are produced as by adk_lod_params_bwd: their Adam stays with the caller.
extern "C" int adk_lod_params_bwd_adam(int N, float* xyz, float* opacity_raw, float* scaling_raw, float* rotation, float* local_feat,
                                       const float* global_feat, const int64_t* cls_id, const float* d_max, int local_dim, int global_dim,
                                       int hidden_dim, const float* W1, const float* b1, const float* W2, const float* b2, const float* viewmat,
                                       const float* v_opac_eff, const float* v_scale_eff, const float* v_quat_eff, const float* v_xyz,
                                       float* v_global_feat, float* v_mlp, void* workspace, int64_t workspace_bytes,
                                       const uint8_t* visible, float* m_xyz, float* v2_xyz, float* lr_xyz, float lr_decay_xyz, float lr_min_xyz,
                                       float* m_opacity, float* v2_opacity, const float* lr_opacity, float* m_scaling, float* v2_scaling,
                                       const float* lr_scaling, float* m_rotation, float* v2_rotation, const float* lr_rotation,
                                       float* m_local, float* v2_local, const float* lr_local, float beta1, float beta2, float eps,
                                       hipStream_t stream)
{
    if (N < 0) return ADK_EINVAL;
    if (local_dim != LOD_L || global_dim != LOD_G || hidden_dim != LOD_HID) return ADK_EUNSUPPORTED;
    if (!v_mlp) return ADK_EINVAL;
    if (N == 0) return adk::clear_bytes(v_mlp, LOD_NW * sizeof(float), stream);
    if (!xyz || !opacity_raw || !scaling_raw || !rotation || !local_feat || !global_feat || !cls_id || !d_max || !W1 || !b1 || !W2 || !b2 || !viewmat) return ADK_EINVAL;
    if (!v_opac_eff || !v_scale_eff || !v_quat_eff || !v_xyz || !v_global_feat || !workspace) return ADK_EINVAL;
    if (!visible || !m_xyz || !v2_xyz || !lr_xyz || !m_opacity || !v2_opacity || !lr_opacity || !m_scaling || !v2_scaling || !lr_scaling ||
        !m_rotation || !v2_rotation || !lr_rotation || !m_local || !v2_local || !lr_local) return ADK_EINVAL;
    if (workspace_bytes < adk_lod_params_bwd_workspace_bytes(N)) return ADK_EWORKSPACE;
    if (((uintptr_t)rotation | (uintptr_t)local_feat | (uintptr_t)global_feat | (uintptr_t)v_quat_eff | (uintptr_t)m_rotation | (uintptr_t)v2_rotation |
         (uintptr_t)m_local | (uintptr_t)v2_local) & 15) return ADK_EINVAL;
    int nb = (int)adk::ceil_div(N, 64);
    if (nb > LOD_BWD_MAX_BLOCKS) nb = LOD_BWD_MAX_BLOCKS;
    adk::LodAdam A;
    A.visible = visible;
    A.m_xyz = m_xyz; A.v_xyz = v2_xyz; A.lr_xyz = lr_xyz; A.lr_decay_xyz = lr_decay_xyz; A.lr_min_xyz = lr_min_xyz;
    A.m_opacity = m_opacity; A.v_opacity = v2_opacity; A.lr_opacity = lr_opacity;
    A.m_scaling = m_scaling; A.v_scaling = v2_scaling; A.lr_scaling = lr_scaling;
    A.m_rotation = m_rotation; A.v_rotation = v2_rotation; A.lr_rotation = lr_rotation;
    A.m_local = m_local; A.v_local = v2_local; A.lr_local = lr_local;
    A.b1 = beta1; A.b2 = beta2; A.eps = eps;
    const char* we = getenv("ADK_LOD_BWD_WAVES");
    const int waves = (we && we[0] == '2') ? 2 : 1;   // default: one wave per chunk (the two-wave form measured 60 % slower, profiles/r04_ab_lod_waves.txt)
#define ADK_LOD_ARGS N, xyz, opacity_raw, scaling_raw, rotation, local_feat, global_feat, cls_id, d_max, W1, b1, W2, b2, viewmat, v_opac_eff, v_scale_eff, \
                     v_quat_eff, const_cast<float*>(v_xyz), (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, v_global_feat, (float*)workspace, A
    if (waves == 2) hipLaunchKernelGGL((adk::lod_params_bwd_kernel<true, 2>), dim3(nb), dim3(128), 0, stream, ADK_LOD_ARGS);
    else hipLaunchKernelGGL((adk::lod_params_bwd_kernel<true, 1>), dim3(nb), dim3(64), 0, stream, ADK_LOD_ARGS);
#undef ADK_LOD_ARGS
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(adk::lod_reduce_partials_kernel, dim3((LOD_NW + 31) / 32), dim3(1024), 0, stream, (const float*)workspace, nb * waves, v_mlp);
    ADK_RETURN_LAST_ERROR();
}

// counts[g] = number of keyframes k with |xyz[g] - centre_k| < 2 d_max[g], centre_k = -R_k^T t_k,
// R_k = sixD2mtx(r6[k]) -- the loop of weed_out_gaussians (h3dgsv3.py:943-950) for all keyframes at once.
extern "C" int adk_lod_visible_count(int N, const float* xyz, const float* d_max, int n_keyframes, const float* r6,
                                     const float* t, int32_t* counts, hipStream_t stream)
{
    if (N < 0 || n_keyframes < 0) return ADK_EINVAL;
    if (N == 0) return 0;
    if (!xyz || !d_max || !counts || (n_keyframes > 0 && (!r6 || !t))) return ADK_EINVAL;
    hipLaunchKernelGGL(adk::lod_visible_count_kernel, dim3((unsigned)adk::ceil_div(N, 256)), dim3(256), 0, stream, N, xyz, d_max,
                       n_keyframes, r6, t, counts);
    ADK_RETURN_LAST_ERROR();
}
