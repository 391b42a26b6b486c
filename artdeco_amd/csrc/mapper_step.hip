// The mapper's optimisation step -- forward, loss, backward -- enqueued by ONE host call (include/artdeco_hip.h: adk_mapper_step).
//
// Mirrors the order of SceneModel.optimization_step between zero_grad and the optimiser steps (Reconstruct/scene/scene_models/
// h3dgsv3.py:418-455) on the stages this library already exports one by one; nothing here computes, it sequences (three tiny launches
// ride in a neighbour's: the two zero fills in the pose's, the visibility masks in the projection's, the pose backward in the camera
// gradient's finalisation; and the LoD forward and the projection forward are one kernel -- adk_internal.hpp).  The point is the
// step's single host wait (the intersection count sizes the tile lists, as upstream's isect_tiles -> n_isects read does): with the
// stages driven from Python the host needs ~0.1 ms between that wait and the forward rasteriser's launch on a fast box and several
// times that on a slow one (DESIGN finding 34), during which the GPU has only the pre-launched scatter to run.  Here the wait, the
// sort and the rasteriser are consecutive statements of one function, and the host's total per step falls from ~0.55 ms to the
// caller's bookkeeping around one call.
#include <hip/hip_runtime.h>

#include <chrono>
#include <mutex>
#include <vector>

#include "adk_common.hpp"
#include "adk_internal.hpp"
#include "artdeco_hip.h"

namespace adk {

// 16 B per lane zero fill (the gradient records the raster backward accumulates into, the voxel-feature gradient the LoD backward scatters
// into): a kernel, not hipMemsetAsync (see clear_bytes).  nbytes is a multiple of 16 and p 16 B aligned in both uses; the tail goes through
// clear_bytes otherwise.
__global__ __launch_bounds__(256) void step_zero_kernel(float4* __restrict__ p, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
static int step_zero(void* p, int64_t nbytes, hipStream_t stream) {
    if (nbytes <= 0) return 0;
    if ((reinterpret_cast<uintptr_t>(p) & 15) != 0 || (nbytes & 15) != 0) return clear_bytes(p, nbytes, stream);
    const int64_t n16 = nbytes >> 4;
    hipLaunchKernelGGL(step_zero_kernel, dim3(stream_grid(n16, 256)), dim3(256), 0, stream, static_cast<float4*>(p), n16);
    return (int)hipGetLastError();
}

// The pinned landing place of the two binning statistics and the event the host waits on: one per host thread and device (two threads
// stepping two scenes must not share it; a scene's steps are issued by one thread).
struct CountSlot {
    int device = -1;
    int64_t* host = nullptr;
    hipEvent_t ready = nullptr;
};
static CountSlot* count_slot() {
    thread_local std::vector<CountSlot> slots;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (auto& s : slots)
        if (s.device == dev) return &s;
    CountSlot s;
    s.device = dev;
    if (hipHostMalloc(reinterpret_cast<void**>(&s.host), 2 * sizeof(int64_t), hipHostMallocDefault) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(s.host); return nullptr; }
    slots.push_back(s);
    return &slots.back();
}

// Stage timing on request (bench.py prices the roofline kernel with events on the stream it runs on).
struct StageEvents { int stage; hipEvent_t a, b; };
static std::mutex g_ev_lock;
static std::vector<StageEvents> g_events;
// A host that installs a timer and never folds it must not grow this without bound: beyond the cap the oldest pairs are destroyed.
static constexpr size_t kMaxPendingStageEvents = 16384;
// The pairs of the call in progress on this thread: published to g_events only when the call completes (a call that returns
// ADK_STEP_ECAPACITY / ADK_STEP_EROUTE is repeated or replaced, and its forward stages must not be counted twice).
static thread_local std::vector<StageEvents> t_call_events;
static void drop_call_events() {
    for (auto& e : t_call_events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    t_call_events.clear();
}
static void publish_call_events() {
    if (t_call_events.empty()) return;
    std::lock_guard<std::mutex> g(g_ev_lock);
    for (auto& e : t_call_events) g_events.push_back(e);
    t_call_events.clear();
    if (g_events.size() > kMaxPendingStageEvents) {
        const size_t drop = g_events.size() - kMaxPendingStageEvents;
        for (size_t i = 0; i < drop; ++i) { (void)hipEventDestroy(g_events[i].a); (void)hipEventDestroy(g_events[i].b); }
        g_events.erase(g_events.begin(), g_events.begin() + (long)drop);
    }
}

// ---- the scaling regulariser of the step (h3dgsv3.py:443-449: loss += scaling_reg_factor * scale.prod(dim=1).mean(), `scale` = the
// post-mlp_cov scales of the LoD-SELECTED Gaussians, :660,699).  run.sh trains with 0; with a non-zero factor the one-call step used to
// hand the step back to the per-stage chain (VERDICT r05 missing-4).  Three small launches, only when the factor is not 0:
//   sum     ws[0] += sum over selected rows of s0 s1 s2 (fp32 product as torch forms it, summed in double), ws[1] += their count
//   finish  loss += factor * ws[0] / max(ws[1], 1);  ws[2] = factor / max(ws[1], 1);  ws[0] = ws[1] = 0 for the next step
//   bwd     v_scales[g] += unit_grad * ws[2] * (s1 s2, s0 s2, s0 s1)   -- after the projection backward WROTE v_scales, before lod_params_bwd reads it
__global__ __launch_bounds__(256) void scale_reg_sum_kernel(int N, const float* __restrict__ scale, const uint8_t* __restrict__ sel, double* __restrict__ ws) {
    __shared__ double red[2][4];
    double p = 0.0, c = 0.0;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < N; g += (int64_t)gridDim.x * 256)
        if (sel[g]) { p += (double)((scale[3 * g] * scale[3 * g + 1]) * scale[3 * g + 2]); c += 1.0; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { p += __shfl_xor(p, o, 64); c += __shfl_xor(c, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = p; red[1][threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double P = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), C = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        if (C != 0.0) { unsafeAtomicAdd(ws, P); unsafeAtomicAdd(ws + 1, C); }
    }
}
__global__ void scale_reg_finish_kernel(double* __restrict__ ws, float factor, float* __restrict__ loss) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double cnt = ws[1] > 1.0 ? ws[1] : 1.0;
    loss[0] = (float)((double)loss[0] + (double)factor * ws[0] / cnt);
    ws[2] = (double)factor / cnt;
    ws[0] = 0.0; ws[1] = 0.0;
}
__global__ __launch_bounds__(256) void scale_reg_bwd_kernel(int N, const float* __restrict__ scale, const uint8_t* __restrict__ sel, const double* __restrict__ ws,
                                                            const float* __restrict__ unit_grad, float* __restrict__ v_scales) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= N || !sel[g]) return;
    const float c = (float)ws[2] * unit_grad[0];
    const float s0 = scale[3 * g], s1 = scale[3 * g + 1], s2 = scale[3 * g + 2];
    v_scales[3 * g] += c * (s1 * s2); v_scales[3 * g + 1] += c * (s0 * s2); v_scales[3 * g + 2] += c * (s0 * s1);
}

struct StageScope {
    hipStream_t stream;
    hipEvent_t a = nullptr, b = nullptr;
    int stage;
    StageScope(const AdkMapperStepArgs* A, int stage_, hipStream_t s) : stream(s), stage(stage_) {
        if ((A->time_mask >> stage_) & 1u) {
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
            (void)hipEventRecord(a, stream);
        }
    }
    ~StageScope() {
        if (a == nullptr) return;
        (void)hipEventRecord(b, stream);
        t_call_events.push_back({stage, a, b});
    }
};

} // namespace adk

#ifndef ADK_STEP_FUSE_FWD
#define ADK_STEP_FUSE_FWD 1
#endif
#define ADK_STAGE_ENUM(n) STAGE_##n,
enum { ADK_MAPPER_STAGES(ADK_STAGE_ENUM) STAGE_COUNT };
static_assert(STAGE_COUNT == ADK_MAPPER_N_STAGES, "ADK_MAPPER_N_STAGES out of date");

extern "C" int64_t adk_mapper_step_args_bytes(void) { return (int64_t)sizeof(AdkMapperStepArgs); }

extern "C" int64_t adk_mapper_step_timings(double* sum_ms, double* min_ms, int64_t* count)
{
    std::vector<adk::StageEvents> evs;
    {
        std::lock_guard<std::mutex> g(adk::g_ev_lock);
        evs.swap(adk::g_events);
    }
    int64_t seen[ADK_MAPPER_N_STAGES];
    for (int s = 0; s < ADK_MAPPER_N_STAGES; ++s) {
        if (sum_ms) sum_ms[s] = 0.0;
        if (min_ms) min_ms[s] = 0.0;
        if (count) count[s] = 0;
        seen[s] = 0;
    }
    int64_t n = 0;
    for (auto& e : evs) {
        float ms = 0.f;
        if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess && e.stage >= 0 &&
            e.stage < ADK_MAPPER_N_STAGES) {
            if (sum_ms) sum_ms[e.stage] += ms;
            if (min_ms) min_ms[e.stage] = seen[e.stage] > 0 ? (ms < min_ms[e.stage] ? ms : min_ms[e.stage]) : ms;
            seen[e.stage] += 1;
            if (count) count[e.stage] += 1;
            ++n;
        }
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    return n;
}

#define ADK_STEP_TRY(stage_id, call)                      \
    do {                                                  \
        const int rc_ = (call);                           \
        if (rc_ != 0) { out->stage = (stage_id); return rc_; } \
    } while (0)

static int mapper_step_impl(const AdkMapperStepArgs* A, AdkMapperStepOut* out, adk_stream_t stream);

extern "C" int adk_mapper_step(const AdkMapperStepArgs* A, AdkMapperStepOut* out, adk_stream_t stream)
{
    adk::drop_call_events();
    const int rc = mapper_step_impl(A, out, stream);
    if (rc == ADK_OK) adk::publish_call_events();
    else adk::drop_call_events();      // an aborted attempt's stage pairs are not part of any step's timing
    return rc;
}

static int mapper_step_impl(const AdkMapperStepArgs* A, AdkMapperStepOut* out, adk_stream_t stream)
{
    if (A == nullptr || out == nullptr) return ADK_EINVAL;
    out->n_isects = 0; out->max_tile = 0; out->stage = -1; out->reserved = 0; out->wait_ns = 0;
    const int N = A->N, W = A->width, H = A->height, tpw = A->tile_px_w, tph = A->tile_px_h;
    if (N <= 0 || A->V <= 0 || W <= 0 || H <= 0 || A->isect_capacity <= 0) return ADK_EINVAL;
    if (!adk_bin_local_supported_t(W, H, tpw, tph)) return ADK_STEP_EROUTE;
    if (!A->color_adam && (A->v_dc == nullptr || A->v_rest == nullptr)) return ADK_EINVAL;
    auto f = [](void* p) { return static_cast<float*>(p); };
    auto cf = [](void* p) { return static_cast<const float*>(p); };
    const int64_t HW = (int64_t)W * H;
    adk::CountSlot* slot = adk::count_slot();
    if (slot == nullptr) return ADK_EINVAL;
    // 256 B-aligned view of the binning table, as the per-stage binding forms it
    const uintptr_t tb0 = reinterpret_cast<uintptr_t>(A->bin_table), tb = (tb0 + 255) & ~(uintptr_t)255;
    void* const table = reinterpret_cast<void*>(tb);
    const int64_t table_bytes = A->bin_table_bytes - (int64_t)(tb - tb0);

    // ---- forward: pose -> LoD / mlp_cov -> projection -> binning -> rasteriser (h3dgsv3.py:626-680)
    // the pose's launch also zeroes the two buffers later stages set bits in / scatter into (the voxel visibility mask: the projection;
    // the voxel-feature gradient: the LoD backward), instead of a ~4 us launch each
    ADK_STEP_TRY(STAGE_lod_params_fwd, adk::pose6d_fwd_clear(cf(A->r6), cf(A->t), f(A->viewmat), A->gvis, (int64_t)A->V, A->v_global_feat,
                                                             (int64_t)A->V * 16 * sizeof(float), stream));
#if ADK_STEP_FUSE_FWD
    {
        // LoD / mlp_cov forward and projection forward as ONE kernel (adk_internal.hpp): timed as the projection's stage
        adk::StageScope ts(A, STAGE_project_fwd, stream);
        const adk::ProjectMasks masks = {static_cast<const int64_t*>(A->cls_id), (int64_t)A->V, static_cast<uint8_t*>(A->vis),
                                         static_cast<uint8_t*>(A->gvis)};
        ADK_STEP_TRY(STAGE_project_fwd,
                     adk::lod_project_fwd_launch(N, cf(A->xyz), cf(A->opacity_raw), cf(A->scaling_raw), cf(A->rotation), cf(A->local_feat),
                                                 cf(A->global_feat), static_cast<const int64_t*>(A->cls_id), cf(A->d_max), cf(A->W1), cf(A->b1),
                                                 cf(A->W2), cf(A->b2), f(A->opac), f(A->scale), f(A->quat), static_cast<uint8_t*>(A->sel),
                                                 cf(A->f_dc), cf(A->f_rest), A->sh_K, A->sh_degree, cf(A->viewmat), cf(A->Kmat), W, H, A->eps2d,
                                                 A->near_plane, A->far_plane, A->radius_clip, f(A->rec), static_cast<int32_t*>(A->radii),
                                                 static_cast<uint32_t*>(A->depth_keys), static_cast<uint32_t*>(A->gauss_ids),
                                                 static_cast<int32_t*>(A->tiles_per_gauss), &masks, stream));
    }
#else
    {
        adk::StageScope ts(A, STAGE_lod_params_fwd, stream);
        ADK_STEP_TRY(STAGE_lod_params_fwd,
                     adk_lod_params_fwd(N, cf(A->xyz), cf(A->opacity_raw), cf(A->scaling_raw), cf(A->rotation), cf(A->local_feat),
                                        cf(A->global_feat), static_cast<const int64_t*>(A->cls_id), cf(A->d_max), 16, 16, 32, cf(A->W1),
                                        cf(A->b1), cf(A->W2), cf(A->b2), cf(A->viewmat), f(A->opac), f(A->scale), f(A->quat),
                                        static_cast<uint8_t*>(A->sel), stream));
    }
    {
        adk::StageScope ts(A, STAGE_project_fwd, stream);
        // the visibility masks of h3dgsv3.py:695-698 come out of the projection itself (adk_visibility_masks' rule on the radii it forms)
        const adk::ProjectMasks masks = {static_cast<const int64_t*>(A->cls_id), (int64_t)A->V, static_cast<uint8_t*>(A->vis),
                                         static_cast<uint8_t*>(A->gvis)};
        ADK_STEP_TRY(STAGE_project_fwd,
                     adk::project_fwd_launch(N, cf(A->xyz), cf(A->quat), cf(A->scale), cf(A->opac), cf(A->f_dc), cf(A->f_rest), A->sh_K,
                                             A->sh_degree, 0 /* SH */, cf(A->viewmat), cf(A->Kmat), W, H, A->eps2d, A->near_plane, A->far_plane,
                                             A->radius_clip, 0, f(A->rec), static_cast<int32_t*>(A->radii), static_cast<uint32_t*>(A->depth_keys),
                                             static_cast<uint32_t*>(A->gauss_ids), static_cast<int32_t*>(A->tiles_per_gauss), &masks, stream));
    }
#endif
    {
        adk::StageScope ts(A, STAGE_bin_count, stream);
        ADK_STEP_TRY(STAGE_bin_count,
                     adk_bin_local_count_t(N, static_cast<const int32_t*>(A->tiles_per_gauss), cf(A->rec), W, H, tpw, tph,
                                           static_cast<int32_t*>(A->offsets), static_cast<int64_t*>(A->bin_stats), table, table_bytes, stream));
    }
    // the count travels to pinned memory while the scatter -- which only needs a CAPACITY -- already runs
    if (hipMemcpyAsync(slot->host, A->bin_stats, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipEventRecord(slot->ready, stream) != hipSuccess) { out->stage = STAGE_bin_count; return ADK_EINVAL; }
    {
        adk::StageScope ts(A, STAGE_bin_scatter, stream);
        ADK_STEP_TRY(STAGE_bin_scatter,
                     adk_bin_local_scatter_t(N, A->isect_capacity, static_cast<const uint32_t*>(A->depth_keys),
                                             static_cast<const int32_t*>(A->tiles_per_gauss), cf(A->rec), W, H, tpw, tph,
                                             static_cast<const int32_t*>(A->offsets), table, table_bytes, A->pairs, stream));
    }
    const auto wait_t0 = std::chrono::steady_clock::now();
    if (hipEventSynchronize(slot->ready) != hipSuccess) { out->stage = STAGE_bin_count; return ADK_EINVAL; }   // the one host wait of the step
    out->wait_ns = (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - wait_t0).count();
    const int64_t n_isects = slot->host[0], max_tile = slot->host[1];
    out->n_isects = n_isects; out->max_tile = max_tile;
    // a tile list above 8 192 entries takes the long-list sort (round 5), which needs the second key buffer; only a list beyond ITS limit
    // (4 M entries on one tile) still leaves for the global route.  Nothing of the caller's has been modified yet.
    if (max_tile > 8192 && (A->pairs2 == nullptr || max_tile > adk_bin_local_sort_long_max())) return ADK_STEP_EROUTE;
    if (n_isects > A->isect_capacity) return ADK_STEP_ECAPACITY;
    {
        adk::StageScope ts(A, STAGE_bin_sort, stream);
        ADK_STEP_TRY(STAGE_bin_sort, adk_bin_local_sort_long_t(n_isects, max_tile, W, H, tpw, tph, static_cast<const int32_t*>(A->offsets), A->pairs,
                                                               A->pairs2, A->pairs2 ? A->isect_capacity * 8 : 0, static_cast<int32_t*>(A->flatten_ids),
                                                               nullptr, stream));
    }
    {
        adk::StageScope ts(A, STAGE_raster_fwd, stream);
        ADK_STEP_TRY(STAGE_raster_fwd,
                     adk_raster_fwd_t(W, H, tpw, tph, cf(A->rec), static_cast<const int32_t*>(A->flatten_ids), static_cast<const int32_t*>(A->offsets),
                                      n_isects, nullptr, f(A->render_colors), f(A->render_alphas), f(A->final_T), static_cast<int32_t*>(A->last_ids),
                                      nullptr, stream));
    }

    // ---- loss (h3dgsv3.py:690-694, 611-614, 430-448)
    const float* gt_used = A->mask_outliers ? cf(A->gt_used) : cf(A->gt);
    float* const dm0 = f(A->dm);
    float* const dm1 = dm0 + 3 * HW;
    float* const dm2 = dm0 + 6 * HW;
    {
        adk::StageScope ts(A, STAGE_photometric_fwd, stream);
        ADK_STEP_TRY(STAGE_photometric_fwd,
                     adk_photometric_fwd(W, H, cf(A->render_colors), cf(A->render_alphas), cf(A->bg), cf(A->exposure), cf(A->gt), cf(A->mono),
                                         cf(A->rdk), A->mask_outliers, f(A->image), A->mask_outliers ? f(A->gt_used) : nullptr, f(A->invdepth),
                                         A->photo_ws, A->photo_ws_bytes, stream));
    }
    {
        adk::StageScope ts(A, STAGE_ssim_fwd, stream);
        ADK_STEP_TRY(STAGE_ssim_fwd, adk_fused_ssim_fwd_sums(cf(A->image), gt_used, 1, 3, H, W, (float)(0.01 * 0.01), (float)(0.03 * 0.03), nullptr, dm0, dm1, dm2,
                                                             f(A->ssim_sums), stream));
    }
    {
        adk::StageScope ts(A, STAGE_photometric_loss, stream);
        ADK_STEP_TRY(STAGE_photometric_loss, adk_photometric_loss_sums(W, H, cf(A->ssim_sums), A->n_ssim_sums, A->lambda_dssim, A->depth_weight,
                                                                       A->photo_ws, A->photo_ws_bytes, f(A->parts), f(A->loss), stream));
    }

    const bool scale_reg = A->scaling_reg_factor != 0.f;
    if (scale_reg) { // h3dgsv3.py:443-449
        if (!A->reg_ws) { out->stage = STAGE_photometric_loss; return ADK_EINVAL; }
        hipLaunchKernelGGL(adk::scale_reg_sum_kernel, dim3(adk::stream_grid(N, 256)), dim3(256), 0, stream, N, cf(A->scale),
                           static_cast<const uint8_t*>(A->sel), static_cast<double*>(A->reg_ws));
        hipLaunchKernelGGL(adk::scale_reg_finish_kernel, dim3(1), dim3(64), 0, stream, static_cast<double*>(A->reg_ws), A->scaling_reg_factor, f(A->loss));
        ADK_STEP_TRY(STAGE_photometric_loss, (int)hipGetLastError());
    }

    // ---- backward, in the order the autograd engine runs the nodes
    {
        adk::StageScope ts(A, STAGE_ssim_bwd, stream);
        ADK_STEP_TRY(STAGE_ssim_bwd, adk_fused_ssim_bwd(cf(A->image), gt_used, nullptr, A->ssim_grad_scale, dm0, dm1, dm2, 1, 3, H, W,
                                                        f(A->v_img), stream));
    }
    {
        adk::StageScope ts(A, STAGE_photometric_bwd, stream);
        ADK_STEP_TRY(STAGE_photometric_bwd,
                     adk_photometric_bwd(W, H, cf(A->render_colors), cf(A->render_alphas), cf(A->bg), cf(A->exposure), cf(A->gt), cf(A->mono),
                                         cf(A->rdk), A->mask_outliers, cf(A->v_img), cf(A->unit_grad), A->lambda_dssim, A->depth_weight, f(A->v_col),
                                         f(A->v_alpha), f(A->v_exposure), stream));
    }
    ADK_STEP_TRY(STAGE_raster_bwd, adk::step_zero(A->v_rec, (int64_t)N * 12 * sizeof(float), stream));
    {
        adk::StageScope ts(A, STAGE_raster_bwd, stream);
        ADK_STEP_TRY(STAGE_raster_bwd,
                     adk_raster_bwd_t(W, H, tpw, tph, cf(A->rec), static_cast<const int32_t*>(A->flatten_ids), static_cast<const int32_t*>(A->offsets),
                                      n_isects, nullptr, cf(A->final_T), static_cast<const int32_t*>(A->last_ids), cf(A->v_col), cf(A->v_alpha),
                                      f(A->v_rec), stream));
    }
    {
        adk::StageScope ts(A, STAGE_project_bwd, stream);
        float* const cam_grad = A->pose_grad ? f(A->cam_grad) : nullptr;
        float* const v_viewmat = A->pose_grad ? f(A->v_viewmat) : nullptr;
        // Keyframe.get_Rt's backward rides in the launch that finalises the camera gradient
        const float* const pose_r6 = A->pose_grad ? cf(A->r6) : nullptr;
        if (A->color_adam) {
            ADK_STEP_TRY(STAGE_project_bwd,
                         adk::project_bwd_adam_launch(N, cf(A->xyz), cf(A->quat), cf(A->scale), f(A->f_dc), f(A->f_rest), A->sh_K, A->sh_degree,
                                                      cf(A->viewmat), cf(A->Kmat), W, H, A->eps2d, A->near_plane, A->far_plane, 0,
                                                      static_cast<const int32_t*>(A->radii), cf(A->v_rec), f(A->v_means), f(A->v_quats),
                                                      f(A->v_scales), f(A->v_opac), cam_grad, v_viewmat, f(A->exp_avg_dc), f(A->exp_avg_sq_dc),
                                                      f(A->exp_avg_rest), f(A->exp_avg_sq_rest), cf(A->lr_dc), cf(A->lr_rest), A->adam_b1,
                                                      A->adam_b2, A->adam_eps, pose_r6, f(A->v_r6), f(A->v_t), stream));
        } else {
            ADK_STEP_TRY(STAGE_project_bwd,
                         adk::project_bwd_launch(N, cf(A->xyz), cf(A->quat), cf(A->scale), cf(A->f_dc), cf(A->f_rest), A->sh_K, A->sh_degree, 0,
                                                 cf(A->viewmat), cf(A->Kmat), W, H, A->eps2d, A->near_plane, A->far_plane, 0,
                                                 static_cast<const int32_t*>(A->radii), cf(A->v_rec), f(A->v_means), f(A->v_quats), f(A->v_scales),
                                                 f(A->v_opac), f(A->v_dc), f(A->v_rest), cam_grad, v_viewmat, nullptr, pose_r6, f(A->v_r6), f(A->v_t),
                                                 stream));
        }
    }
    if (scale_reg) {
        hipLaunchKernelGGL(adk::scale_reg_bwd_kernel, dim3((unsigned)adk::ceil_div((int64_t)N, (int64_t)256)), dim3(256), 0, stream, N, cf(A->scale),
                           static_cast<const uint8_t*>(A->sel), static_cast<const double*>(A->reg_ws), cf(A->unit_grad), f(A->v_scales));
        ADK_STEP_TRY(STAGE_lod_params_bwd, (int)hipGetLastError());
    }
    {
        adk::StageScope ts(A, STAGE_lod_params_bwd, stream);
        // v_means doubles as the LoD backward's v_xyz_add: the fade term is accumulated into the rasteriser's gradient of the means
        ADK_STEP_TRY(STAGE_lod_params_bwd,
                     adk_lod_params_bwd(N, cf(A->xyz), cf(A->opacity_raw), cf(A->scaling_raw), cf(A->rotation), cf(A->local_feat), cf(A->global_feat),
                                        static_cast<const int64_t*>(A->cls_id), cf(A->d_max), 16, 16, 32, cf(A->W1), cf(A->b1), cf(A->W2), cf(A->b2),
                                        cf(A->viewmat), cf(A->v_opac), cf(A->v_scales), cf(A->v_quats), f(A->v_means), f(A->v_opacity_raw),
                                        f(A->v_scaling_raw), f(A->v_rotation), f(A->v_local_feat), f(A->v_global_feat), f(A->v_mlp), A->lod_ws,
                                        A->lod_ws_bytes, stream));
    }
    return ADK_OK;
}
