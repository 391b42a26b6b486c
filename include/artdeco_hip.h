/*
 * artdeco_hip.h -- C ABI of libartdeco_hip.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary for the native operators on ARTDECO's mapper hot
 * path.  Each entry point names the reference interface it
 * replaces (paths relative to the ARTDECO tree).  Rules common to all of them:
 *
 *   - plain pointers and sizes only; every pointer is DEVICE memory on the
 *     current HIP device unless the comment says otherwise;
 *   - the caller owns all memory (inputs, outputs, scratch workspace); the
 *     library never allocates, frees or retains a pointer;
 *   - work is enqueued on the explicit `stream`; no call synchronises;
 *   - the return value is 0 on success, a positive hipError_t value if a launch
 *     failed, or a negative ADK_E* code for a rejected argument;
 *   - re-entrant: no mutable global state.
 *
 * Tensors are contiguous row-major fp32 unless stated.  Shapes use the
 * reference's names: N Gaussians, C cameras, P pixels/points, I tile
 * intersections.
 */
#ifndef ARTDECO_HIP_H
#define ARTDECO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* adk_stream_t; /* == hipStream_t */

#define ADK_OK 0
#define ADK_EINVAL (-1)
#define ADK_EWORKSPACE (-2)
#define ADK_EUNSUPPORTED (-3)

/* ABI version of this header; bumped on any signature change. */
#define ADK_ABI_VERSION 1
int adk_abi_version(void);

/* Streaming float4 copy of nbytes (multiple of 16, 16 B aligned pointers); used
 * by bench.py to measure the box's achievable HBM bandwidth.  No reference
 * counterpart. */
int adk_stream_copy(void* dst, const void* src, int64_t nbytes, adk_stream_t stream);

/* ------------------------------------------------------------------ fused-ssim
 * Replaces fusedssim() / fusedssim_backward() --
 * Reconstruct/submodules/fused-ssim/ssim.cu:434-473 and :481-517 (bound in
 * ext.cpp:4-7, wrapped by fused_ssim/__init__.py:8-42).
 *
 * img1,img2,ssim_map,dm_*: [B,CH,H,W].  dm_dmu1 == NULL selects inference mode
 * (train=False): only ssim_map is written.  "same" zero padding; the "valid"
 * crop is done by the caller exactly as __init__.py:13-14 does.
 */
int adk_fused_ssim_fwd(const float* img1, const float* img2, int B, int CH, int H, int W,
                       float C1, float C2, float* ssim_map, float* dm_dmu1,
                       float* dm_dsigma1_sq, float* dm_dsigma12, adk_stream_t stream);

/* dL_dmap: [B,CH,H,W], or NULL meaning "every element equals dL_scalar" (the
 * gradient of map.mean(), __init__.py:42) which saves one 4 B/px read.
 * Writes dL_dimg1 [B,CH,H,W] (ssim.cu:420). */
int adk_fused_ssim_bwd(const float* img1, const float* img2, const float* dL_dmap, float dL_scalar,
                       const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                       int B, int CH, int H, int W, float* dL_dimg1, adk_stream_t stream);

/* ------------------------------------------------- diff_gaussian_rasterization
 * Replaces adamUpdate(param, grad, exp_avg, exp_avg_sq, visible, lr, b1, b2,
 * eps, N, M) [UPSTREAM on-the-fly-nvs fork, not vendored]; call sites
 * Reconstruct/scene/optimizers.py:116-128 and :144-156.
 *
 * In place on param/exp_avg/exp_avg_sq ([N,M] flat), rows with visible[row]==0
 * untouched.  lr is a device tensor with lr_numel in {1, N, N*M} selecting
 * lr[0] / lr[row] / lr[row*M+col].  No bias correction.
 */
int adk_adam_update(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                    const uint8_t* visible, const float* lr, int64_t lr_numel, float b1, float b2,
                    float eps, int64_t N, int64_t M, adk_stream_t stream);

/* Replaces adamUpdateBasic(param, grad, exp_avg, exp_avg_sq, lr, b1, b2, eps);
 * call sites optimizers.py:48-57 and :90-99 (python-float lr, dense). */
int adk_adam_update_basic(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr,
                          float b1, float b2, float eps, int64_t numel, adk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ARTDECO_HIP_H */
