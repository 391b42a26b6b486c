// Entry points shared between translation units of libartdeco_hip.so that are NOT part of the C ABI (include/artdeco_hip.h): forms of exported
// stages that only the one-call optimisation step (mapper_step.hip) uses, where folding a tiny launch into a neighbour saves its ~4 us.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace adk {

// adk_project_fwd that also writes the visibility masks of SceneModel.render (h3dgsv3.py:695-698) -- what adk_visibility_masks computes from the
// radii this kernel has just formed: vis[g] = radii[g] > 0 on both axes; gvis[cls_id[g]] = 1 for every visible g.  gvis must have been zeroed by
// an earlier launch (pose6d_fwd_clear).
struct ProjectMasks { const int64_t* cls_id; int64_t V; uint8_t* vis; uint8_t* gvis; };
int project_fwd_launch(int N, const float* means, const float* quats, const float* scales, const float* opacities, const float* colors_in,
                       const float* sh_rest, int sh_K, int sh_degree, int color_mode, const float* viewmat, const float* Kmat, int width, int height,
                       float eps2d, float near_plane, float far_plane, float radius_clip, int inv_depth, float* rec, int32_t* radii,
                       uint32_t* depth_keys, uint32_t* gauss_ids, int32_t* tiles_per_gauss, const ProjectMasks* masks, hipStream_t stream);

// adk_lod_params_fwd + adk_project_fwd (SH colours in ARTDECO's split f_dc / f_rest layout, gsplat's "RGB+D") as ONE kernel: the activated
// parameters (opac_eff, scale_eff, quat_eff, selected) are written for the backward but not re-read by the projection.
int lod_project_fwd_launch(int N, const float* xyz, const float* opacity_raw, const float* scaling_raw, const float* rotation,
                           const float* local_feat, const float* global_feat, const int64_t* cls_id, const float* d_max, const float* W1,
                           const float* b1, const float* W2, const float* b2, float* opac_eff, float* scale_eff, float* quat_eff,
                           uint8_t* selected, const float* f_dc, const float* f_rest, int sh_K, int sh_degree, const float* viewmat,
                           const float* Kmat, int width, int height, float eps2d, float near_plane, float far_plane, float radius_clip,
                           float* rec, int32_t* radii, uint32_t* depth_keys, uint32_t* gauss_ids, int32_t* tiles_per_gauss,
                           const ProjectMasks* masks, hipStream_t stream);

// adk_pose6d_fwd + zero fill of up to two byte spans in the same launch.
int pose6d_fwd_clear(const float* r6, const float* t, float* Rt, void* a, int64_t na, void* b, int64_t nb, hipStream_t stream);

struct ColorAdam;
// adk_project_bwd / adk_project_bwd_adam; pose_r6 != nullptr: the single-thread launch that turns cam_grad into v_viewmat also runs
// adk_pose6d_bwd's arithmetic on it (v_r6 [3,2], v_t [3]).
int project_bwd_launch(int N, const float* means, const float* quats, const float* scales, const float* colors_in, const float* sh_rest, int sh_K,
                       int sh_degree, int color_mode, const float* viewmat, const float* Kmat, int width, int height, float eps2d, float near_plane,
                       float far_plane, int inv_depth, const int32_t* radii, const float* v_rec, float* v_means, float* v_quats, float* v_scales,
                       float* v_opacities, float* v_colors, float* v_sh_rest, float* cam_grad, float* v_viewmat, const ColorAdam* opt,
                       const float* pose_r6, float* v_r6, float* v_t, hipStream_t stream);
// the ColorAdam of adk_project_bwd_adam's arguments (argument checks included); returns ADK_OK or the code adk_project_bwd_adam would
int project_bwd_adam_launch(int N, const float* means, const float* quats, const float* scales, float* f_dc, float* f_rest, int sh_K, int sh_degree,
                            const float* viewmat, const float* Kmat, int width, int height, float eps2d, float near_plane, float far_plane,
                            int inv_depth, const int32_t* radii, const float* v_rec, float* v_means, float* v_quats, float* v_scales,
                            float* v_opacities, float* cam_grad, float* v_viewmat, float* m_dc, float* v_dc, float* m_rest, float* v_rest,
                            const float* lr_dc, const float* lr_rest, float b1, float b2, float eps, const float* pose_r6, float* v_r6, float* v_t,
                            hipStream_t stream);

} // namespace adk
