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
#define ADK_ABI_VERSION 19
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

/* The training forward for a caller that only needs map.mean() (fused_ssim/__init__.py:42; the mapper's loss,
 * h3dgsv3.py:441): besides the dm_* maps it leaves sum(ssim_map) as adk_fused_ssim_fwd_sums_count(B,CH,H,W) partial
 * sums (one per strip of the kernel's grid, fixed order => deterministic) in block_sums; ssim_map may be NULL, in
 * which case the map is never written.  adk_photometric_loss_sums consumes them. */
int64_t adk_fused_ssim_fwd_sums_count(int B, int CH, int H, int W);
int adk_fused_ssim_fwd_sums(const float* img1, const float* img2, int B, int CH, int H, int W, float C1, float C2,
                            float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                            float* block_sums, adk_stream_t stream);

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

/* Multi-tensor form of the two above: everything SparseGaussianAdam.step
 * (optimizers.py:77-161) does for n <= 16 tensors in ONE launch, including the per-element lr decay
 * `lr[vis] *= decay; lr.clamp_min_(lr_min)` of :158-161.  All array arguments are HOST arrays of length n read
 * during the call; their entries are device pointers / sizes per tensor.  visibles[i] == NULL => dense update;
 * lr_ptrs[i] == NULL => python-float lr lr_vals[i].  Same IEEE-unfused arithmetic as adk_adam_update. */
int adk_adam_update_multi(int n, float* const* params, const float* const* grads, float* const* exp_avgs,
                          float* const* exp_avg_sqs, const uint8_t* const* visibles, float* const* lr_ptrs,
                          const int64_t* lr_numels, const float* lr_vals, const float* lr_decays,
                          const float* lr_mins, const int64_t* rows, const int64_t* Ms, float b1, float b2,
                          float eps, adk_stream_t stream);
/* The same with one Adam configuration PER TENSOR (b1s, b2s, epss: host arrays of length n): the keyframe's own
 * BaseAdam (scene/keyframe.py:113-125, betas (0.8, 0.99), three adamUpdateBasic calls of 6 / 3 / 12 floats per step,
 * optimizers.py:41-57) rides in the Gaussians' launch. */
int adk_adam_update_multi_betas(int n, float* const* params, const float* const* grads, float* const* exp_avgs,
                                float* const* exp_avg_sqs, const uint8_t* const* visibles, float* const* lr_ptrs,
                                const int64_t* lr_numels, const float* lr_vals, const float* lr_decays,
                                const float* lr_mins, const int64_t* rows, const int64_t* Ms, const float* b1s,
                                const float* b2s, const float* epss, adk_stream_t stream);

/* ---------------------------------------------------------------------- gsplat
 * The five stages behind gsplat.rendering.rasterization(...) [UPSTREAM gsplat >= 1.5, not
 * vendored] as called at Reconstruct/scene/scene_models/h3dgsv3.py:664-680 (one camera per
 * call; packed=False; rasterize_mode="classic"; tile 16).  Stage boundaries are where the
 * caller must size a tensor (n_isects) -- everything else is fused.
 *
 * Packed per-Gaussian "splat record" rec[N][12] (three float4, 16 B aligned):
 *   [0] mean2d.x [1] mean2d.y [2] opacity [3] radius_x | [4..6] conic a,b,c [7] radius_y |
 *   [8..11] colour channels (color_mode 0/1: r,g,b,depth; 2: depth,0,0,0)
 * Gradient record v_rec[N][12] mirrors it ([2] = v_opacity, [3],[7] unused), except that slots
 * [0],[1] carry s = sum over pixels of v_sigma * (mean2d - pixel); adk_project_bwd forms
 * v_mean2d = conic * s itself (once per Gaussian instead of once per (splat, pixel)).
 */

/* Replaces fully_fused_projection fwd + spherical_harmonics fwd + the counting half of
 * isect_tiles.  colors_in: SH coefficients [N,sh_K,3] (color_mode 0), RGB [N,3] (1) or NULL (2).
 * viewmat [4,4] world->camera and Kmat [3,3], row-major, DEVICE memory.
 * Out: rec, radii int32 [N,2], depth_keys u32 [N] (float bits of z, 0xFFFFFFFF if culled),
 * gauss_ids u32 [N] (0..N-1), tiles_per_gauss int32 [N].
 * sh_rest != NULL (color_mode 0 only): band 0 is read from colors_in [N,1,3] and bands 1..sh_K-1 from
 * sh_rest [N,sh_K-1,3] -- ARTDECO's separate f_dc / f_rest tensors, no concatenation needed.
 * inv_depth != 0 stores 1/z instead of z in the depth channel (the invdepth output of the
 * on-the-fly-nvs GaussianRasterizer, Reconstruct/webviewer/scene_models.py:596). */
int adk_project_fwd(int N, const float* means, const float* quats, const float* scales,
                    const float* opacities, const float* colors_in, const float* sh_rest, int sh_K, int sh_degree,
                    int color_mode, const float* viewmat, const float* Kmat, int width, int height, float eps2d,
                    float near_plane, float far_plane, float radius_clip, int inv_depth, float* rec, int32_t* radii,
                    uint32_t* depth_keys, uint32_t* gauss_ids, int32_t* tiles_per_gauss, adk_stream_t stream);

/* Replaces fully_fused_projection bwd + spherical_harmonics bwd (+ the torch.inverse(viewmats)
 * autograd edge of rasterization()).  Any v_* output may be NULL.  cam_grad: 128 zeroed BYTES of 8-byte-aligned
 * scratch (16 doubles since ABI v19: the 15 camera sums are accumulated in fp64 -- as fp32 atomics they sat 4e-4 off the fp64 gradient at
 * 1 M Gaussians; the parameter keeps its float* type), required iff v_viewmat [4,4] is requested; the call leaves them zeroed again
 * (stream-ordered), so a caller may keep one accumulator per stream instead of clearing a fresh one for every call. */
int adk_project_bwd(int N, const float* means, const float* quats, const float* scales,
                    const float* colors_in, const float* sh_rest, int sh_K, int sh_degree, int color_mode,
                    const float* viewmat, const float* Kmat, int width, int height, float eps2d,
                    float near_plane, float far_plane, int inv_depth, const int32_t* radii,
                    const float* v_rec, float* v_means, float* v_quats, float* v_scales,
                    float* v_opacities, float* v_colors, float* v_sh_rest, float* cam_grad,
                    float* v_viewmat, adk_stream_t stream);

/* adk_project_bwd with the sparse-Adam step of the SH coefficients (ARTDECO's f_dc [N,1,3] / f_rest [N,K-1,3],
 * Reconstruct/scene/optimizers.py:106-128) applied in the same pass: the coefficient gradients are never written;
 * f_dc, f_rest and their moments (m_*, v_* = exp_avg, exp_avg_sq) are updated IN PLACE for every Gaussian with
 * radii > 0 -- the rows adamUpdate(..., visible = radii > 0, ...) touches -- bit-identically to adk_project_bwd
 * followed by adk_adam_update.  lr_dc / lr_rest: 0-dim DEVICE tensors (optimizers.py:70-73). */
int adk_project_bwd_adam(int N, const float* means, const float* quats, const float* scales,
                         float* f_dc, float* f_rest, int sh_K, int sh_degree,
                         const float* viewmat, const float* Kmat, int width, int height, float eps2d,
                         float near_plane, float far_plane, int inv_depth, const int32_t* radii,
                         const float* v_rec, float* v_means, float* v_quats, float* v_scales,
                         float* v_opacities, float* cam_grad, float* v_viewmat,
                         float* m_dc, float* v_dc, float* m_rest, float* v_rest, const float* lr_dc,
                         const float* lr_rest, float b1, float b2, float eps, adk_stream_t stream);

/* *n_isects (int64, device) = sum(tiles_per_gauss): the size of the intersection list, known right after
 * the projection.  Lets the host read it (pinned copy + event) while adk_bin_depth_order is still running,
 * instead of the stream-draining read gsplat does after isect_tiles (rendering.py, `isect_tiles` -> n_isects). */
int adk_bin_count_isects(int N, const int32_t* tiles_per_gauss, int64_t* n_isects, adk_stream_t stream);

/* Replaces isect_tiles + radix sort + isect_offset_encode, in two calls around the single
 * point where the caller needs a size (n_isects).  Output order is bit-identical to a stable
 * sort of upstream's 64-bit (tile<<32 | depth bits) keys. */
int64_t adk_bin_depth_workspace_bytes(int N);
int adk_bin_depth_order(int N, const uint32_t* depth_keys, const uint32_t* gauss_ids,
                        const int32_t* tiles_per_gauss, uint32_t* sorted_ids, uint32_t* block_offs,
                        int64_t* n_isects, void* workspace, int64_t workspace_bytes, adk_stream_t stream);
int64_t adk_bin_tiles_workspace_bytes(int64_t n_isects);
int adk_bin_tiles(int N, int64_t n_isects, const uint32_t* sorted_ids, const uint32_t* block_offs,
                  const int32_t* tiles_per_gauss, const float* rec, int width, int height,
                  int32_t* flatten_ids, uint32_t* tile_ids, int32_t* offsets, void* workspace,
                  int64_t workspace_bytes, adk_stream_t stream);
/* The same outputs by the TILE-LOCAL route (default when supported): counting sort by tile with a per-slice tile histogram
 * in LDS, then one workgroup per tile sorts its (depth bits << 32 | id) keys in LDS.  5 launches instead of 18, no sort of
 * the N depth keys at all; bit-identical (tile, depth, id) order.
 *   adk_bin_local_supported(w, h)  1 when the tile histogram fits LDS (<= 32768 tiles), else use the two calls above;
 *   adk_bin_local_count            offsets [tile_h*tile_w] and stats [2] int64 (device): n_isects, entries of the fullest tile;
 *   adk_bin_local_scatter          keys of every (Gaussian, tile) pair into the tiles' segments of `pairs` (capacity x 8 B; entries past
 *                                  the capacity are dropped, so it may be launched with an ESTIMATED capacity before n_isects is read);
 *   adk_bin_local_sort             n_isects / max_tile = the host copies of stats; max_tile > 8192 -> ADK_EUNSUPPORTED (use the
 *                                  global route); sorts every tile's segment in LDS; tile_ids may be NULL. */
int adk_bin_local_supported(int width, int height);
int64_t adk_bin_local_workspace_bytes(int width, int height);
int adk_bin_local_count(int N, const int32_t* tiles_per_gauss, const float* rec, int width, int height,
                        int32_t* offsets, int64_t* stats, void* workspace, int64_t workspace_bytes,
                        adk_stream_t stream);
int64_t adk_bin_local_pairs_bytes(int64_t n_isects);
int adk_bin_local_scatter(int N, int64_t capacity, const uint32_t* depth_keys, const int32_t* tiles_per_gauss,
                          const float* rec, int width, int height, const int32_t* offsets,
                          const void* workspace, int64_t workspace_bytes, void* pairs, adk_stream_t stream);
int adk_bin_local_sort(int64_t n_isects, int64_t max_tile, int width, int height, const int32_t* offsets,
                       const void* pairs, int32_t* flatten_ids, uint32_t* tile_ids, adk_stream_t stream);
/* The same three steps for an INTERNAL tile shape tile_px_w x tile_px_h in {16x16, 32x16} (the shapes adk_raster_fwd_t / adk_raster_bwd_t take; anything else: ADK_EINVAL): a wider tile lists every Gaussian
 * gsplat lists for one of the 16x16 tiles inside it, in the same (depth, id) order; offsets has one entry per internal tile.
 * (`adk_bin_local_*` without the suffix = 16x16 = gsplat's isect_tiles / isect_offset_encode outputs.) */
int adk_bin_local_supported_t(int width, int height, int tile_px_w, int tile_px_h);
int64_t adk_bin_local_workspace_bytes_t(int width, int height, int tile_px_w, int tile_px_h);
int adk_bin_local_count_t(int N, const int32_t* tiles_per_gauss, const float* rec, int width, int height, int tile_px_w, int tile_px_h,
                          int32_t* offsets, int64_t* stats, void* workspace, int64_t workspace_bytes, adk_stream_t stream);
int adk_bin_local_scatter_t(int N, int64_t capacity, const uint32_t* depth_keys, const int32_t* tiles_per_gauss, const float* rec,
                            int width, int height, int tile_px_w, int tile_px_h, const int32_t* offsets, const void* workspace,
                            int64_t workspace_bytes, void* pairs, adk_stream_t stream);
int adk_bin_local_sort_t(int64_t n_isects, int64_t max_tile, int width, int height, int tile_px_w, int tile_px_h, const int32_t* offsets,
                         const void* pairs, int32_t* flatten_ids, uint32_t* tile_ids, adk_stream_t stream);
/* adk_bin_local_sort_t WITHOUT the 8 192-entry limit (round 5; gsplat's isect_tiles + radix sort have none, h3dgsv3.py:664-680): tiles of up to
 * 8 192 entries as before, longer ones -- up to adk_bin_local_sort_long_max() = 4 194 304 entries per tile -- by one workgroup each that
 * partitions the tile's segment recursively on the key range, ping-ponging between `pairs` and `scratch`, and sorts the pieces in registers.
 * `pairs` is therefore READ AND OVERWRITTEN; scratch: scratch_bytes >= 8 * n_isects, 8 B aligned (NULL allowed while max_tile <= 8 192).
 * Same (tile, depth, id) order, bit for bit.  max_tile above the limit: ADK_EUNSUPPORTED (global route). */
int adk_bin_local_sort_long_t(int64_t n_isects, int64_t max_tile, int width, int height, int tile_px_w, int tile_px_h, const int32_t* offsets,
                              void* pairs, void* scratch, int64_t scratch_bytes, int32_t* flatten_ids, uint32_t* tile_ids, adk_stream_t stream);
int64_t adk_bin_local_sort_long_max(void);
/* Optional: rebuild upstream's sorted int64 isect_ids for meta['isect_ids']. */
int adk_bin_make_isect_ids(int64_t n_isects, const uint32_t* tile_ids, const int32_t* flatten_ids,
                           const uint32_t* depth_keys, int64_t* isect_ids, adk_stream_t stream);

/* Replaces rasterize_to_pixels fwd: render_colors [H,W,4], render_alphas [H,W], last_ids [H,W];
 * final_T [H,W] = the exact final transmittance of each pixel, saved for the backward (upstream
 * recovers it as 1 - render_alphas, which loses up to 1e-4 relative where alpha ~ 1);
 * last_ids [H,W] is the forward -> backward hand-off only (no caller reads it): the list index the backward starts this pixel at -- the
 * entry in front of the splat the pixel STOPPED at (T (1 - alpha) <= 1e-4), or the tile's last entry if it never stopped (upstream stores
 * the last CONTRIBUTING entry; every entry between the two fails the same alpha >= 1/255 / sigma >= 0 tests in the backward as it did in the
 * forward, so the gradients are the same -- ABI v19);
 * backgrounds [4] or NULL; main_ids [H,W] or NULL = id of the Gaussian with the largest alpha*T per
 * pixel, -1 if none (mainGaussID of the on-the-fly-nvs GaussianRasterizer). */
int adk_raster_fwd(int width, int height, const float* rec, const int32_t* flatten_ids,
                   const int32_t* offsets, int64_t n_isects, const float* backgrounds,
                   float* render_colors, float* render_alphas, float* final_T, int32_t* last_ids,
                   int32_t* main_ids, adk_stream_t stream);

/* adk_raster_fwd / adk_raster_bwd on lists binned for an internal tile of tile_px_w x tile_px_h (16x16 or 32x16): per pixel the same
 * splats are composited in the same order, so the forward is bit-identical to the 16x16 form; one wave serves the whole internal tile. */
int adk_raster_fwd_t(int width, int height, int tile_px_w, int tile_px_h, const float* rec, const int32_t* flatten_ids,
                     const int32_t* offsets, int64_t n_isects, const float* backgrounds,
                     float* render_colors, float* render_alphas, float* final_T, int32_t* last_ids,
                     int32_t* main_ids, adk_stream_t stream);
int adk_raster_bwd_t(int width, int height, int tile_px_w, int tile_px_h, const float* rec, const int32_t* flatten_ids,
                     const int32_t* offsets, int64_t n_isects, const float* backgrounds,
                     const float* final_T, const int32_t* last_ids, const float* v_render_colors,
                     const float* v_render_alphas, float* v_rec, adk_stream_t stream);

/* Replaces rasterize_to_pixels bwd: accumulates into v_rec [N,12] (caller zero-fills it). */
int adk_raster_bwd(int width, int height, const float* rec, const int32_t* flatten_ids,
                   const int32_t* offsets, int64_t n_isects, const float* backgrounds,
                   const float* final_T, const int32_t* last_ids, const float* v_render_colors,
                   const float* v_render_alphas, float* v_rec, adk_stream_t stream);

/* -------------------------------------------------------------- mast3r_slam_backends
 * Replaces iter_proj(rays_img_with_grad, pts_3d_norm, p_init, max_iter, lambda_init, cost_thresh)
 * -- VSLAM/backend/src/gn.cpp:84-99 -> matching_kernels.cu:119-316.
 * rays [batch,h,w,9], pts [batch,n,3], p_init [batch,n,2] -> p_new [batch,n,2] float,
 * converged [batch,n] (1 byte per entry, 0/1). */
int adk_iter_proj(const float* rays_img_with_grad, const float* pts_3d_norm, const float* p_init, int batch,
                  int h, int w, int n, int max_iter, float lambda_init, float cost_thresh, float* p_new,
                  uint8_t* converged, adk_stream_t stream);

/* Replaces refine_matches(D11, D21, p1, radius, dilation_max) -- gn.cpp:101-114 ->
 * matching_kernels.cu:25-116.  dtype 0 = float16, 1 = float32 descriptors (score accumulates in
 * that type).  D11 [batch,h,w,fdim], D21 [batch,n,fdim], p1/p1_new [batch,n,2] int64 (u,v). */
int adk_refine_matches(const void* D11, const void* D21, const int64_t* p1, int dtype, int batch, int h,
                       int w, int n, int fdim, int radius, int dilation_max, int64_t* p1_new,
                       adk_stream_t stream);

/* ----------------------------------------------------------------------- curope
 * Replaces rope_2d(tokens, positions, base, fwd) -- VSLAM/thirdparty/mast3r/dust3r/croco/models/
 * curope/curope.cpp:49-65 (CUDA kernel kernels.cu:17-108).  In place on tokens [B,N,H,D]
 * (dtype 0 = float16, 1 = float32, 2 = bfloat16; D % 4 == 0) whose last two dims are dense (strides D, 1, as
 * kernels.cu:90 requires) with element strides stride_b / stride_n for batch / token -- the
 * reference hands over a transposed view of the qkv projection (blocks.py, curope2d.py:37);
 * positions [B,N,2] int64 (y, x); fwd = +F0 forward, -F0 backward (curope2d.py:20,27). */
int adk_rope_2d(void* tokens, const int64_t* positions, int dtype, int B, int N, int64_t stride_b,
                int64_t stride_n, int H, int D, float base, float fwd, adk_stream_t stream);

/* The same rotation split in two: every block of the model rotates q and k by the SAME positions, so the powf / cosf / sinf of
 * kernels.cu:38-44 are evaluated once into table [B*N][2 (y, x)][D/4][2 (cos, sin)] float32 (adk_rope_2d_table, the identical
 * fp32 expressions) and adk_rope_2d_apply streams the tokens through it: results bit-identical to adk_rope_2d.
 * apply: D % 16 == 0, strides multiples of 4 elements. */
int adk_rope_2d_table(const int64_t* positions, int64_t n_tokens, int D, float base, float fwd, float* table, adk_stream_t stream);
int adk_rope_2d_apply(void* tokens, const float* table, int dtype, int B, int N, int64_t stride_b, int64_t stride_n, int H, int D,
                      adk_stream_t stream);

/* -------------------------------------------------------------------- attention
 * Replaces the attention of the MASt3R blocks -- VSLAM/thirdparty/mast3r/dust3r/croco/models/blocks.py:97-111 (self:
 * `attn = (q @ k.transpose(-2, -1)) * self.scale; attn = attn.softmax(dim=-1); x = (attn @ v).transpose(1, 2).reshape(B, N, C)`)
 * and :138-157 (cross, same arithmetic on separate q / k / v projections).  Head dim 64, float16 operands, float32 scores,
 * softmax and accumulation.  q [B,H,Nq,64], k / v [B,H,Nk,64] as base pointers + ELEMENT strides {batch, head, token} (last
 * dim dense; the model passes views of its fused qkv projection), out [B,Nq,H,64] contiguous, i.e. already the
 * `.transpose(1, 2).reshape(B, N, C)` of blocks.py:109.  Strides multiples of 8, pointers 16-byte aligned. */
int adk_attention_fwd_f16(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk,
                          const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, float scale,
                          adk_stream_t stream);

/* Replaces the `x = x + f(norm(x))` glue of the MASt3R blocks -- croco/models/blocks.py:88-95 (Block.forward) and
 * :176-191 (DecoderBlock.forward): residual add, nn.LayerNorm(eps=1e-6) and the cast of its output to the GEMM operand type.
 * x_in [rows,C] float32; delta [rows,C] float16 or NULL; gamma / beta [C] float32.  With delta: x_out [rows,C] float32 (may be
 * x_in) = x_in + delta and y = LayerNorm(x_out); without: y = LayerNorm(x_in), x_out ignored.  y_out [rows,C] float16 when
 * y_f16 != 0, else float32.  Biased variance, fp32 statistics.  C % 4 == 0, C <= 2048. */
int adk_add_layernorm(const float* x_in, const void* delta, const float* gamma, const float* beta, float eps, int rows, int C,
                      float* x_out, void* y_out, int y_f16, adk_stream_t stream);

/* -------------------------------------------------------------------- simple-knn
 * Exact K nearest neighbours, squared distances, self excluded by index.  Workspace sized by the
 * number of points the search structure is built over (P for the first two, N for indexQ).
 * Unlike the reference (simple_knn.cu:200,203,480,483) nothing is copied to the host. */
int64_t adk_knn_workspace_bytes(int64_t n_struct_points);

/* Replaces distIndex2(points, K) = SimpleKNN::knn_index2 -- spatial.cu:28-41, simple_knn.cu:468-522.
 * points [P,3]; dists float [P*K], indices int32 [P*K], row = input point; neighbour order within
 * a row unspecified (ascending here); fewer than K neighbours leaves (FLT_MAX, -1).  K in 1..8, 12, 16. */
int adk_knn_index2(const float* points, int P, int K, float* dists, int32_t* indices, void* workspace,
                   int64_t workspace_bytes, adk_stream_t stream);
/* MEASUREMENT ONLY ("runtime is search-bound, so also report points-visited/query"): adk_knn_index2 with K = 3 and the search's
 * work summed over the P queries into stats[4] (uint64, zeroed by the caller): boxes scanned, points visited, super-box tests, box tests. */
int adk_knn_index2_stats(const float* points, int P, float* dists, int32_t* indices, void* workspace, int64_t workspace_bytes,
                         unsigned long long* stats, adk_stream_t stream);

/* Replaces distCUDA2(points) = SimpleKNN::knn -- spatial.cu:14-25, simple_knn.cu:188-227:
 * mean_dists[P] = mean of the 3 smallest squared distances. */
int adk_knn_mean_dist3(const float* points, int P, float* mean_dists, void* workspace,
                       int64_t workspace_bytes, adk_stream_t stream);

/* Replaces distIndexQ(points, q_idx, n_idx, K) = SimpleKNN::knn_indexQ -- spatial.cu:43-58,
 * simple_knn.cu:592-651: queries points[q_idx[i]], candidates only points[n_idx[*]];
 * dists/indices [Q*K] in query order. */
int adk_knn_indexQ(const float* points, int P, const int32_t* q_idx, int Q, const int32_t* n_idx, int N,
                   int K, float* dists, int32_t* indices, void* workspace, int64_t workspace_bytes,
                   adk_stream_t stream);

/* ------------------------------------------------------ fused LoD / mlp_cov glue
 * Replaces the torch ops of SceneModel.render between the parameter dictionary and the rasteriser
 * (Reconstruct/scene/scene_models/h3dgsv3.py:626-662): LoD selection + fade, sigmoid/exp
 * activations, the mlp_cov (Linear(32,32)-ReLU-Linear(32,7)) scale/rotation modulation.  All N
 * Gaussians are processed; unselected ones get opacity 0 (culled by adk_project_fwd), so no
 * compaction and no host sync.  local_dim = global_dim = 16, hidden_dim = 32 (run.sh) only.
 * W1 [32,32], b1 [32], W2 [7,32], b2 [7] row-major as torch.nn.Linear stores them.
 * Out: opac_eff [N], scale_eff [N,3], quat_eff [N,4] (un-normalised), selected [N] (0/1 bytes). */
int adk_lod_params_fwd(int N, const float* xyz, const float* opacity_raw, const float* scaling_raw,
                       const float* rotation, const float* local_feat, const float* global_feat,
                       const int64_t* cls_id, const float* d_max, int local_dim, int global_dim, int hidden_dim,
                       const float* W1, const float* b1, const float* W2, const float* b2, const float* viewmat,
                       float* opac_eff, float* scale_eff, float* quat_eff, uint8_t* selected, adk_stream_t stream);

int64_t adk_lod_params_bwd_workspace_bytes(int N);

/* Backward of the above.  v_xyz_add [N,3] is ACCUMULATED into (LoD fade term); v_global_feat
 * [V,16] must be zero-filled by the caller (atomic scatter); v_mlp [1287] = dW1 | db1 | dW2 | db2
 * (weight gradients contracted on the matrix cores, deterministic two-stage reduction). */
int adk_lod_params_bwd(int N, const float* xyz, const float* opacity_raw, const float* scaling_raw,
                       const float* rotation, const float* local_feat, const float* global_feat,
                       const int64_t* cls_id, const float* d_max, int local_dim, int global_dim, int hidden_dim,
                       const float* W1, const float* b1, const float* W2, const float* b2, const float* viewmat,
                       const float* v_opac_eff, const float* v_scale_eff, const float* v_quat_eff,
                       float* v_xyz_add, float* v_opacity_raw, float* v_scaling_raw, float* v_rotation,
                       float* v_local_feat, float* v_global_feat, float* v_mlp, void* workspace,
                       int64_t workspace_bytes, adk_stream_t stream);

/* adk_lod_params_bwd with the sparse-Adam step of xyz / opacity / scaling / rotation / local_feat applied INSIDE the kernel: replaces
 * the adamUpdate calls of SparseGaussianAdam.step for those five keys (Reconstruct/scene/optimizers.py:136-161; 27 of a Gaussian's 75
 * floats -- the 48 SH colours take theirs inside adk_project_bwd_adam) on the rows with visible[g] != 0 (visibility = radii > 0,
 * h3dgsv3.py:695).  The five parameter tensors and their moments (m_*, v2_* = exp_avg, exp_avg_sq) are updated IN PLACE, their
 * gradients are never written; lr_xyz [N,3] is xyz's per-element learning rate, decayed in place on the visible rows after the update,
 * lr = max(lr * lr_decay_xyz, lr_min_xyz) (optimizers.py:158-161); lr_opacity .. lr_local are 0-dim device tensors.  v_xyz (the
 * rasteriser's gradient of the means) is only read.  v_global_feat (zero-filled by the caller) and v_mlp as adk_lod_params_bwd.
 * Arithmetic = adk_adam_update's, IEEE-unfused: bit-identical to adk_lod_params_bwd followed by adk_adam_update_multi. */
int adk_lod_params_bwd_adam(int N, float* xyz, float* opacity_raw, float* scaling_raw, float* rotation, float* local_feat,
                            const float* global_feat, const int64_t* cls_id, const float* d_max, int local_dim, int global_dim,
                            int hidden_dim, const float* W1, const float* b1, const float* W2, const float* b2, const float* viewmat,
                            const float* v_opac_eff, const float* v_scale_eff, const float* v_quat_eff, const float* v_xyz,
                            float* v_global_feat, float* v_mlp, void* workspace, int64_t workspace_bytes,
                            const uint8_t* visible, float* m_xyz, float* v2_xyz, float* lr_xyz, float lr_decay_xyz, float lr_min_xyz,
                            float* m_opacity, float* v2_opacity, const float* lr_opacity, float* m_scaling, float* v2_scaling,
                            const float* lr_scaling, float* m_rotation, float* v2_rotation, const float* lr_rotation,
                            float* m_local, float* v2_local, const float* lr_local, float beta1, float beta2, float eps,
                            adk_stream_t stream);

/* Replaces the per-keyframe loop of SceneModel.weed_out_gaussians (h3dgsv3.py:943-950): counts[g] = number of
 * keyframes whose camera centre -R^T t (R = sixD2mtx(r6[k]), r6 [n_kf,3,2], t [n_kf,3]) lies within 2 d_max[g]
 * of xyz[g].  One launch for all keyframes. */
int adk_lod_visible_count(int N, const float* xyz, const float* d_max, int n_keyframes, const float* r6,
                          const float* t, int32_t* counts, adk_stream_t stream);

/* Replaces the exposure correction of SceneModel.render_from_id (h3dgsv3.py:611-614):
 * out[3,P] = clamp(E[:3,:3] @ img[3,P] + E[:3,3,None], 0, 1), E [3,4] row-major (device). */
int adk_exposure_fwd(const float* E, const float* img, int64_t P, float* out, adk_stream_t stream);
/* Backward: v_img [3,P]; v_E [12] is ACCUMULATED into (caller zero-fills). */
int adk_exposure_bwd(const float* E, const float* img, const float* v_out, int64_t P, float* v_img,
                     float* v_E, adk_stream_t stream);

/* ------------------------------------------------- training-loss image chain
 * The image-space half of SceneModel.optimization_step between the rasteriser
 * output and loss.backward(), Reconstruct/scene/scene_models/h3dgsv3.py:
 *   :690-694 background composite + invdepth = 1 / depth,
 *   :611-614 exposure correction + clamp,
 *   :432-439 (is_important == False) outlier mask,
 *   :440-448 radial-decay-weighted L1 on colour and inverse depth, loss mix.
 * colors4 [H,W,4] / alphas [H,W] are the rasteriser's outputs as they are
 * (adk_raster_fwd); gt_image [3,H,W], mono_idepth [H,W], rdk [H,W] (the radial
 * decay kernel, utils.py:818-827); bg [3]; exposure [3,4] row-major, all device.
 * Writes image [3,H,W] (what fused_ssim is then called on), invdepth [H,W]
 * (unmasked, Keyframe.latest_invdepth) and, when mask_outliers != 0, gt_used
 * [3,H,W] = gt * mask.  Partial sums stay in the workspace. */
int64_t adk_photometric_workspace_bytes(int W, int H);
int adk_photometric_fwd(int W, int H, const float* colors4, const float* alphas, const float* bg,
                        const float* exposure, const float* gt_image, const float* mono_idepth,
                        const float* rdk, int mask_outliers, float* image, float* gt_used,
                        float* invdepth, void* workspace, int64_t workspace_bytes, adk_stream_t stream);
/* loss_out [4] = { lambda (1 - ssim) + (1 - lambda) l1 + depth_weight depth, l1, ssim, depth };
 * ssim_map [3,H,W] from adk_fused_ssim_fwd(image, gt).  Same workspace, same stream, after _fwd. */
int adk_photometric_loss(int W, int H, const float* ssim_map, float lambda_dssim, float depth_weight,
                         void* workspace, int64_t workspace_bytes, float* loss_out, adk_stream_t stream);
/* The same from the n_sums partial sums of adk_fused_ssim_fwd_sums(image, gt) instead of the map (one launch, no pass
 * over the map).  total_out (nullable): a second copy of loss_out[0] in a buffer of its own, so that a binding can hand
 * out the differentiable scalar and the by-product vector as two tensors without a copy kernel. */
int adk_photometric_loss_sums(int W, int H, const float* ssim_block_sums, int64_t n_sums, float lambda_dssim,
                              float depth_weight, void* workspace, int64_t workspace_bytes, float* loss_out,
                              float* total_out, adk_stream_t stream);
/* Backward of the whole chain.  v_image_ssim [3,H,W]: adk_fused_ssim_bwd(...) with dL_dmap = NULL and
 * dL_scalar = -lambda / (3 W H); v_loss: DEVICE scalar, autograd's gradient of the loss.  Writes
 * v_colors4 [H,W,4], v_alphas [H,W] (the layouts adk_raster_bwd consumes) and v_exposure [12]. */
int adk_photometric_bwd(int W, int H, const float* colors4, const float* alphas, const float* bg,
                        const float* exposure, const float* gt_image, const float* mono_idepth,
                        const float* rdk, int mask_outliers, const float* v_image_ssim, const float* v_loss,
                        float lambda_dssim, float depth_weight, float* v_colors4, float* v_alphas,
                        float* v_exposure, adk_stream_t stream);

/* Keyframe.get_Rt (Reconstruct/scene/keyframe.py:150-154) = [sixD2mtx(rW2C) | tW2C ; 0 0 0 1]
 * (utils.py:223-229): r6 [3,2] row-major, t [3] -> Rt [4,4]; and its backward. */
int adk_pose6d_fwd(const float* r6, const float* t, float* Rt, adk_stream_t stream);
int adk_pose6d_bwd(const float* r6, const float* v_Rt, float* v_r6, float* v_t, adk_stream_t stream);
/* Keyframe.set_Rt (scene/keyframe.py:156-159) as one launch: r6 [3,2] <- Rt[:3,:2], t [3] <- Rt[:3,3], centre [3] = -Rt[:3,:3]^T Rt[:3,3]
 * (Rt [4,4] row-major, contiguous).  Round 5: run_system.py's SLAM-keyframe loop calls get_Rt twice and set_Rt once per keyframe. */
int adk_pose6d_set(const float* Rt, float* r6, float* t, float* centre, adk_stream_t stream);
/* torch.linalg.inv / torch.inverse / Tensor.inverse of 4x4 fp32 matrices as run_system.py:221-224 calls them (three per mapper keyframe on a
 * SLAM keyframe; h3dgsv3.py:1000 once per add_keyframe): n matrices, element (r, c) of matrix i at in[i * batch_stride + r * row_stride +
 * c * col_stride] (element strides: the script inverts a transposed view), out [n,4,4] contiguous.  Gauss-Jordan, partial pivoting, fp64 inside.
 * No read-back: a singular matrix gives a NaN-filled result, info[i] = 1 + the failing column (info [n] or NULL) and -- ABI v19 --
 * *singular_count += 1 (system-scope atomic: the counter may be host-mapped pinned memory, so that the host learns of it at its next wait
 * without this call ever synchronising; NULL = not counted), not an exception. */
int adk_inv4x4(const float* in, float* out, int64_t n, int64_t batch_stride, int64_t row_stride, int64_t col_stride, int32_t* info,
               int32_t* singular_count, adk_stream_t stream);

/* Visibility masks of SceneModel.render (h3dgsv3.py:695-698): vis[g] = radii[g] > 0 (both axes);
 * gvis[cls_id[g]] = 1 for every visible g (gvis [V] bytes, cleared here; may be NULL). */
int adk_visibility_masks(int N, const int* radii, const int64_t* cls_id, int64_t V, uint8_t* vis,
                         uint8_t* gvis, adk_stream_t stream);

/* ------------------------------------------------------------ torch_scatter
 * Replaces torch_scatter.scatter_max / scatter_min (1-D) [UPSTREAM pytorch_scatter, not vendored]:
 * module-level import at Reconstruct/scene/scene_models/h3dgsv3.py:35, call :289 (update_voxel).
 * out[j] = max (min) of src[i] over index[i] == j, arg[j] = FIRST such i; empty groups: out = 0,
 * arg = n.  dtype 0 = float32, 1 = int32, 2 = int64 (src and out); index int64; arg int64. */
int adk_scatter_argmax(int64_t n, const void* src, int dtype, const int64_t* index, int64_t dim_size,
                       int is_min, void* out, int64_t* arg, adk_stream_t stream);

/* ------------------------------------------------ SceneModel.update_voxel
 * Replaces the torch chain of Reconstruct/scene/scene_models/h3dgsv3.py:227-316 (voxel hash, 3x torch.unique,
 * scatter_max majority vote, searchsorted, boolean-mask writes), called per LoD level at :884-887.
 * Stage 1 -- adk_voxel_bounds: minc [3] float = componentwise minimum of old ++ new points (:262-263), info [4] int64 =
 * grid extents nx, ny, nz (:267) and max(cls_id) (:258; -1 when N = 0); workspace >= 64 bytes.  The caller reads `info`
 * (it decides the number of radix passes) and passes it back to
 * Stage 2 -- adk_voxel_assign: updated_orig [N] / updated_new [M] int64 and *new_voxel_count (int64, device), exactly the
 * method's three results (two with N = 0, the cold start of :244-255).  xyz [N,3], new_xyz [M,3] float, cls_id [N] int64.
 * use_reciprocal selects how `(p - min) / voxel_size` (:266) is rounded: 0 = true fp32 division (torch's CPU kernel), 1 =
 * multiplication by the fp32 reciprocal (torch's GPU kernel for tensor / python-float, i.e. what the reference computes
 * where it actually runs); the two differ in the last bit for ~1e-7 of the coordinates. */
int adk_voxel_bounds(const float* xyz, int64_t N, const float* new_xyz, int64_t M, const int64_t* cls_id,
                     float voxel_size, int use_reciprocal, float* minc, int64_t* info, void* workspace, int64_t workspace_bytes,
                     adk_stream_t stream);
int64_t adk_voxel_workspace_bytes(int64_t N, int64_t M);
int adk_voxel_assign(const float* xyz, int64_t N, const float* new_xyz, int64_t M, const int64_t* cls_id,
                     float voxel_size, int use_reciprocal, const float* minc, int64_t nx, int64_t ny, int64_t nz, int64_t max_cls,
                     int64_t* updated_orig, int64_t* updated_new, int64_t* new_voxel_count, void* workspace,
                     int64_t workspace_bytes, adk_stream_t stream);
/* Stage 2 for ANOTHER batch of new points against the same old points (the next LoD level of add_new_gaussians, :884-887 inside
 * the loop of :775): searches the voxel table a previous adk_voxel_assign left in `workspace` (same N, same workspace, not written
 * in between) instead of rebuilding it.  Valid while xyz, cls_id, voxel_size, use_reciprocal, minc, nx / ny / nz and max_cls are
 * those of that call -- the caller checks the new batch's adk_voxel_bounds against them; updated_orig of that call still holds. */
int adk_voxel_assign_new(int64_t N, const float* new_xyz, int64_t M, float voxel_size, int use_reciprocal, const float* minc,
                         int64_t nx, int64_t ny, int64_t nz, int64_t max_cls, int64_t* updated_new, int64_t* new_voxel_count,
                         void* workspace, int64_t workspace_bytes, adk_stream_t stream);

/* ------------------------------------------- mast3r_slam_backends: Gauss-Newton
 * Replaces gauss_newton_points / gauss_newton_rays / gauss_newton_calib --
 * VSLAM/backend/src/gn.cpp:3-82, gn_kernels.cu:455-811 / :813-1215 / :1218-1637; callers
 * VSLAM/mast3r_slam/global_opt.py:158-173 and :208-228.
 * kind 0 = points (sigma_a = sigma_point), 1 = rays (sigma_a = sigma_ray, sigma_b = sigma_dist),
 * 2 = calib (sigma_a = sigma_pixel, sigma_b = sigma_depth, K [3,3] row-major, height, width,
 * pixel_border, z_eps).  Twc [P,8] = (t3, q xyzw, s), updated IN PLACE; Xs [P,n,3]; Cs [P,n];
 * ii / jj [E] int64 = position of each factor's keyframes in the pose arrays (the searchsorted indices
 * of gn_kernels.cu:163-169); idx_ii2jj [E,n] int64; valid_match [E,n] bytes; Q [E,n].  The first
 * num_fix (= 1, as in the reference) poses are held fixed.  dx_out [P-num_fix,7] receives the last
 * step.  Hs_dbg [4,E,7,7] / gs_dbg [2,E,7] (both or neither) receive the reference's per-factor blocks
 * of the last executed iteration.  The whole solve (assembly, fp64 Cholesky, retraction, termination
 * test) runs on the device; all max_iter iterations are enqueued and turn into no-ops once
 * |dx| < delta_thresh; the call never synchronises. */
int64_t adk_gn_workspace_bytes(int num_poses, int num_edges, int num_points);
int adk_gauss_newton(int kind, int num_poses, int num_edges, int num_points, float* Twc, const float* Xs,
                     const float* Cs, const float* K, const int64_t* ii, const int64_t* jj,
                     const int64_t* idx_ii2jj, const uint8_t* valid_match, const float* Q, int height,
                     int width, int pixel_border, float z_eps, float sigma_a, float sigma_b, float C_thresh,
                     float Q_thresh, int max_iter, float delta_thresh, int num_fix, float* dx_out,
                     float* Hs_dbg, float* gs_dbg, void* workspace, int64_t workspace_bytes,
                     adk_stream_t stream);

/* ------------------------------------------------------- frontend Sim(3) tracker
 * Replaces the body of CameraTracker.track between the MASt3R match and the keyframe decision --
 * VSLAM/CameraTracker.py:62-153: get_points_poses (:189-219; constrain_points_to_ray geometry.py:38-43,
 * local_diag_cov_from_X1 utils_uncertainty.py:5-53), the validity masks (:83-87), the insufficient-match test
 * (:90-91), opt_pose_calib_sim3 (:296-396, with the covariance filter :335-346; optimize_focal is not supported),
 * and the counts / quantile behind check_keyframe (:159-167) and check_keyframe_map (:170-186).
 * Xf_canon / Cf / Qf [n,3],[n],[n]: the frame's canonical pointmap, summed confidence (average = Cf * inv_Nf,
 * ImageFrame.get_average_conf) and descriptor confidence, frame pixel order; Xk_canon / Ck / Qk: the keyframe's
 * (Qk = the keyframe-in-frame descriptor confidence Qkf); idx_f2k [n] int64 and valid_match [n] bytes in keyframe
 * pixel order; K [3,3] row-major, T_WCf / T_WCk [8] = (t3, q xyzw, s): all DEVICE memory.  n = height * width.
 * result [32] floats (device): [0..7] new T_WCf (quat2unit'd; the input pose when lost or failed), [8..15] T_CkCf,
 * [16] lost (matches below min_match_frac), [17] Cholesky failed, [18] iterations done so far, [19] valid_opt
 * count, [20] valid_kf count, [21] number of distinct matched frame pixels, [22] dist_quantile_q-quantile of the
 * match displacement over valid_opt (torch.quantile semantics), [23] cost of the last linearisation, [24] finished
 * (converged, lost or failed), [25] last covariance-filter threshold, [26] fx, [27] fy as the iterations left them
 * (= K's unless optimize_focal), [28..31] zero.
 * optimize_focal != 0 = the reference's --optimize_focal (CameraTracker.py:308-320,367-377; geometry.py:110-112): every
 * iteration rebuilds the matched frame points from their pixel and depth with the current focal, the system gets an 8th
 * unknown shared by fx and fy (Jacobian column exactly as project_calib writes it), the pose takes tau[:7], both focal
 * lengths take tau[7].  K itself is never written: the caller copies result[26..27] into its K (the reference updates
 * self.K_slam in place).
 * A call enqueues num_iters Gauss-Newton iterations (they become no-ops once converged) and never synchronises.
 * resume = 0: full call (prepare, gather, statistics, init, iterations).  resume = 1: only further iterations on the
 * state left in the SAME workspace by the previous call (same inputs): the host reads result[24] and continues in
 * chunks until finished or the reference's max_iters (50) is reached -- a no-op launch still costs ~4 us of GPU time,
 * so enqueueing all 50 iterations up front would triple the cost of a frame that converges in three.
 * dbg_* (optional, resume = 0): constrained frame points [n,3], local variances [n,3], valid_opt [n] bytes, the
 * summed accumulators of iteration 0 [36] (28 lower-triangle H, 7 J^T e, cost; [45] = 36 + 8 + 1 with optimize_focal). */
int64_t adk_track_workspace_bytes(int height, int width);
int adk_track_frame(int height, int width, const float* K, const float* Xf_canon, const float* Cf, float inv_Nf,
                    const float* Qf, const float* Xk_canon, const float* Ck, float inv_Nk, const float* Qk,
                    const int64_t* idx_f2k, const uint8_t* valid_match, const float* T_WCf, const float* T_WCk,
                    float sigma_pixel, float sigma_depth, float huber_k, float C_conf, float Q_conf,
                    float min_match_frac, int pixel_border, float depth_eps, float rel_error, float delta_norm,
                    int num_iters, int covariance_filter, int optimize_focal, float dist_quantile_q, int resume, float* result,
                    float* dbg_Xc, float* dbg_var, uint8_t* dbg_valid_opt, float* dbg_acc0, void* workspace,
                    int64_t workspace_bytes, adk_stream_t stream);
/* Point fusion after a successful track (CameraTracker.py:136-141 + ImageFrame.update_pointmap, ImageFrame.py:30-48):
 * X_canon = (C X_canon + Ckf (T_CkCf Xkf)) / (C + Ckf), C += Ckf, in place; a no-op when result[16] or result[17]
 * is set (decided on the device: no host read needed in between). */
int adk_track_fuse_pointmap(int64_t n, const float* result, const float* Xkf, const float* Ckf, float* X_canon,
                            float* C, adk_stream_t stream);

/* ------------------------------------------------------- prune-and-append
 * Replaces the body of SparseGaussianAdam.add_and_prune (Reconstruct/scene/optimizers.py:163-219; called on
 * every important frame, h3dgsv3.py:938 and :953): `torch.cat([x[valid_mask], extension])` for every
 * parameter, both Adam moments and the per-element learning rate -- ~40 boolean-index host syncs and ~100
 * launches there; one mask scan, one host read (the kept count, to size the outputs) and ONE launch here.
 * adk_compact_plan: keep [N] bytes (torch.bool) -> *n_keep (int64, device) and a plan in the workspace.
 * adk_compact_apply: dst[t] [K+E, words[t]] = concat(src[t][keep] (order preserved), ext[t] or E rows of
 * fill_bits[t]) for t < n_tensors <= 48; words[t] = 4-byte words per row (2 per int64 element); host arrays. */
int64_t adk_compact_workspace_bytes(int64_t N);
int adk_compact_plan(int64_t N, const uint8_t* keep, int64_t* n_keep, void* workspace, int64_t workspace_bytes,
                     adk_stream_t stream);
int adk_compact_apply(int n_tensors, const void* const* src, const void* const* ext, void* const* dst,
                      const uint32_t* fill_bits, const int* words, int64_t N, int64_t E, const uint8_t* keep,
                      const int64_t* n_keep, const void* workspace, adk_stream_t stream);

/* ------------------------------------------------------------ densification: the image-space chain of add_new_gaussians
 * Replaces the torch / MIOpen operator chain SceneModel.add_new_gaussians runs per LoD level of every important frame --
 * Reconstruct/scene/scene_models/h3dgsv3.py:775-903 with Reconstruct/utils.py:93-108 (get_lapla_norm), :121-131 (RGB2SH,
 * inverse_sigmoid), :188-216 (depth2points, sample) -- see artdeco_amd/csrc/densify.hip.  All maps are row-major fp32.
 *
 * adk_densify_proba: proba [h,w] = scaler * clamp(disc (*) |sum_c Laplacian(resize(src))|, 0, 1), the whole of
 *   `get_lapla_norm(F.interpolate(src, (h, w), bilinear, align_corners=True), disc_kernel) * init_proba_scaler` (h3dgsv3.py:781-782,
 *   789-795) in one pass; pool2 != 0 applies F.avg_pool2d(src, 2) first (:776).  src [channels<=3, src_h, src_w]; disc [7,7] (the
 *   scene model's disc_kernel, :209-220); img_out [channels,h,w] or NULL (the resized image; f_dc is read from it, :853). */
int adk_densify_proba(const float* src, int channels, int src_h, int src_w, int pool2, int h, int w, const float* disc,
                      float scaler, float* img_out, float* proba, adk_stream_t stream);

/* out[0] = min(cap, torch.quantile(x, q)) on the device (h3dgsv3.py:815 reads it back with .item()): rank_below = floor(q (n-1)),
 * weight = q (n-1) - rank_below, both computed by the caller in fp32 as torch does; exact order statistics (radix select). */
int adk_densify_quantile(const float* x, int64_t n, int64_t rank_below, float weight, float cap, float* out, adk_stream_t stream);

/* mask [h*w] (bytes) = rnd < (init_proba - penalty) * ratio  and  sample(conf) >= 0  and  sample(depth) > depth_min[0]
 * (h3dgsv3.py:798-821); penalty may be NULL (empty map).  depth_map / conf_map [map_h, map_w] = keyframe.point_map[:, 2] /
 * keyframe.mono_depth_conf; sample_w / sample_h = keyframe.width // lod, keyframe.height // lod (utils.py:203-216). */
int adk_densify_select(int h, int w, const float* init_proba, const float* penalty, const float* rnd, float ratio,
                       const float* depth_map, const float* conf_map, int map_h, int map_w, int sample_w, int sample_h,
                       const float* depth_min, uint8_t* mask, adk_stream_t stream);

/* The attributes of the selected pixels in boolean-index (row-major) order (h3dgsv3.py:847-872, 891): xyz [L,3] world points,
 * f_dc [L,3], scaling [L,3] (log), opacity [L] (logit), d_max [L].  plan_workspace: what adk_compact_plan(h*w, mask, ...) left
 * (its n_keep is L).  focal / centre_x / centre_y / map_width: the scene model's f, centre, width; Rt [4,4] = Keyframe.get_Rt(),
 * approx_centre [3] (device). */
int adk_densify_emit(int h, int w, int lod, const uint8_t* mask, const void* plan_workspace, const float* img_lod,
                     const float* init_proba, const float* depth_map, const float* conf_map, int map_h, int map_w, int sample_w,
                     int sample_h, float focal, float centre_x, float centre_y, int map_width, const float* Rt, const float* approx_centre,
                     float* xyz, float* f_dc, float* scaling, float* opacity, float* d_max, adk_stream_t stream);

/* valid_gs_mask of h3dgsv3.py:894-903: sigmoid(opacity_raw) > 0.05 and focal * max(exp(scaling_raw)) / |xyz - centre| < map_width / 2. */
int adk_prune_mask(int64_t N, const float* opacity_raw, const float* scaling_raw, const float* xyz, const float* centre, float focal,
                   int map_width, uint8_t* mask, adk_stream_t stream);

/* SceneModel.rigid_transform_gs (h3dgsv3.py:956-966 -> Reconstruct/utils.py:28-62): xyz_out = R xyz + t, quat_out =
 * quaternion(R * R(quat)) (wxyz) with (R | t) = delta[ids[i]]; delta [n_keyframes,4,4] = new_c2w @ inverse(old_c2w) per keyframe. */
int adk_rigid_transform(int64_t N, const int64_t* ids, int64_t n_keyframes, const float* delta, const float* xyz, const float* quat,
                        float* xyz_out, float* quat_out, adk_stream_t stream);

/* ------------------------------------------------------ the mapper's optimisation step as ONE call
 * Forward, loss and backward of SceneModel.optimization_step (Reconstruct/scene/scene_models/h3dgsv3.py:418-455: render at the
 * keyframe's level -> exposure / clamp / L1 + fused-SSIM + inverse-depth loss -> loss.backward()) with every stage above enqueued by
 * one host call: pose6d_fwd, lod_params_fwd, project_fwd, tile-local binning (count | scatter | sort), raster_fwd, visibility masks,
 * photometric_fwd, fused_ssim_fwd_sums, photometric_loss_sums, fused_ssim_bwd, photometric_bwd, raster_bwd, project_bwd (with the SH
 * colours' sparse-Adam step when color_adam != 0), lod_params_bwd, pose6d_bwd (the masks, two zero fills and the pose backward ride in a
 * neighbouring launch each, and lod_params_fwd + project_fwd are one kernel: same arithmetic, bit-identical results, four launches less).
 * The optimiser steps that follow (Keyframe.step,
 * SparseGaussianAdam.step: h3dgsv3.py:456-462) stay with the caller: they belong to ARTDECO's optimiser objects and consume the
 * gradient buffers this call filled (adk_adam_update_multi_betas, one launch).
 *
 * Why a call of its own: the step has ONE host wait (the intersection count sizes the tile lists).  Everything a host does between
 * that wait and the forward rasteriser's launch is time the GPU idles, and a Python host spends 0.55 ms per step on ~1 200 calls
 * around the 20 C entries; here the wait, the sort's launch and the rasteriser's launch are consecutive statements.
 *
 * Every buffer is the caller's (a "plan": allocated once per (N, V, width, height), reused by every step); the call allocates
 * nothing on the device.  Field kinds: P = device pointer, L = int64_t, I = int32_t, F = float; pointers come first so that the
 * layout has no padding (bindings build their struct from this list: artdeco_amd/native_step.py parses it).
 *   inputs       r6 [3,2], t [3], exposure [3,4], bg [3], gt [3,H,W], mono [1,H,W] (mono inverse depth), rdk [H,W], Kmat [3,3],
 *                unit_grad [1] (the 1.0 loss.backward() starts from);
 *   Gaussians    xyz .. d_max as adk_lod_params_fwd; f_dc [N,1,3], f_rest [N,sh_K-1,3] and, for color_adam, their moments exp_avg_* / exp_avg_sq_* and
 *                0-dim learning rates lr_dc / lr_rest as adk_project_bwd_adam (adam_b1 / adam_b2 / adam_eps);
 *   per step     loss [1], invdepth [H,W] (Keyframe.latest_invdepth) -- the two results a caller keeps beyond the step;
 *   plan         everything else: intermediates (viewmat [16] .. last_ids), masks vis [N] / gvis [V] (bytes), loss scratch, and the
 *                gradients v_* the optimisers read: v_means (= gradient of xyz, LoD fade term included), v_opacity_raw, v_scaling_raw,
 *                v_rotation, v_local_feat, v_global_feat, v_mlp [1287] = dW1 | db1 | dW2 | db2, v_exposure [12], v_r6 [6], v_t [3];
 *                v_dc / v_rest only without color_adam.  cam_grad: 128 bytes (16 doubles, ABI v19) zeroed once by the caller (left zeroed by the call);
 *                pairs [isect_capacity] 8 B keys, pairs2 [isect_capacity] (the long tile lists' ping-pong buffer, adk_bin_local_sort_long_t; may be
 *                NULL), flatten_ids [isect_capacity]; bin_table: adk_bin_local_workspace_bytes_t + 256 bytes.
 * Return: ADK_OK; ADK_STEP_ECAPACITY when the frame has more intersections than isect_capacity (out->n_isects says how many: grow
 * pairs / pairs2 / flatten_ids and call again); ADK_STEP_EROUTE when a tile list is too long for the tile-local sort (out->max_tile above
 * adk_bin_local_sort_long_max(), or above 8 192 with pairs2 == NULL: use the per-stage calls with the global route).  Both are decided
 * BEFORE anything the caller owns has been modified.  Any other
 * negative value: the failing stage's own code, out->stage = its index in ADK_MAPPER_STAGES.
 * scaling_reg_factor (ABI v19): h3dgsv3.py:443-449's regulariser, loss += factor * mean over the LoD-selected Gaussians of the product of their
 * post-mlp_cov scales -- three small launches inside the call when it is not 0 (run.sh: 0); reg_ws: 32 bytes (4 doubles) zeroed once by the
 * caller (left zeroed), may be NULL while the factor is 0.
 * ssim_grad_scale = -lambda_dssim / (3 W H), formed by the caller in double precision as the per-stage binding does (adk_fused_ssim_bwd's
 * dL_scalar).  time_mask: bit s set = bracket stage s with a pair of events (adk_mapper_step_timings folds and frees them). */
#define ADK_STEP_ECAPACITY (-16)
#define ADK_STEP_EROUTE (-17)
#define ADK_MAPPER_STEP_FIELDS(P, L, I, F) \
    P(r6) P(t) P(exposure) P(bg) P(gt) P(mono) P(rdk) P(Kmat) P(unit_grad) \
    P(xyz) P(opacity_raw) P(scaling_raw) P(rotation) P(local_feat) P(global_feat) P(W1) P(b1) P(W2) P(b2) P(cls_id) P(d_max) \
    P(f_dc) P(f_rest) P(exp_avg_dc) P(exp_avg_sq_dc) P(exp_avg_rest) P(exp_avg_sq_rest) P(lr_dc) P(lr_rest) \
    P(loss) P(invdepth) \
    P(viewmat) P(opac) P(scale) P(quat) P(sel) P(rec) P(radii) P(depth_keys) P(gauss_ids) P(tiles_per_gauss) \
    P(offsets) P(bin_stats) P(bin_table) P(pairs) P(pairs2) P(flatten_ids) \
    P(render_colors) P(render_alphas) P(final_T) P(last_ids) P(vis) P(gvis) \
    P(image) P(gt_used) P(dm) P(parts) P(ssim_sums) P(photo_ws) P(v_img) P(v_col) P(v_alpha) P(v_exposure) \
    P(v_rec) P(v_means) P(v_quats) P(v_scales) P(v_opac) P(v_dc) P(v_rest) P(cam_grad) P(v_viewmat) \
    P(v_opacity_raw) P(v_scaling_raw) P(v_rotation) P(v_local_feat) P(v_global_feat) P(v_mlp) P(lod_ws) P(v_r6) P(v_t) P(reg_ws) \
    L(isect_capacity) L(bin_table_bytes) L(lod_ws_bytes) L(photo_ws_bytes) L(n_ssim_sums) \
    I(N) I(V) I(width) I(height) I(tile_px_w) I(tile_px_h) I(sh_K) I(sh_degree) I(mask_outliers) I(color_adam) I(pose_grad) I(time_mask) I(reserved) \
    F(eps2d) F(near_plane) F(far_plane) F(radius_clip) F(lambda_dssim) F(depth_weight) F(adam_b1) F(adam_b2) F(adam_eps) F(ssim_grad_scale) F(scaling_reg_factor)
#define ADK_MAPPER_STAGES(S) \
    S(lod_params_fwd) S(project_fwd) S(bin_count) S(bin_scatter) S(bin_sort) S(raster_fwd) S(photometric_fwd) S(ssim_fwd) \
    S(photometric_loss) S(ssim_bwd) S(photometric_bwd) S(raster_bwd) S(project_bwd) S(lod_params_bwd)
#define ADK_MAPPER_N_STAGES 14
#define ADK_FIELD_P(n) void* n;
#define ADK_FIELD_L(n) int64_t n;
#define ADK_FIELD_I(n) int32_t n;
#define ADK_FIELD_F(n) float n;
typedef struct AdkMapperStepArgs { ADK_MAPPER_STEP_FIELDS(ADK_FIELD_P, ADK_FIELD_L, ADK_FIELD_I, ADK_FIELD_F) } AdkMapperStepArgs;
/* wait_ns: host nanoseconds spent inside the call's ONE wait (the intersection count): wall time of a step minus this is the host's own work */
typedef struct AdkMapperStepOut { int64_t n_isects; int64_t max_tile; int32_t stage; int32_t reserved; int64_t wait_ns; } AdkMapperStepOut;
int64_t adk_mapper_step_args_bytes(void);   /* sizeof(AdkMapperStepArgs): a binding checks its own layout against it */
int adk_mapper_step(const AdkMapperStepArgs* args, AdkMapperStepOut* out, adk_stream_t stream);
/* Folds the event pairs recorded under time_mask since the last call (waits for them): per stage the sum and the minimum of the
 * elapsed times in ms and the number of pairs; arrays of ADK_MAPPER_N_STAGES.  Returns the total number of pairs folded. */
int64_t adk_mapper_step_timings(double* sum_ms, double* min_ms, int64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* ARTDECO_HIP_H */
