"""ARTDECO's densification as generic torch operators -- the "unchanged host code" side of the important-frame path, for the bench
and the tests (harness, not product).  Same operators in the same order as the reference, so that the results are bit-identical
to its own methods run on the same device (tests/test_densify.py checks that against the real class):

  voxel_labels            SceneModel.update_voxel          h3dgsv3.py:227-316
  densify_from_keyframe   SceneModel.add_new_gaussians     h3dgsv3.py:766-940
  edge_probability        utils.get_lapla_norm             Reconstruct/utils.py:93-108
  bilinear_lookup         utils.sample / make_torch_sampler  :203-216
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as F

SH_C0 = 0.28209479177387814           # Reconstruct/utils.py:119
LEVEL_KEYS = ("id", "cls_id", "d_max", "xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "local_feat", "global_feat")


# ------------------------------------------------------------------------------------------------ voxel labels
def _voxel_hashes(points, origin, voxel_size):
    cell = torch.floor((points - origin) / voxel_size).long()
    extent = cell.max(dim=0).values + 1
    weights = torch.tensor([extent[1] * extent[2], extent[2], 1], device=points.device)
    return (cell * weights).sum(dim=1)


def voxel_labels(new_xyz, xyz, cls_id, voxel_size, scatter_max):
    """Majority class per occupied voxel for the existing points, the same label (or a fresh one per new voxel) for the new ones."""
    n_old = xyz.shape[0]
    if n_old == 0:                                                  # cold start: every occupied voxel is a new class
        hashes = _voxel_hashes(new_xyz, new_xyz.min(dim=0).values, voxel_size)
        distinct, which = torch.unique(hashes, return_inverse=True)
        return which.unsqueeze(-1), distinct.shape[0]
    labels = cls_id.squeeze(-1)
    top_label = labels.max().item()
    everything = torch.cat([xyz, new_xyz], dim=0)
    hashes = _voxel_hashes(everything, everything.min(dim=0).values, voxel_size)
    old_hash, new_hash = hashes[:n_old], hashes[n_old:]
    voxels, voxel_of_point = torch.unique(old_hash, return_inverse=True)
    span = top_label + 1
    pairs, pair_count = torch.unique(voxel_of_point * span + labels, return_counts=True)
    _, winner = scatter_max(pair_count, pairs // span)
    voxel_label = (pairs % span)[winner]
    relabelled = voxel_label[voxel_of_point].unsqueeze(-1)
    slot = torch.searchsorted(voxels, new_hash).clamp(max=voxels.shape[0] - 1)
    known = voxels[slot] == new_hash
    new_label = torch.zeros(new_xyz.shape[0], dtype=torch.long, device=new_xyz.device)
    if known.any():
        new_label[known] = voxel_label[slot[known]]
    opened = 0
    if (~known).any():
        fresh, rank = torch.unique(new_hash[~known], return_inverse=True)
        opened = fresh.shape[0]
        new_label[~known] = rank + top_label + 1
    return relabelled, new_label.unsqueeze(-1), opened


# ------------------------------------------------------------------------------------------------ image-space pieces
def edge_probability(img, disc, device):
    """|Laplacian| summed over the channels, image border cleared, averaged over the disc, clipped to [0, 1]."""
    stencil = torch.tensor([[0, 1, 0], [1, -4, 1], [0, 1, 0]], device=device, dtype=torch.float32).unsqueeze(0).unsqueeze(0)
    response = F.conv2d(img[None], stencil.repeat(1, img.shape[0], 1, 1), padding="same")
    strength = torch.linalg.vector_norm(response, ord=1, dim=1, keepdim=True)
    strength[..., :, 0] = 0
    strength[..., :, -1] = 0
    strength[..., 0, :] = 0
    strength[..., -1, :] = 0
    return F.conv2d(strength, disc, padding="same")[0, 0].clamp(0, 1)


def resize(img, h, w):
    return F.interpolate(img[None], (h, w), mode="bilinear", align_corners=True)[0]


def bilinear_lookup(grid_map, uv, width, height):
    """grid_map [1,C,Hs,Ws] read at pixel positions uv [..., 2] of a width x height raster (align_corners sampling)."""
    at = uv.clone()
    at[..., 0] = at[..., 0] * (2.0 / (width - 1)) - 1.0
    at[..., 1] = at[..., 1] * (2.0 / (height - 1)) - 1.0
    return F.grid_sample(grid_map, at, mode="bilinear", align_corners=True)


def _level_candidates(scene, keyframe, keyframe_id, pooled, lod):
    """(resized image, image probability, sample mask, uv, depth, confidence) of one LoD level, before the validity filter."""
    dev = scene.device
    h, w = scene.height // lod, scene.width // lod
    img = resize(pooled, h, w)
    p_img = edge_probability(img, scene.disc_kernel, device=dev)
    p_map = 0
    if scene.xyz.shape[0] > 0:
        pkg = scene.render_from_id(keyframe_id)
        seen = resize(pkg["render"], h, w)
        _unused_depth = 1 / resize(pkg["invdepth"], h, w)[0].clamp_min(1e-8)      # computed and dropped by the reference too (:790)
        p_map = edge_probability(seen, scene.disc_kernel, device=dev)
    p_img *= scene.init_proba_scaler
    p_map *= scene.init_proba_scaler
    chosen = torch.rand_like(p_img) < (p_img - p_map) * scene.gs_add_ratio
    uv = scene.uvs[lod][chosen]
    where = uv[None, None, ...]
    depth = bilinear_lookup(keyframe.point_map[:, 2:], where, keyframe.width // lod, keyframe.height // lod)[0, 0, 0]
    conf = bilinear_lookup(keyframe.mono_depth_conf, where, keyframe.width // lod, keyframe.height // lod)[0, 0, 0]
    return img, p_img, chosen, uv, depth, conf


def _new_gaussians(scene, keyframe, img, p_img, chosen, uv, depth, conf, lod):
    """World points and initial attributes of the surviving samples of one level (h3dgsv3.py:847-891)."""
    dev = scene.device
    focal, centre = scene.f / lod, scene.centre / lod
    rays = torch.cat([(uv[..., :2] - centre) / focal, torch.ones_like(uv[..., 0:1])], dim=-1)
    world = (depth.unsqueeze(-1) * rays - keyframe.get_t()) @ keyframe.get_R()
    f_dc = (img[:, chosen].permute(1, 0).unsqueeze(1) - 0.5) / SH_C0
    spacing = 1 / (torch.sqrt(p_img[chosen]))
    spacing.clamp_(1, scene.width / 10)
    spacing.mul_(1 / scene.f)
    spacing *= torch.linalg.vector_norm(world - keyframe.approx_centre[None], dim=-1)
    scaling = torch.log(lod * spacing.clamp(1e-6, 1e6)).unsqueeze(-1).repeat(1, 3)
    alpha = torch.ones(f_dc.shape[0], 1, device=dev)
    alpha[: uv.shape[0]] *= 0.2 * conf[..., None]
    opacity = torch.log(alpha / (1 - alpha))
    n = f_dc.shape[0]
    rotation = torch.zeros((n, 4), device=dev)
    rotation[:, 0] = 1
    return {"xyz": world, "f_dc": f_dc, "scaling": scaling, "opacity": opacity, "rotation": rotation,
            "f_rest": torch.zeros(n, (scene.max_sh_degree + 1) * (scene.max_sh_degree + 1) - 1, 3, device=dev),
            "local_feat": torch.zeros((n, scene.local_feat_dim), device=dev).float(),
            "d_max": (depth.unsqueeze(-1) * lod).to(dev)}


def _prune_mask(scene, keyframe):
    if scene.xyz.shape[0] == 0:
        return torch.ones(0, device=scene.device, dtype=torch.bool)
    keep = scene.opacity[:, 0] > 0.05
    distance = torch.linalg.vector_norm(scene.xyz - keyframe.approx_centre[None], dim=-1)
    keep *= scene.f * scene.scaling.max(dim=-1)[0] / distance < 0.5 * scene.width
    return keep


def densify_from_keyframe(scene, keyframe_id=-1):
    keyframe = scene.keyframes[keyframe_id]
    if keyframe.is_test:
        return
    dev = scene.device
    pooled = F.avg_pool2d(keyframe.image_pyr[0], 2)
    per_level = {k: [] for k in LEVEL_KEYS}
    prune = None
    for lod in scene.lods:
        img, p_img, chosen, uv, depth, conf = _level_candidates(scene, keyframe, keyframe_id, pooled, lod)
        floor = min(1e-2, torch.quantile(keyframe.point_map[:, 2], 0.02).item())
        usable = (conf >= 0) * (depth > floor)
        chosen[chosen.clone()] = usable
        g = _new_gaussians(scene, keyframe, img, p_img, chosen, uv[usable], depth[usable], conf[usable], lod)
        if len(scene.xyz) > 0:
            relabelled, g["cls_id"], opened = scene.update_voxel(g["xyz"], scene.xyz, scene.cls_id, scene.voxel_size)
            scene.gaussian_params["cls_id"]["val"] = relabelled
        else:
            g["cls_id"], opened = scene.update_voxel(g["xyz"], scene.xyz, scene.cls_id, scene.voxel_size)
        g["global_feat"] = torch.zeros((opened, scene.global_feat_dim), device=dev)
        prune = _prune_mask(scene, keyframe)
        keyframe_id = len(scene.keyframes) - 1 if keyframe_id == -1 else keyframe_id
        g["id"] = torch.full((len(g["xyz"]), 1), keyframe_id, device=dev, dtype=torch.long)
        for k in LEVEL_KEYS:
            per_level[k].append(g[k])
    extension = {k: torch.concat(v, dim=0) for k, v in per_level.items()}
    lock = getattr(scene, "lock", None)
    with (lock if lock is not None else contextlib.nullcontext()):
        scene.optimizer.add_and_prune(extension, prune)
    scene.weed_out_gaussians()
