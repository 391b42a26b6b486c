"""Host-side mirror of ARTDECO's mapper hot path, used by bench.py, smoke() and the tests.

ARTDECO's own host code (Reconstruct/scene/scene_models/h3dgsv3.py, scene/optimizers.py,
scene/keyframe.py) cannot travel to the GPU box and must not be copied, so this module restates
JUST the hot-path operator interface with the same names, argument meaning and call order:

  MapperScene.render(width, height, view_matrix, bg)      <- SceneModel.render           h3dgsv3.py:617-700
  MapperScene.render_from_id(...)                         <- SceneModel.render_from_id   h3dgsv3.py:595-615
  MapperScene.optimization_step(is_important)             <- SceneModel.optimization_step h3dgsv3.py:401-469
  SparseGaussianAdam.step(vis, N, gvis, Ng) / BaseAdam    <- scene/optimizers.py:17-161
  Keyframe (6D pose + t + 3x4 exposure, own BaseAdam)     <- scene/keyframe.py:103-125, 150-155, 186-191
  StreamKeyframe (pyramids from image + point map + conf)  <- scene/keyframe.py:26-126 (what run_system.py:177-192 constructs)
  MapperScene.add_keyframe / rigid_transform_gs            <- h3dgsv3.py:981-1009, 956-966
  MapperScene.add_new_gaussians / update_voxel             <- h3dgsv3.py:766-940, 227-316 (harness/densify_mirror.py)

Every native call goes through the drop-in modules exactly as the reference's imports do
(`gsplat.rendering.rasterization`, `fused_ssim`, `diff_gaussian_rasterization.adamUpdate*`), so what
is timed here is the path `run_system.py` would exercise.  One deliberate deviation: the radial
decay weights are cached per resolution (the reference rebuilds them on the CPU and uploads them
every step, h3dgsv3.py:431 -> utils.py:818-827).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

import artdeco_amd

artdeco_amd.install_dropins()

import gsplat  # noqa: E402  (drop-in)
from diff_gaussian_rasterization import adamUpdate, adamUpdateBasic  # noqa: E402
from fused_ssim import fused_ssim  # noqa: E402
from torch_scatter import scatter_max  # noqa: E402  (drop-in; h3dgsv3.py:35)


# ------------------------------------------------------------------------------- optimisers
class BaseAdam:
    """scene/optimizers.py:17-57: dense Adam over {name: {"val","lr"}} via adamUpdateBasic."""

    @torch.no_grad()
    def __init__(self, params, betas=(0.9, 0.999), eps=1e-15):
        self.params, self.betas, self.eps = params, betas, eps
        for p in self.params.values():
            if "exp_avg" not in p:
                p["exp_avg"] = torch.zeros_like(p["val"], memory_format=torch.preserve_format)
                p["exp_avg_sq"] = torch.zeros_like(p["val"], memory_format=torch.preserve_format)

    def zero_grad(self):
        for p in self.params.values():
            p["val"].grad = None

    @torch.no_grad()
    def step(self):
        for p in self.params.values():
            v = p["val"]
            if v.grad is None:
                continue
            adamUpdateBasic(v, v.grad, p["exp_avg"], p["exp_avg_sq"], p["lr"], self.betas[0], self.betas[1], self.eps)


_NO_OPT = ("id", "cls_id", "d_max")


def _add_and_prune(self, extension_tensors, valid_mask):
    """scene/optimizers.py:163-219 restated: prune rows by `valid_mask`, append the extension rows, for every parameter,
    its two moments and (keys in lr_dict) its per-element learning rate; `global_feat` is extended but never pruned."""
    for key, param in self.params.items():
        if key not in extension_tensors:
            continue
        ext = extension_tensors[key]
        empty = ext.numel() == 0 or ext.dim() == 0
        if key == "global_feat":
            param["val"] = param["val"].detach().contiguous() if empty else torch.cat([param["val"].detach(), ext], dim=0).contiguous()
            param["val"].requires_grad = True
            param["exp_avg"] = torch.cat([param["exp_avg"], torch.zeros_like(ext)], dim=0).contiguous()
            param["exp_avg_sq"] = torch.cat([param["exp_avg_sq"], torch.zeros_like(ext)], dim=0).contiguous()
            if key in self.lr_dict:
                param["lr"] = torch.cat([param["lr"], torch.ones_like(ext) * self.lr_dict[key]["lr_init"]], dim=0).contiguous()
            continue
        kept = param["val"].detach()[valid_mask]
        param["val"] = kept.contiguous() if empty else torch.cat([kept, ext], dim=0).contiguous()
        if key in ("id", "cls_id", "d_max"):
            continue
        param["val"].requires_grad = True
        param["exp_avg"] = torch.cat([param["exp_avg"][valid_mask], torch.zeros_like(ext)], dim=0).contiguous()
        param["exp_avg_sq"] = torch.cat([param["exp_avg_sq"][valid_mask], torch.zeros_like(ext)], dim=0).contiguous()
        if key in self.lr_dict:
            param["lr"] = torch.cat([param["lr"][valid_mask], torch.ones_like(ext) * self.lr_dict[key]["lr_init"]], dim=0).contiguous()


class SparseGaussianAdam(BaseAdam):
    """scene/optimizers.py:59-161: visibility-gated Adam; lr is a 0-dim device tensor, or a
    per-element tensor for keys in lr_dict (xyz), python floats for the mlp_* entries."""

    def __init__(self, params, betas=(0.9, 0.999), eps=1e-15, lr_dict=None, device="cuda:0"):
        super().__init__(params, betas, eps)  # the reference keeps id / cls_id / d_max in the same dict (optimizers.py:61-68)
        self.all_params = params
        self.lr_dict = lr_dict or {}
        for key, p in self.params.items():
            if key in _NO_OPT or key.startswith("mlp"):
                continue
            if key not in self.lr_dict:
                p["lr"] = torch.tensor(p["lr"], dtype=torch.float, device=device)
            else:
                p["lr"] = torch.ones_like(p["val"]) * self.lr_dict[key]["lr_init"]

    add_and_prune = torch.no_grad()(_add_and_prune)

    @torch.no_grad()
    def step(self, visibility, N, global_visibility, N_global):
        b1, b2 = self.betas
        for key, p in self.params.items():
            if key in _NO_OPT:
                continue
            v = p["val"]
            if v.grad is None:
                continue
            if key.startswith("mlp"):
                adamUpdateBasic(v, v.grad, p["exp_avg"], p["exp_avg_sq"], p["lr"], b1, b2, self.eps)
                if key in self.lr_dict:
                    p["lr"] = max(p["lr"] * self.lr_dict[key]["lr_decay"], self.lr_dict[key]["lr_init"] * 0.1)
                continue
            vis, n = (global_visibility, N_global) if key == "global_feat" else (visibility, N)
            adamUpdate(v, v.grad, p["exp_avg"], p["exp_avg_sq"], vis, p["lr"], b1, b2, self.eps, n, v.numel() // n)
            if key in self.lr_dict:
                p["lr"][vis] *= self.lr_dict[key]["lr_decay"]
                p["lr"].clamp_min_(self.lr_dict[key]["lr_init"] * 0.1)


# ------------------------------------------------------------------------------- keyframe
def sixD2mtx(r):
    """Reconstruct/utils.py:223-229."""
    b1 = r[..., 0]
    b1 = b1 / torch.norm(b1, dim=-1, keepdim=True)
    b2 = r[..., 1] - torch.sum(b1 * r[..., 1], dim=-1, keepdim=True) * b1
    b2 = b2 / torch.norm(b2, dim=-1, keepdim=True)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack([b1, b2, b3], dim=-1)


def radial_decay_kernel(H, W, sigma=0.5):
    """Reconstruct/utils.py:818-827."""
    y = torch.linspace(-1, 1, steps=H)
    x = torch.linspace(-1, 1, steps=W)
    yy, xx = torch.meshgrid(y, x, indexing="ij")
    return torch.exp(-(xx ** 2 + yy ** 2) / (2 * sigma ** 2))


class Keyframe:
    """The optimisable part of scene/keyframe.py: 6D rotation + translation + 3x4 exposure."""

    def __init__(self, image, mono_idepth, Rt, device, lr_poses=1e-4, lr_exposure=1e-3, depth_loss_weight=1e-2,
                 depth_loss_weight_decay=0.9, is_test=False):
        self.device = device
        self.image_pyr = [image]            # [3,H,W] at the training level
        self.idepth_pyr = [mono_idepth]     # [1,H,W]
        self.pyr_lvl = 0
        self.is_test = is_test
        self.depth_loss_weight = depth_loss_weight
        self.depth_loss_weight_decay = depth_loss_weight_decay
        self.rW2C = nn.Parameter(Rt[:3, :2].clone().contiguous())
        self.tW2C = nn.Parameter(Rt[:3, 3].clone().contiguous())
        self.exposure = nn.Parameter(torch.eye(3, 4, device=device))
        params = {"rW2C": {"val": self.rW2C, "lr": lr_poses}, "tW2C": {"val": self.tW2C, "lr": lr_poses}}
        if not is_test:
            params["exposure"] = {"val": self.exposure, "lr": lr_exposure}
        self.optimizer = BaseAdam(params, betas=(0.8, 0.99))
        self.latest_invdepth = None

    def get_Rt(self):
        Rt = torch.eye(4, device=self.device)
        Rt[:3, :3] = sixD2mtx(self.rW2C)
        Rt[:3, 3] = self.tW2C
        return Rt

    def get_mono_idepth(self, lvl=0):
        return self.idepth_pyr[lvl]

    def zero_grad(self):
        self.optimizer.zero_grad()

    @torch.no_grad()
    def step(self):
        self.optimizer.step()
        self.depth_loss_weight *= self.depth_loss_weight_decay


def _pyramid(base, levels):
    out = [base]
    for _ in range(levels - 1):
        out.append(F.avg_pool2d(out[-1], 2))
    return out


class StreamKeyframe(Keyframe):
    """What run_system.py:177-192 constructs for every mapped frame (scene/keyframe.py:26-126): the frame's image, the inverse depth
    and confidence of its dense point map brought to the image size (bilinear, align_corners), `pyr_levels` 2x2-average pyramids of
    all three, the 6D pose + exposure parameters (exposure inherited from the previous keyframe) and their Adam (no pose learning
    rate for the first keyframe, a fixed one for test frames, which carry no exposure parameter)."""

    def __init__(self, image, Rt, point_map, point_conf, f, device, *, index=0, prev_kf=None, is_test=False, pyr_levels=2,
                 lr_poses=1e-4, lr_exposure=1e-3, depth_loss_weight_init=1e-2, depth_loss_weight_decay=0.9):
        self.device = torch.device(device)
        self.is_test, self.index, self.f = is_test, index, f
        self.height, self.width = image.shape[1], image.shape[2]
        self.latest_invdepth = None
        self.num_steps = 0
        self.depth_loss_weight, self.depth_loss_weight_decay = depth_loss_weight_init, depth_loss_weight_decay
        # dense geometry from the SLAM side, at ITS resolution
        self.point_map = point_map.permute(2, 0, 1)[None]                                   # [1,3,Hs,Ws]
        z = self.point_map[:, 2:, ...]
        self.mono_depth_conf = point_conf[None, None, ...].to(torch.float32)                # [1,1,Hs,Ws]
        inverse_z = torch.where(z != 0, 1.0 / (z + 1e-4), 1e4)
        to_image = lambda m: F.interpolate(m, (self.height, self.width), mode="bilinear", align_corners=True)[0]
        self.idepth_pyr = _pyramid(to_image(inverse_z), pyr_levels)
        self.idepth_conf_pyr = _pyramid(to_image(self.mono_depth_conf), pyr_levels)
        self.image_pyr = _pyramid(image, pyr_levels)
        self.pyr_lvl = pyr_levels - 1
        self.centre = torch.tensor([(self.width - 1) / 2, (self.height - 1) / 2]).to(self.device)
        # optimisable state
        self.rW2C = nn.Parameter(Rt[:3, :2].clone().contiguous())
        self.tW2C = nn.Parameter(Rt[:3, 3].clone().contiguous())
        self.exposure = nn.Parameter(torch.eye(3, 4, device=self.device) if prev_kf is None else prev_kf.exposure.clone().detach())
        pose_lr = 1e-4 if is_test else (0 if index == 0 else lr_poses)
        params = {"rW2C": {"val": self.rW2C, "lr": pose_lr}, "tW2C": {"val": self.tW2C, "lr": pose_lr}}
        if not is_test:
            params["exposure"] = {"val": self.exposure, "lr": lr_exposure}
        self.optimizer = BaseAdam(params, betas=(0.8, 0.99))
        self.approx_centre = -Rt[:3, :3].T @ Rt[:3, 3]

    def get_R(self):
        return sixD2mtx(self.rW2C)

    def get_t(self):
        return self.tW2C

    def set_Rt(self, Rt):
        self.rW2C.data.copy_(Rt[:3, :2])
        self.tW2C.data.copy_(Rt[:3, 3])
        self.approx_centre = -Rt[:3, :3].T @ Rt[:3, 3]


def quaternion_to_rotation_matrix(q):
    """kornia.geometry.conversions.quaternion_to_rotation_matrix for (w, x, y, z) [UPSTREAM kornia, a pip dependency that is
    not in /root/reference; restated from its published definition: normalise, then the standard matrix]."""
    q = F.normalize(q, p=2.0, dim=-1, eps=1e-12)
    w, x, y, z = q.unbind(-1)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = torch.ones_like(w)
    return torch.stack([one - (tyy + tzz), txy - twz, txz + twy, txy + twz, one - (txx + tzz), tyz - twx,
                        txz - twy, tyz + twx, one - (txx + tyy)], dim=-1).view(*q.shape[:-1], 3, 3)


def rotation_matrix_to_quaternion(R, eps=1e-8):
    """kornia.geometry.conversions.rotation_matrix_to_quaternion -> (w, x, y, z) [UPSTREAM kornia, restated from its published
    definition: the four-branch form selected by the trace and the largest diagonal entry]."""
    m = R.reshape(*R.shape[:-2], 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.unbind(-1)
    trace = m00 + m11 + m22

    def pos():
        sq = torch.sqrt(trace + 1.0 + eps) * 2.0
        return torch.stack([0.25 * sq, (m21 - m12) / sq, (m02 - m20) / sq, (m10 - m01) / sq], -1)

    def c1():
        sq = torch.sqrt(1.0 + m00 - m11 - m22 + eps) * 2.0
        return torch.stack([(m21 - m12) / sq, 0.25 * sq, (m01 + m10) / sq, (m02 + m20) / sq], -1)

    def c2():
        sq = torch.sqrt(1.0 + m11 - m00 - m22 + eps) * 2.0
        return torch.stack([(m02 - m20) / sq, (m01 + m10) / sq, 0.25 * sq, (m12 + m21) / sq], -1)

    def c3():
        sq = torch.sqrt(1.0 + m22 - m00 - m11 + eps) * 2.0
        return torch.stack([(m10 - m01) / sq, (m02 + m20) / sq, (m12 + m21) / sq, 0.25 * sq], -1)

    w2 = torch.where((m11 > m22)[..., None], c2(), c3())
    w1 = torch.where(((m00 > m11) & (m00 > m22))[..., None], c1(), w2)
    return torch.where((trace > 0)[..., None], pos(), w1)


def update_gaussians(old_c2ws, new_c2ws, positions, quaternions):
    """Reconstruct/utils.py:28-62: move every Gaussian by the pose update of the keyframe that created it."""
    delta_T = torch.bmm(new_c2ws, torch.inverse(old_c2ws))
    R, t = delta_T[:, :3, :3], delta_T[:, :3, 3]
    rot_orig = quaternion_to_rotation_matrix(quaternions)
    new_positions = torch.einsum("bij,bj->bi", R, positions) + t
    return new_positions, rotation_matrix_to_quaternion(torch.bmm(R, rot_orig))


# ------------------------------------------------------------------------------- scene
class MapperScene:
    """Parameter dictionary + render + optimisation step of the LoD Gaussian scene model."""

    def __init__(self, width, height, fx, device, *, sh_degree=3, local_feat_dim=16, global_feat_dim=16,
                 lambda_dssim=0.2, eps2d=0.01, rad_decay=math.sqrt(5.0), scaling_reg_factor=0.0,
                 position_lr_init=5e-5, position_lr_decay=1 - 2e-5, feature_lr=5e-3, opacity_lr=0.1,
                 scaling_lr=0.01, rotation_lr=2e-3, feat_lr=4e-3, mlp_cov_lr_init=4e-3, mlp_cov_lr_decay=1 - 2e-5,
                 init_proba_scaler=2.0, gs_add_ratio=1.0, voxel_size=0.1):
        self.device = torch.device(device)
        self.width, self.height = width, height
        self.tanfovx = width / (2 * fx)
        self.tanfovy = height / (2 * fx)
        self.active_sh_degree = sh_degree
        self.lambda_dssim, self.eps2d, self.rad_decay = lambda_dssim, eps2d, rad_decay
        self.scaling_reg_factor = scaling_reg_factor
        self._lr = dict(xyz=position_lr_init, f_dc=feature_lr, f_rest=feature_lr / 20.0, scaling=scaling_lr,
                        rotation=rotation_lr, opacity=opacity_lr, local_feat=feat_lr, global_feat=feat_lr)
        self.lr_dict = {"xyz": {"lr_init": position_lr_init, "lr_decay": position_lr_decay}}
        d = local_feat_dim + global_feat_dim
        self.mlp_cov = nn.Sequential(nn.Linear(d, d), nn.ReLU(True), nn.Linear(d, 7)).to(self.device)
        self.mlp_params = {}
        for n, p in self.mlp_cov.named_parameters():
            name = "mlp_cov_" + n.replace(".", "_")
            self.mlp_params[name] = {"val": p, "lr": mlp_cov_lr_init}
            self.lr_dict[name] = {"lr_init": mlp_cov_lr_init, "lr_decay": mlp_cov_lr_decay}
        self.gaussian_params = {}
        self.keyframes: list[Keyframe] = []
        self._rdk = {}
        self.local_feat_dim, self.global_feat_dim = local_feat_dim, global_feat_dim
        # -- what add_new_gaussians / add_keyframe read (h3dgsv3.py:72-110, 195-225; run.sh / dataloaders/args.py values)
        self.f = float(fx)
        self.centre = torch.tensor([(width - 1) / 2, (height - 1) / 2], device=self.device)   # h3dgsv3.py:75
        self.max_sh_degree = sh_degree
        self.init_proba_scaler, self.gs_add_ratio, self.voxel_size = init_proba_scaler, gs_add_ratio, voxel_size
        self.approx_cam_centres = None
        self.sorted_frame_indices = None
        self.cam_centres = None
        radius = 3
        self.disc_kernel = torch.zeros(1, 1, 2 * radius + 1, 2 * radius + 1).to(self.device)
        y, x = torch.meshgrid(torch.arange(-radius, radius + 1), torch.arange(-radius, radius + 1), indexing="ij")
        x, y = x.to(self.device), y.to(self.device)
        self.disc_kernel[0, 0, torch.sqrt(x ** 2 + y ** 2) <= radius + 0.5] = 1
        self.disc_kernel = self.disc_kernel / self.disc_kernel.sum()
        self.lods = [1, 2, 4, 8]
        self.uvs = dict()
        for lod in self.lods:
            self.uvs[lod] = torch.stack(torch.meshgrid(torch.arange(0, width // lod), torch.arange(0, height // lod), indexing="xy"),
                                        dim=-1).float().to(self.device)

    # -- population ---------------------------------------------------------------------------
    def set_gaussians(self, means, quats, log_scales, opacity_logits, sh, d_max=1e3, n_voxels=None, seed=0):
        """Install N Gaussians (the state add_new_gaussians would have built, h3dgsv3.py:766-940)."""
        dev = self.device
        N = means.shape[0]
        n_vox = n_voxels or max(1, N // 8)
        g = torch.Generator().manual_seed(seed)
        # detach + clone: a caller's tensor must never BECOME the parameter (on the same device .to() returns the tensor
        # itself, so two scenes built from one cloud would share -- and the second would get a non-leaf -- parameters)
        val = lambda t: t.detach().clone().to(dev).contiguous().requires_grad_(True)
        P = self.gaussian_params = {
            "id": {"val": torch.zeros(N, 1, dtype=torch.long, device=dev)},   # the keyframe that created it (h3dgsv3.py:126, 910)
            "cls_id": {"val": torch.randint(0, n_vox, (N, 1), generator=g).to(dev)},
            "d_max": {"val": torch.full((N, 1), float(d_max), device=dev)},
            "xyz": {"val": val(means)},
            "f_dc": {"val": val(sh[:, :1, :])},
            "f_rest": {"val": val(sh[:, 1:, :])},
            "scaling": {"val": val(log_scales)},
            "rotation": {"val": val(quats)},
            "opacity": {"val": val(opacity_logits.reshape(N, 1))},
            "local_feat": {"val": val(torch.zeros(N, self.local_feat_dim))},
            "global_feat": {"val": val(torch.zeros(n_vox, self.global_feat_dim))},
        }
        for k, lr in self._lr.items():
            P[k]["lr"] = lr
        self.optimizer = SparseGaussianAdam({**P, **self.mlp_params}, (0.5, 0.99), lr_dict=self.lr_dict, device=dev)

    def add_keyframe(self, kf: Keyframe):
        """h3dgsv3.py:981-1009 (the part that touches the device: the list of approximate camera centres and the keyframe
        order by distance to the newest one, read back to the host)."""
        self.keyframes.append(kf)
        centre = getattr(kf, "approx_centre", None)
        if centre is None:
            return
        if self.approx_cam_centres is None:
            self.approx_cam_centres = centre[None]
        else:
            self.approx_cam_centres = torch.cat([self.approx_cam_centres, centre[None]], dim=0)
        dist_to_last = torch.linalg.vector_norm(self.approx_cam_centres - centre[None], dim=-1)
        self.sorted_frame_indices = torch.argsort(dist_to_last).cpu()

    def reset_optimizer(self):
        """h3dgsv3.py:317-330: a NEW SparseGaussianAdam over the current parameters (called at the start of every finetune epoch)."""
        for key, pd in self.gaussian_params.items():
            if key in _NO_OPT:
                continue
            if not pd["val"].requires_grad:
                pd["val"].requires_grad = True
        self.optimizer = SparseGaussianAdam({**self.gaussian_params, **self.mlp_params}, (0.5, 0.99), lr_dict=self.lr_dict,
                                            device=self.device)

    # -- reference-named accessors (h3dgsv3.py:332-372) --------------------------------------------
    id = property(lambda s: s.gaussian_params["id"]["val"])
    xyz = property(lambda s: s.gaussian_params["xyz"]["val"])
    f_dc = property(lambda s: s.gaussian_params["f_dc"]["val"])
    f_rest = property(lambda s: s.gaussian_params["f_rest"]["val"])
    scaling = property(lambda s: torch.exp(s.gaussian_params["scaling"]["val"]))
    rotation = property(lambda s: s.gaussian_params["rotation"]["val"])
    opacity = property(lambda s: torch.sigmoid(s.gaussian_params["opacity"]["val"]))
    cls_id = property(lambda s: s.gaussian_params["cls_id"]["val"])
    d_max = property(lambda s: s.gaussian_params["d_max"]["val"])
    local_feat = property(lambda s: s.gaussian_params["local_feat"]["val"])
    global_feat = property(lambda s: s.gaussian_params["global_feat"]["val"])

    # -- render: h3dgsv3.py:617-700 -------------------------------------------------------------------
    def render(self, width, height, view_matrix, bg=None):
        dev = self.device
        bg = torch.zeros(3, device=dev) if bg is None else bg
        xyz = self.xyz
        cam_centre = view_matrix.detach().inverse()[:3, 3].to(dev)
        ob_dist = (xyz - cam_centre).norm(dim=1, keepdim=True)
        selection_mask = (ob_dist < 2 * self.d_max).squeeze(-1)
        alpha_mask = torch.logical_and(ob_dist > self.d_max, ob_dist < 2 * self.d_max).squeeze(-1)
        alpha_ratio = (2 * self.d_max - ob_dist) / self.d_max
        alpha_ratio[~alpha_mask] = 1.0

        xyz = xyz[selection_mask]
        opacity = (self.opacity * alpha_ratio)[selection_mask]
        scaling = self.scaling[selection_mask]
        rotation = self.rotation[selection_mask]
        feats = torch.concat([self.f_dc[selection_mask], self.f_rest[selection_mask]], dim=1)
        fl_x, fl_y = width / (2 * self.tanfovx), height / (2 * self.tanfovy)
        Ks = torch.tensor([[fl_x, 0, width / 2.0], [0, fl_y, height / 2.0], [0, 0, 1]], device=dev)[None]
        local_feat = self.local_feat[selection_mask]
        ids = self.cls_id[selection_mask].squeeze(-1).long()
        scale_rot = self.mlp_cov(torch.cat([self.global_feat[ids], local_feat], dim=1))
        scaling = scaling * torch.sigmoid(scale_rot[:, :3])
        rotation = F.normalize(rotation * scale_rot[:, 3:])

        colors, alphas, meta = gsplat.rendering.rasterization(
            means=xyz, quats=rotation, scales=scaling, opacities=opacity.squeeze(-1), colors=feats,
            viewmats=view_matrix.unsqueeze(0), Ks=Ks, width=width, height=height, render_mode="RGB+D",
            rasterize_mode="classic", absgrad=False, packed=False, sh_degree=self.active_sh_degree, eps2d=self.eps2d)

        rendered_color = colors[..., 0:3].permute([0, 3, 1, 2])
        rendered_depth = colors[..., 3:4].permute([0, 3, 1, 2])
        rendered_alpha = alphas.permute([0, 3, 1, 2])
        rendered_color = rendered_color + (1.0 - rendered_alpha) * bg[None, :, None, None]
        invdepth = 1.0 / rendered_depth
        visible_mask = torch.zeros_like(selection_mask, dtype=torch.bool, device=dev)
        visible_mask[selection_mask.clone()] = meta["radii"][0].max(dim=1).values > 0
        global_visible_mask = torch.zeros(len(self.global_feat), dtype=torch.bool, device=dev)
        global_visible_mask[self.cls_id[visible_mask].squeeze(-1)] = True
        return {"render": rendered_color[0], "invdepth": invdepth[0], "visibility_filter": visible_mask,
                "global_visibility_filter": global_visible_mask, "scale": scaling}

    def render_from_id(self, keyframe_id, pyr_lvl=0, bg=None):
        kf = self.keyframes[keyframe_id]
        view_matrix = kf.get_Rt().to(self.device)
        scale = 2 ** pyr_lvl
        width, height = self.width // scale, self.height // scale
        pkg = self.render(width, height, view_matrix, bg)
        pkg["render"] = (kf.exposure[:3, :3] @ pkg["render"].view(3, -1)) + kf.exposure[:3, 3, None]
        pkg["render"] = pkg["render"].clamp(0, 1).view(3, height, width)
        return pkg

    # -- weed_out_gaussians: h3dgsv3.py:942-953 (+ make_dummy_ext_tensor :750-763) ------------------------------------
    visible_threshold = 0.0  # run.sh --visible_threshold 0

    def make_dummy_ext_tensor(self):
        P = self.gaussian_params
        keys = ("id", "cls_id", "d_max", "xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "local_feat", "global_feat")
        return {k: P[k]["val"][:0].detach() for k in keys if k in P}

    @torch.no_grad()
    def weed_out_gaussians(self):
        visible_count = torch.zeros(self.xyz.shape[0], dtype=torch.int, device=self.device)
        for keyframe in self.keyframes:
            view_matrix = keyframe.get_Rt().transpose(0, 1).to(self.device)
            cam_centre = view_matrix.detach().inverse()[3, :3].to(self.device)
            ob_dist = (self.xyz - cam_centre).norm(dim=1, keepdim=True)
            selection_mask = (ob_dist < 2 * self.d_max).squeeze(-1)
            visible_count += selection_mask.int()
        visible_count = visible_count / len(self.keyframes)
        weed_mask = visible_count > self.visible_threshold
        self.optimizer.add_and_prune(self.make_dummy_ext_tensor(), weed_mask)

    # -- densification (harness/densify_mirror.py): update_voxel h3dgsv3.py:227-316, add_new_gaussians :766-940 ------------------
    def update_voxel(self, new_xyz, xyz, cls_id, voxel_size=0.1):
        from harness import densify_mirror
        return densify_mirror.voxel_labels(new_xyz, xyz, cls_id, voxel_size, scatter_max)

    @torch.no_grad()
    def add_new_gaussians(self, keyframe_id=-1):
        from harness import densify_mirror
        return densify_mirror.densify_from_keyframe(self, keyframe_id)

    # -- rigid_transform_gs: h3dgsv3.py:956-966 ---------------------------------------------------------------------------------
    @torch.no_grad()
    def rigid_transform_gs(self, old_c2ws, new_c2ws, cam_centres):
        old_c2ws = old_c2ws[self.id.squeeze(-1)]
        new_c2ws = new_c2ws[self.id.squeeze(-1)]
        new_xyz, new_rotation = update_gaussians(old_c2ws, new_c2ws, self.xyz, self.rotation)
        self.gaussian_params["xyz"]["val"] = new_xyz
        self.gaussian_params["rotation"]["val"] = new_rotation
        self.cam_centres = cam_centres

    def _rdk_for(self, h, w):
        key = (h, w)
        if key not in self._rdk:
            self._rdk[key] = radial_decay_kernel(h, w, self.rad_decay).to(self.device)
        return self._rdk[key]

    # -- optimisation step: h3dgsv3.py:401-469 ------------------------------------------------------------
    def optimization_step(self, keyframe_id=-1, is_important=True):
        if len(self.xyz) == 0:
            return None
        kf = self.keyframes[keyframe_id]
        lvl = kf.pyr_lvl
        kf.zero_grad()
        self.optimizer.zero_grad()
        pkg = self.render_from_id(keyframe_id, pyr_lvl=lvl, bg=torch.rand(3, device=self.device))
        image, invdepth, scale = pkg["render"], pkg["invdepth"], pkg["scale"]
        gt_image = kf.image_pyr[lvl]
        mono_idepth = kf.get_mono_idepth(lvl)
        c, h, w = image.shape
        rdk = self._rdk_for(h, w)
        if not is_important:
            error_map = rdk * (image - gt_image).abs()
            alpha_mask = ~((error_map[0] > 0.2) | (error_map[1] > 0.2) | (error_map[1] > 0.2))
            image, gt_image = image * alpha_mask, gt_image * alpha_mask
            invdepth, mono_idepth = invdepth * alpha_mask, mono_idepth * alpha_mask
        l1_loss = (rdk * (image - gt_image).abs()).mean()
        ssim_loss = 1 - fused_ssim(image[None], gt_image[None])
        depth_loss = (rdk * (invdepth - mono_idepth).abs()).mean()
        scaling_reg = scale.prod(dim=1).mean()
        loss = (self.lambda_dssim * ssim_loss + (1 - self.lambda_dssim) * l1_loss
                + kf.depth_loss_weight * depth_loss + self.scaling_reg_factor * scaling_reg)
        loss.backward()
        with torch.no_grad():
            kf.step()
            if not kf.is_test:
                self.optimizer.step(pkg["visibility_filter"], pkg["visibility_filter"].shape[0],
                                    pkg["global_visibility_filter"], pkg["global_visibility_filter"].shape[0])
            kf.latest_invdepth = pkg["invdepth"].detach()
        return loss.detach()


# ------------------------------------------------------------------------------- synthetic workload
def synthetic_cloud(N, width, height, seed=0, sh_k=16, z_range=(2.0, 6.0), sigma_px=2.0):
    """Seeded frustum-uniform Gaussian cloud of SURVEY.md 8(d) (camera = identity, fx = fy = 0.8 W)."""
    g = torch.Generator().manual_seed(seed)
    fx = 0.8 * width
    z = torch.rand(N, generator=g) * (z_range[1] - z_range[0]) + z_range[0]
    u = torch.rand(N, generator=g) * 2.1 - 1.05
    v = torch.rand(N, generator=g) * 2.1 - 1.05
    means = torch.stack([u * z * width / (2 * fx), v * z * height / (2 * fx), z], -1)
    s0 = sigma_px * 4.0 / fx
    scales = torch.exp(math.log(s0) + 0.3 * torch.randn(N, 3, generator=g))
    quats = torch.randn(N, 4, generator=g)
    quats = quats / quats.norm(dim=1, keepdim=True)
    opacities = torch.sigmoid(torch.randn(N, generator=g))
    sh = 0.3 * torch.randn(N, sh_k, 3, generator=g)
    return dict(means=means, quats=quats, scales=scales, opacities=opacities, sh=sh, fx=fx)


def build_synthetic_mapper(N, width, height, device, seed=0, n_keyframes=4, targets="random", lod=False, order="random", sigma_px=2.0):
    """MapperScene + keyframes on the synthetic cloud; the training resolution IS (width, height).
    targets: "random" = uniform-noise keyframe images (every parameter gets a large gradient: what the parity tests
    want); "render" = each keyframe observes the cloud itself (its own render and inverse depth), i.e. a converged map,
    so the cloud keeps the SURVEY 8(d) statistics while it is optimised.  With noise targets the optimiser dissolves
    the cloud within ~30 steps (opacities and radii shrink, intersections 3.7 M -> 2.3 M at 1 M Gaussians / 1080p,
    raster backward 1.0 -> 0.5 ms): a benchmark on them times a workload that gets lighter every step.
    lod: give every Gaussian the d_max the reference would have given it, creation depth x LoD level (h3dgsv3.py:891 with
    self.lods = [1, 2, 4, 8], :222): levels drawn 55/25/13/7 %, creation depth = distance to the camera x U(0.35, 1.5) (the map
    was built from keyframes nearer and farther than the one rendering it).  About 8 % of the cloud then fails
    `dist < 2 d_max` and is culled, another ~25 % is faded by the alpha ratio (h3dgsv3.py:628-639); with lod=False d_max is
    1e3 and the LoD logic never triggers (the SURVEY 8(d) statistics, used for the headline)."""
    c = synthetic_cloud(N, width, height, seed, sigma_px=sigma_px)   # sigma_px = 2: SURVEY 8(d)'s cloud; larger: a denser frame (longer tile lists)
    if order == "raster":
        # the order add_new_gaussians appends in: image raster order of the creating view (boolean-mask indexing of the pixel grid,
        # h3dgsv3.py:800).  SURVEY 8(d)'s cloud is in RANDOM order, the worst case for every kernel that scatters by screen position.
        m, fx_ = c["means"], c["fx"]
        px = (fx_ * m[:, 0] / m[:, 2] + width / 2).clamp(0, width - 1).long()
        py = (fx_ * m[:, 1] / m[:, 2] + height / 2).clamp(0, height - 1).long()
        perm = torch.argsort(py * width + px)
        c = {k: (v[perm] if torch.is_tensor(v) else v) for k, v in c.items()}
    elif order != "random":
        raise ValueError("order must be 'random' or 'raster'")
    torch.manual_seed(seed)  # nn.Linear's default init draws from the global generator
    scene = MapperScene(width, height, c["fx"], device)
    # Features start at zero (h3dgsv3.py:873-877), so mlp_cov initially outputs its last bias.
    # Pin that bias so the effective scale = exp(scaling) * sigmoid(0) and rotation * 1: the raw
    # log-scales are stored 2x larger and the rendered cloud has exactly the 8(d) statistics.
    with torch.no_grad():
        last = scene.mlp_cov[2]
        last.weight.mul_(0.01)
        last.bias.copy_(torch.tensor([0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 1.0]))
    op = c["opacities"].clamp(1e-4, 1 - 1e-4)
    scene.set_gaussians(c["means"], c["quats"], torch.log(2.0 * c["scales"]), torch.log(op / (1 - op)), c["sh"], seed=seed)
    if lod:
        gl = torch.Generator().manual_seed(seed + 7)
        level = torch.tensor([1.0, 2.0, 4.0, 8.0])[torch.multinomial(torch.tensor([0.55, 0.25, 0.13, 0.07]), N, True, generator=gl)]
        z_create = c["means"].norm(dim=1) * (0.35 + 1.15 * torch.rand(N, generator=gl))
        scene.gaussian_params["d_max"]["val"] = (z_create * level).reshape(N, 1).to(device)
    g = torch.Generator().manual_seed(seed + 1)
    for i in range(n_keyframes):
        Rt = torch.eye(4)
        Rt[:3, 3] = 0.02 * torch.randn(3, generator=g)
        img = torch.rand(3, height, width, generator=g).to(device)
        idepth = (0.15 + 0.35 * torch.rand(1, height, width, generator=g)).to(device)
        scene.add_keyframe(Keyframe(img, idepth, Rt.to(device), torch.device(device)))
    if targets == "render":
        with torch.no_grad():
            for i, kf in enumerate(scene.keyframes):
                pkg = scene.render_from_id(i, bg=torch.full((3,), 0.5, device=scene.device))
                kf.image_pyr[0] = pkg["render"].clamp(0, 1).contiguous()
                kf.idepth_pyr[0] = pkg["invdepth"].contiguous()
    elif targets != "random":
        raise ValueError("targets must be 'random' or 'render'")
    return scene
