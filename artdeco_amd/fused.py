"""Fused replacement for the torch glue of SceneModel.render.

`patch_scene_model(scene_model)` swaps the instance's `render` for `fused_render`, which reads the
same parameter dictionary (Reconstruct/scene/scene_models/h3dgsv3.py:124-171), the same `mlp_cov`,
`d_max`, `cls_id`, `tanfovx/y`, `active_sh_degree`, `args.low_pass_filter_eps`, and returns the same
package (`render`, `invdepth`, `visibility_filter`, `global_visibility_filter`, `scale`) --
h3dgsv3.py:617-700 -- so optimization_step / render_from_id / the optimiser run unchanged.

What changes underneath: the LoD cull, activations and mlp_cov run in ONE HIP kernel per direction
(artdeco_amd/csrc/lod_params.hip) over all Gaussians with no boolean-mask compaction (unselected
Gaussians get opacity 0 and are culled by the projection), f_dc / f_rest are consumed in place
(no concatenation), and the masks are produced without a host sync.  One observable difference:
`scale` is returned for all N Gaussians with rows of unselected ones set to 1 (the reference
returns only the selected rows); ARTDECO only uses it for `scale.prod(dim=1).mean()` with
scaling_reg_factor = 0 (dataloaders/args.py:90).
"""
from __future__ import annotations

import types

import torch

from . import _lib
from .rasterizer import _WS, _stage, render_camera

_NW = 32 * 32 + 32 + 7 * 32 + 7


class FusedLodParams(torch.autograd.Function):
    """(xyz, opacity_raw[N,1], scaling_raw[N,3], rotation[N,4], local_feat[N,16], global_feat[V,16],
    W1, b1, W2, b2 | cls_id, d_max, viewmat) -> opac_eff [N], scale_eff [N,3], quat_eff [N,4], selected [N] bool"""

    @staticmethod
    def forward(ctx, xyz, opacity_raw, scaling_raw, rotation, local_feat, global_feat, W1, b1, W2, b2, cls_id, d_max, viewmat):
        lib = _lib.load()
        _lib.require_cuda(xyz, opacity_raw, scaling_raw, rotation, local_feat, global_feat, W1, cls_id, d_max, viewmat)
        dev, N = xyz.device, xyz.shape[0]
        c = lambda t: t.detach().contiguous()
        xyz, opacity_raw, scaling_raw, rotation = c(xyz), c(opacity_raw), c(scaling_raw), c(rotation)
        local_feat, global_feat, W1, b1, W2, b2 = c(local_feat), c(global_feat), c(W1), c(b1), c(W2), c(b2)
        cls_id, d_max, viewmat = c(cls_id), c(d_max), c(viewmat).float()
        if cls_id.dtype != torch.int64:
            raise TypeError("cls_id must be int64 (h3dgsv3.py:128)")
        L, G, Hd = local_feat.shape[1], global_feat.shape[1], W1.shape[0]
        if W1.shape != (Hd, G + L) or W2.shape != (7, Hd):
            raise ValueError("mlp_cov must be Linear(G+L, G+L) -> ReLU -> Linear(G+L, 7)")
        with torch.cuda.device(dev):
            opac = torch.empty(N, dtype=torch.float32, device=dev)
            scale = torch.empty(N, 3, dtype=torch.float32, device=dev)
            quat = torch.empty(N, 4, dtype=torch.float32, device=dev)
            sel = torch.empty(N, dtype=torch.bool, device=dev)
            with _stage("lod_params_fwd"):
                rc = lib.adk_lod_params_fwd(N, xyz.data_ptr(), opacity_raw.data_ptr(), scaling_raw.data_ptr(),
                                            rotation.data_ptr(), local_feat.data_ptr(), global_feat.data_ptr(),
                                            cls_id.data_ptr(), d_max.data_ptr(), L, G, Hd, W1.data_ptr(), b1.data_ptr(),
                                            W2.data_ptr(), b2.data_ptr(), viewmat.data_ptr(), opac.data_ptr(),
                                            scale.data_ptr(), quat.data_ptr(), sel.data_ptr(), _lib.stream_of(xyz))
        _lib.check(rc, "adk_lod_params_fwd")
        ctx.dims = (L, G, Hd)
        ctx.save_for_backward(xyz, opacity_raw, scaling_raw, rotation, local_feat, global_feat, W1, b1, W2, b2, cls_id, d_max, viewmat)
        ctx.mark_non_differentiable(sel)
        return opac, scale, quat, sel

    @staticmethod
    def backward(ctx, v_opac, v_scale, v_quat, _v_sel):
        lib = _lib.load()
        (xyz, opacity_raw, scaling_raw, rotation, local_feat, global_feat, W1, b1, W2, b2, cls_id, d_max,
         viewmat) = ctx.saved_tensors
        L, G, Hd = ctx.dims
        dev, N = xyz.device, xyz.shape[0]
        z = lambda t, ref: (torch.zeros_like(ref) if t is None else t.contiguous())
        with torch.cuda.device(dev):
            v_opac = z(v_opac, opacity_raw.view(-1))
            v_scale, v_quat = z(v_scale, scaling_raw), z(v_quat, rotation)
            v_xyz = torch.zeros_like(xyz)
            v_o, v_s, v_r = torch.empty_like(opacity_raw), torch.empty_like(scaling_raw), torch.empty_like(rotation)
            v_lf = torch.empty_like(local_feat)
            v_gf = torch.zeros_like(global_feat)
            v_mlp = torch.empty(_NW, dtype=torch.float32, device=dev)
            ws = _WS.get(dev, int(lib.adk_lod_params_bwd_workspace_bytes(N)))
            with _stage("lod_params_bwd"):
                rc = lib.adk_lod_params_bwd(N, xyz.data_ptr(), opacity_raw.data_ptr(), scaling_raw.data_ptr(),
                                            rotation.data_ptr(), local_feat.data_ptr(), global_feat.data_ptr(),
                                            cls_id.data_ptr(), d_max.data_ptr(), L, G, Hd, W1.data_ptr(), b1.data_ptr(),
                                            W2.data_ptr(), b2.data_ptr(), viewmat.data_ptr(), v_opac.data_ptr(),
                                            v_scale.data_ptr(), v_quat.data_ptr(), v_xyz.data_ptr(), v_o.data_ptr(),
                                            v_s.data_ptr(), v_r.data_ptr(), v_lf.data_ptr(), v_gf.data_ptr(),
                                            v_mlp.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_of(xyz))
        _lib.check(rc, "adk_lod_params_bwd")
        vW1 = v_mlp[:Hd * (G + L)].view(Hd, G + L)
        vb1 = v_mlp[Hd * (G + L):Hd * (G + L) + Hd]
        o2 = Hd * (G + L) + Hd
        vW2 = v_mlp[o2:o2 + 7 * Hd].view(7, Hd)
        vb2 = v_mlp[o2 + 7 * Hd:o2 + 7 * Hd + 7]
        return v_xyz, v_o, v_s, v_r, v_lf, v_gf, vW1, vb1, vW2, vb2, None, None, None


def supported(scene) -> bool:
    """The fused kernel is specialised for run.sh's dimensions (16 + 16 features, hidden 32)."""
    try:
        lin1, act, lin2 = scene.mlp_cov[0], scene.mlp_cov[1], scene.mlp_cov[2]
        return (scene.local_feat.shape[1] == 16 and scene.global_feat.shape[1] == 16 and lin1.weight.shape == (32, 32)
                and lin2.weight.shape == (7, 32) and isinstance(act, torch.nn.ReLU))
    except Exception:
        return False


def fused_render(self, width: int, height: int, view_matrix: torch.Tensor, bg: torch.Tensor | None = None):
    """Drop-in body for SceneModel.render (h3dgsv3.py:617-700)."""
    dev = self.device
    bg = torch.zeros(3, device=dev) if bg is None else bg.to(dev)
    lock = getattr(self, "lock", None)
    if lock is not None:
        lock.acquire()
    try:
        lin1, lin2 = self.mlp_cov[0], self.mlp_cov[2]
        P = self.gaussian_params
        opac, scaling, quat, sel = FusedLodParams.apply(
            P["xyz"]["val"], P["opacity"]["val"], P["scaling"]["val"], P["rotation"]["val"], P["local_feat"]["val"],
            P["global_feat"]["val"], lin1.weight, lin1.bias, lin2.weight, lin2.bias, P["cls_id"]["val"], P["d_max"]["val"],
            view_matrix.detach())
        fl_x, fl_y = width / (2 * self.tanfovx), height / (2 * self.tanfovy)
        K = torch.tensor([[fl_x, 0, width / 2.0], [0, fl_y, height / 2.0], [0, 0, 1]], dtype=torch.float32, device=dev)
        eps2d = self.args.low_pass_filter_eps if hasattr(self, "args") else self.eps2d
        out = render_camera(P["xyz"]["val"], quat, scaling, opac, P["f_dc"]["val"], view_matrix.float(), K, width, height,
                            sh_degree=self.active_sh_degree, eps2d=eps2d, sh_rest=P["f_rest"]["val"])
        col4, alphas, radii = out[0], out[1], out[2]
        rendered_alpha = alphas.permute(2, 0, 1)
        rendered_color = col4[..., 0:3].permute(2, 0, 1) + (1.0 - rendered_alpha) * bg[:, None, None]
        invdepth = 1.0 / col4[..., 3:4].permute(2, 0, 1)
        visible_mask = (radii[:, 0] > 0) & (radii[:, 1] > 0)
        cls = P["cls_id"]["val"].view(-1)
        n_vox = P["global_feat"]["val"].shape[0]
        global_visible_mask = torch.zeros(n_vox, dtype=torch.int32, device=dev).index_add_(0, cls, visible_mask.int()) > 0
    finally:
        if lock is not None:
            lock.release()
    return {"render": rendered_color, "invdepth": invdepth, "visibility_filter": visible_mask,
            "global_visibility_filter": global_visible_mask, "scale": scaling}


class ExposureClamp(torch.autograd.Function):
    """clamp(E[:3,:3] @ img.view(3,-1) + E[:3,3,None], 0, 1) without the [3,P]x[P,3] GEMMs (h3dgsv3.py:611-614)."""

    @staticmethod
    def forward(ctx, E, img):
        lib = _lib.load()
        _lib.require_cuda(E, img)
        Ec, ic = E.detach().contiguous().float(), img.detach().contiguous().float()
        P = ic.numel() // 3
        with torch.cuda.device(ic.device):
            out = torch.empty_like(ic)
            with _stage("exposure_fwd"):
                rc = lib.adk_exposure_fwd(Ec.data_ptr(), ic.data_ptr(), P, out.data_ptr(), _lib.stream_of(ic))
        _lib.check(rc, "adk_exposure_fwd")
        ctx.save_for_backward(Ec, ic)
        return out

    @staticmethod
    def backward(ctx, v_out):
        lib = _lib.load()
        Ec, ic = ctx.saved_tensors
        P = ic.numel() // 3
        with torch.cuda.device(ic.device):
            v_out = v_out.contiguous()
            v_img = torch.empty_like(ic)
            v_E = torch.zeros(3, 4, dtype=torch.float32, device=ic.device)
            with _stage("exposure_bwd"):
                rc = lib.adk_exposure_bwd(Ec.data_ptr(), ic.data_ptr(), v_out.data_ptr(), P, v_img.data_ptr(), v_E.data_ptr(),
                                          _lib.stream_of(ic))
        _lib.check(rc, "adk_exposure_bwd")
        return v_E, v_img


def fused_render_from_id(self, keyframe_id, pyr_lvl=0, bg=None):
    """Drop-in body for SceneModel.render_from_id (h3dgsv3.py:595-615)."""
    bg = torch.zeros(3, device=self.device) if bg is None else bg.to(self.device)
    keyframe = self.keyframes[keyframe_id]
    view_matrix = keyframe.get_Rt().to(self.device)
    scale = 2 ** pyr_lvl
    width, height = self.width // scale, self.height // scale
    pkg = self.render(width, height, view_matrix, bg)
    pkg["render"] = ExposureClamp.apply(keyframe.exposure, pkg["render"]).view(3, height, width)
    return pkg


def fused_optimizer_step(self, visibility, N, global_visibility, N_global):
    """Drop-in body for SparseGaussianAdam.step (Reconstruct/scene/optimizers.py:77-161): the same updates, the
    same learning-rate decay, one kernel launch (adk_adam_update_multi) and no boolean-index host sync."""
    import ctypes
    lib = _lib.load()
    skip = ("id", "cls_id", "d_max")
    b1, b2 = self.betas
    ent = []  # (param, grad, m, v, vis, lr_tensor|None, lr_val, decay, lr_min, rows, M)
    keep = []
    for key, pd in self.params.items():
        if key in skip:
            continue
        val = pd["val"]
        if val.grad is None:
            continue
        grad = val.grad.contiguous()
        keep.append(grad)
        if key.startswith("mlp"):
            ent.append((val, grad, pd["exp_avg"], pd["exp_avg_sq"], None, None, float(pd["lr"]), 1.0, 0.0, val.numel(), 1))
            if key in self.lr_dict:  # host-side float schedule, optimizers.py:102-104
                pd["lr"] = max(pd["lr"] * self.lr_dict[key]["lr_decay"], self.lr_dict[key]["lr_init"] * 0.1)
            continue
        vis, n = (global_visibility, N_global) if key == "global_feat" else (visibility, N)
        lr = pd["lr"]
        decay, lr_min = 1.0, 0.0
        if key in self.lr_dict:
            decay, lr_min = float(self.lr_dict[key]["lr_decay"]), float(self.lr_dict[key]["lr_init"] * 0.1)
            if lr.numel() != val.numel():  # per-row lr cannot be decayed per element in place; keep reference path
                decay = 1.0
        ent.append((val, grad, pd["exp_avg"], pd["exp_avg_sq"], vis, lr, 0.0, decay, lr_min, int(n), val.numel() // int(n)))
    if not ent:
        return
    n = len(ent)
    if n > 16:
        raise _lib.AdkError("fused optimizer step supports at most 16 tensors")
    dev = ent[0][0].device
    VP, I64, F32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_float * n
    ptr = lambda t: None if t is None else t.data_ptr()
    for e in ent:
        for t in (e[0], e[2], e[3]):
            if not t.is_contiguous() or t.dtype != torch.float32:
                raise _lib.AdkError("fused optimizer step needs contiguous float32 parameters and moments")
    args = (n, VP(*[e[0].data_ptr() for e in ent]), VP(*[e[1].data_ptr() for e in ent]), VP(*[e[2].data_ptr() for e in ent]),
            VP(*[e[3].data_ptr() for e in ent]), VP(*[ptr(e[4]) for e in ent]), VP(*[ptr(e[5]) for e in ent]),
            I64(*[(e[5].numel() if e[5] is not None else 0) for e in ent]), F32(*[e[6] for e in ent]),
            F32(*[e[7] for e in ent]), F32(*[e[8] for e in ent]), I64(*[e[9] for e in ent]), I64(*[e[10] for e in ent]),
            float(b1), float(b2), float(self.eps))
    with torch.no_grad(), torch.cuda.device(dev):
        with _stage("adam_multi"):
            rc = lib.adk_adam_update_multi(*args, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "adk_adam_update_multi")
    # per-row lr tensors (finetune path, h3dgsv3.py:1240-1247) keep the reference's torch decay
    for key, pd in self.params.items():
        if key in skip or key.startswith("mlp") or key not in self.lr_dict or pd["val"].grad is None:
            continue
        if pd["lr"].numel() != pd["val"].numel():
            vis = global_visibility if key == "global_feat" else visibility
            pd["lr"][vis] *= self.lr_dict[key]["lr_decay"]
            pd["lr"].clamp_min_(self.lr_dict[key]["lr_init"] * 0.1)


def patch_scene_model(scene) -> bool:
    """Install fused_render on this scene-model instance (ARTDECO's SceneModel or artdeco_amd.mapper.MapperScene).
    Returns False (and leaves the object untouched) when the mlp/feature shapes are not the supported ones."""
    if not supported(scene):
        return False
    scene._unfused_render = scene.render
    scene.render = types.MethodType(fused_render, scene)
    if hasattr(scene, "render_from_id"):
        scene._unfused_render_from_id = scene.render_from_id
        scene.render_from_id = types.MethodType(fused_render_from_id, scene)
    opt = getattr(scene, "optimizer", None)
    if opt is not None and hasattr(opt, "lr_dict") and hasattr(opt, "params"):
        opt._unfused_step = opt.step
        opt.step = types.MethodType(fused_optimizer_step, opt)
    return True
