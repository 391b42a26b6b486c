"""The optimisation step as ONE native call (`adk_mapper_step`, include/artdeco_hip.h) behind `fused.fused_train_on_keyframe`.

What `fused._train_on_keyframe_by_hand` does with ~20 C calls, ~40 `torch.empty` and ~200 `data_ptr()` per step -- forward, loss and
backward of SceneModel.optimization_step (Reconstruct/scene/scene_models/h3dgsv3.py:418-455) -- is one call here: every intermediate
and every gradient lives in a PLAN (a set of buffers allocated once per (N, V, width, height) and reused by every step), the
argument block is a ctypes struct whose layout is read from the header's field list, and the step's single host wait sits inside the
C function between the scatter's launch and the sort's.  Same kernels, same launch order, same arithmetic: the results are
bit-identical to the per-stage chain (tests/test_native_step.py).  The optimiser steps (Keyframe.step, SparseGaussianAdam.step)
stay with `fused._apply_steps`: they read the gradients from `.grad`, which this module points at the plan's buffers.

Falls back to the per-stage chain (returns None, nothing modified) when a tensor is not in run.sh's layout, when a leaf already
carries a gradient (the chain accumulates), or when the frame needs the global binning route (a tile list above 4 194 304 entries: lists
above 8 192 take the long-list sort inside the call since round 5).
`ARTDECO_AMD_NATIVE_STEP=0` disables it.
"""
from __future__ import annotations

import ctypes
import os
import re

import torch

from . import _lib
from . import small_inverse as _small_inverse
from . import rasterizer

_GRAIN = 1 << 20
_KIND = {"P": ctypes.c_void_p, "L": ctypes.c_int64, "I": ctypes.c_int32, "F": ctypes.c_float}
ADK_STEP_ECAPACITY, ADK_STEP_EROUTE = -16, -17


def _parse_header():
    """(fields [(kind, name)], stage names) from the header's ADK_MAPPER_STEP_FIELDS / ADK_MAPPER_STAGES lists."""
    src = open(_lib.HEADER).read()
    m = re.search(r"#define ADK_MAPPER_STEP_FIELDS\(P, L, I, F\)(.*?)\n#define ADK_MAPPER_STAGES\(S\)(.*?)\n#define ADK_MAPPER_N_STAGES", src, re.S)
    if m is None:
        raise _lib.AdkError("include/artdeco_hip.h: ADK_MAPPER_STEP_FIELDS / ADK_MAPPER_STAGES not found")
    fields = re.findall(r"\b([PLIF])\((\w+)\)", m.group(1))
    stages = re.findall(r"\bS\((\w+)\)", m.group(2))
    return fields, stages


class _Lazy:
    """The header is parsed on first use, not at import: `import artdeco_amd.fused` must work (and ARTDECO_AMD_NATIVE_STEP=0 must be
    honoured) on an installation whose header is missing or older; `enabled()` then simply says no."""
    fields = stages = args_type = None
    error = None


def _ensure_parsed() -> bool:
    if _Lazy.fields is not None:
        return True
    if _Lazy.error is not None:
        return False
    try:
        fields, stages = _parse_header()
    except (OSError, _lib.AdkError) as e:
        _Lazy.error = e
        return False
    _Lazy.fields, _Lazy.stages = fields, stages
    _Lazy.args_type = type("StepArgs", (ctypes.Structure,), {"_fields_": [(name, _KIND[kind]) for kind, name in fields]})
    return True


def __getattr__(name):      # module attributes that need the header: _FIELDS, STAGES, StepArgs
    if name in ("_FIELDS", "STAGES", "StepArgs"):
        if not _ensure_parsed():
            raise _Lazy.error
        return {"_FIELDS": _Lazy.fields, "STAGES": _Lazy.stages, "StepArgs": _Lazy.args_type}[name]
    raise AttributeError(name)


class StepOut(ctypes.Structure):
    _fields_ = [("n_isects", ctypes.c_int64), ("max_tile", ctypes.c_int64), ("stage", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("wait_ns", ctypes.c_int64)]


_checked = False


def _check_layout(lib) -> None:
    global _checked
    if _checked:
        return
    if int(lib.adk_mapper_step_args_bytes()) != ctypes.sizeof(_Lazy.args_type):
        raise _lib.AdkError(f"AdkMapperStepArgs: the library's struct has {int(lib.adk_mapper_step_args_bytes())} bytes, the binding's "
                            f"{ctypes.sizeof(_Lazy.args_type)} (header and library out of step: rebuild)")
    _checked = True


def enabled() -> bool:
    """On unless ARTDECO_AMD_NATIVE_STEP=0, or ARTDECO_AMD_LOD_ADAM=1 asks for the (slower, opt-in) Gaussian Adam inside the LoD backward,
    which only the per-stage chain offers."""
    return (os.environ.get("ARTDECO_AMD_NATIVE_STEP", "1") != "0" and os.environ.get("ARTDECO_AMD_LOD_ADAM", "0") != "1"
            and _ensure_parsed())


def _round_cap(n: int) -> int:
    return max(((int(n) + _GRAIN - 1) // _GRAIN) * _GRAIN, _GRAIN)


_NW = 32 * 32 + 32 + 7 * 32 + 7
STATS = {"native": 0, "fallback_layout": 0, "fallback_route": 0, "capacity_retries": 0, "plans_built": 0, "long_list_steps": 0, "wait_ns": 0,
         "plan_bytes": 0}    # plan_bytes: device memory held by the step plans of the most recently stepped scene (shared buffers counted once)


_N_GRAIN = int(os.environ.get("ARTDECO_AMD_PLAN_GRAIN", 1 << 16))   # 1 = exact sizes (a plan per N: the lab's A/B)
_V_GRAIN = max(_N_GRAIN >> 4, 1)


def _round_up(n: int, grain: int) -> int:
    return max(((int(n) + grain - 1) // grain) * grain, grain)


class StepPlan:
    """Buffers of one (width, height) for UP TO N Gaussians and V voxels, and the argument block that points at them.  The per-Gaussian /
    per-voxel buffers are sized for the capacities (N, V rounded up to 65 536 / 4 096): a densification that adds 7 000 Gaussians to a
    million keeps the plan -- no allocation of new sizes (each a hipMalloc, a stall of the whole stream), only new views of the same buffers
    for the leaves' `.grad` (`set_sizes`)."""

    #: buffers whose size depends on the map only (N Gaussians, V voxels): ONE set per scene, shared by the plans of all its resolutions
    #: (pyramid levels, densification renders): ~330 B per Gaussian + the LoD workspace that every extra plan used to duplicate
    PER_MAP = frozenset(("opac", "scale", "quat", "sel", "rec", "radii", "depth_keys", "gauss_ids", "tiles_per_gauss", "vis", "gvis", "v_rec",
                         "v_means", "v_quats", "v_scales", "v_opac", "v_opacity_raw", "v_scaling_raw", "v_rotation", "v_local_feat",
                         "v_global_feat", "lod_ws", "v_dc", "v_rest"))

    def __init__(self, lib, dev, N, V, W, H, tile_px, capacity, shared=None):
        self.n, self.v = -1, -1
        N, V = _round_up(N, _N_GRAIN), _round_up(V, _V_GRAIN)
        if shared is None or shared.get("key") != (dev, N, V):
            shared = {"key": (dev, N, V), "t": {}}
        self.shared = shared
        self.dev, self.N, self.V, self.W, self.H, self.tile_px = dev, N, V, W, H, tile_px
        self.args = _Lazy.args_type()
        self.out = StepOut()
        self.t: dict[str, torch.Tensor] = {}
        self.bound: tuple = ()          # the tensors whose pointers the block currently holds (kept alive, compared with `is`) ...
        self.bound_ptrs: tuple = ()     # ... and their data_ptr()s: a storage swap that keeps the tensor's identity (`.data = ...`, `set_`, `module.to()`)
        self.route_miss = 0
        self.skip_until = 0
        self.calls = 0
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        tpw, tph = tile_px
        tile_w, tile_h = (W + tpw - 1) // tpw, (H + tph - 1) // tph
        n_sums = int(lib.adk_fused_ssim_fwd_sums_count(1, 3, H, W))
        def e(name, *shape, **kw):
            if name in self.PER_MAP:
                if name not in shared["t"]:
                    shared["t"][name] = torch.empty(*shape, **kw)
                self.t[name] = shared["t"][name]
            else:
                self.t[name] = torch.empty(*shape, **kw)
        e("viewmat", 4, 4, **f32); e("opac", N, **f32); e("scale", N, 3, **f32); e("quat", N, 4, **f32)
        e("sel", N, dtype=torch.bool, device=dev); e("rec", N, 12, **f32); e("radii", N, 2, **i32)
        e("depth_keys", N, **i32); e("gauss_ids", N, **i32); e("tiles_per_gauss", N, **i32)
        e("offsets", tile_h, tile_w, **i32); e("bin_stats", 2, dtype=torch.int64, device=dev)
        e("bin_table", int(lib.adk_bin_local_workspace_bytes_t(W, H, tpw, tph)) + 256, **u8)
        e("render_colors", H, W, 4, **f32); e("render_alphas", H, W, 1, **f32); e("final_T", H, W, **f32); e("last_ids", H, W, **i32)
        e("vis", N, dtype=torch.bool, device=dev); e("gvis", V, dtype=torch.bool, device=dev)
        e("image", 3, H, W, **f32); e("gt_used", 3, H, W, **f32); e("dm", 3, 3, H, W, **f32); e("parts", 4, **f32)
        e("ssim_sums", n_sums, **f32); e("photo_ws", int(lib.adk_photometric_workspace_bytes(W, H)), **u8)
        e("v_img", 3, H, W, **f32); e("v_col", H, W, 4, **f32); e("v_alpha", H, W, 1, **f32); e("v_exposure", 12, **f32)
        e("v_rec", N, 12, **f32); e("v_means", N, 3, **f32); e("v_quats", N, 4, **f32); e("v_scales", N, 3, **f32); e("v_opac", N, **f32)
        self.t["cam_grad"] = torch.zeros(32, **f32)     # 16 doubles (ABI v19); the projection backward leaves it zeroed again
        e("v_viewmat", 4, 4, **f32)
        e("v_opacity_raw", N, 1, **f32); e("v_scaling_raw", N, 3, **f32); e("v_rotation", N, 4, **f32); e("v_local_feat", N, 16, **f32)
        e("v_global_feat", V, 16, **f32); e("v_mlp", _NW, **f32); e("lod_ws", int(lib.adk_lod_params_bwd_workspace_bytes(N)), **u8)
        e("v_r6", 3, 2, **f32); e("v_t", 3, **f32)
        self.t["unit_grad"] = torch.ones(1, **f32)
        self.t["reg_ws"] = torch.zeros(4, dtype=torch.float64, device=dev)     # the scaling regulariser's sums (left zeroed by the call)
        A = self.args
        for name, ten in self.t.items():
            setattr(A, name, ten.data_ptr())
        A.N, A.V, A.width, A.height, A.tile_px_w, A.tile_px_h = 0, 0, W, H, tpw, tph
        A.bin_table_bytes = self.t["bin_table"].numel()
        A.lod_ws_bytes = self.t["lod_ws"].numel()
        A.photo_ws_bytes = self.t["photo_ws"].numel()
        A.n_ssim_sums = n_sums
        A.ssim_grad_scale = 0.0
        self.set_capacity(capacity)
        self.grads: dict[str, torch.Tensor] = {}
        self.vis = self.gvis = None
        STATS["plans_built"] += 1

    def set_sizes(self, n: int, v: int) -> None:
        """The step's actual sizes: the kernels' N / V and the leaves' `.grad` (views of the plan's buffers, shaped like the leaves)."""
        if n == self.n and v == self.v:
            return
        self.n, self.v = n, v
        self.args.N, self.args.V = n, v
        t = self.t
        m = t["v_mlp"]
        o2 = 32 * 32 + 32
        self.grads = {"xyz": t["v_means"][:n], "opacity": t["v_opacity_raw"][:n], "scaling": t["v_scaling_raw"][:n],
                      "rotation": t["v_rotation"][:n], "local_feat": t["v_local_feat"][:n], "global_feat": t["v_global_feat"][:v],
                      "W1": m[:32 * 32].view(32, 32), "b1": m[32 * 32:o2], "W2": m[o2:o2 + 7 * 32].view(7, 32), "b2": m[o2 + 7 * 32:o2 + 7 * 32 + 7],
                      "exposure": t["v_exposure"].view(3, 4), "r6": t["v_r6"], "t": t["v_t"]}
        self.vis, self.gvis = t["vis"][:n], t["gvis"][:v]
        if "v_dc" in t:
            self.grads["f_dc"], self.grads["f_rest"] = t["v_dc"][:n], t["v_rest"][:n]

    def set_capacity(self, capacity: int) -> None:
        cap = _round_cap(capacity)
        self.t["pairs"] = torch.empty(cap, dtype=torch.int64, device=self.dev)
        self.t["pairs2"] = torch.empty(cap, dtype=torch.int64, device=self.dev)   # tile lists above 8 192 entries ping-pong between the two
        self.t["flatten_ids"] = torch.empty(cap, dtype=torch.int32, device=self.dev)
        self.args.pairs = self.t["pairs"].data_ptr()
        self.args.pairs2 = self.t["pairs2"].data_ptr()
        self.args.flatten_ids = self.t["flatten_ids"].data_ptr()
        self.args.isect_capacity = cap

    def color_grads(self, f_dc, f_rest):
        """Gradient buffers of the SH colours: only a test keyframe's step needs them (no colour Adam inside the projection backward)."""
        if "v_dc" not in self.t or self.t["v_rest"].shape[1:] != f_rest.shape[1:]:
            st = self.shared["t"]
            if "v_dc" not in st or st["v_rest"].shape[1:] != f_rest.shape[1:]:
                st["v_dc"] = torch.empty((self.N,) + tuple(f_dc.shape[1:]), dtype=torch.float32, device=self.dev)
                st["v_rest"] = torch.empty((self.N,) + tuple(f_rest.shape[1:]), dtype=torch.float32, device=self.dev)
            self.t["v_dc"], self.t["v_rest"] = st["v_dc"], st["v_rest"]
            self.grads["f_dc"], self.grads["f_rest"] = self.t["v_dc"][:self.n], self.t["v_rest"][:self.n]
        self.args.v_dc, self.args.v_rest = self.t["v_dc"].data_ptr(), self.t["v_rest"].data_ptr()
        return self.grads["f_dc"], self.grads["f_rest"]


def _ok32(t, shape=None) -> bool:
    return (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and (shape is None or tuple(t.shape) == tuple(shape)))


def _plan_for(scene, lib, dev, N, V, W, H, tile_px):
    plans = scene.__dict__.setdefault("_adk_step_plans", {})
    key = (W, H, tile_px)
    plan = plans.get(key)
    if plan is None or plan.N < N or plan.V < V or plan.dev != dev:
        hint = rasterizer._CAPACITY_HINT.get((dev.index, W, H, tile_px[0]))
        cap = int(hint * 1.25) if hint else (plan.args.isect_capacity if plan is not None else 4 * N)
        skip = plan.skip_until - plan.calls if plan is not None else 0
        plans.pop(key, None)            # the superseded plan goes NOW (its image-sized buffers; the leaves' .grad may still hold views of the per-map set)
        plan = None
        # the per-map buffers of the scene's most recent plan: reused when the capacities agree (another resolution of the same map)
        shared = scene.__dict__.get("_adk_step_shared")
        plan = plans[key] = StepPlan(lib, dev, N, V, W, H, tile_px, cap, shared)
        scene.__dict__["_adk_step_shared"] = plan.shared
        plan.skip_until = max(skip, 0)
        seen, total = set(), 0
        for pl in plans.values():
            for ten in pl.t.values():
                if ten.data_ptr() not in seen:
                    seen.add(ten.data_ptr())
                    total += ten.numel() * ten.element_size()
        STATS["plan_bytes"] = total
    plan.set_sizes(N, V)
    return plan


def train_on_keyframe(scene, keyframe, is_important):
    """One optimisation step of `scene` on `keyframe` through adk_mapper_step + fused._apply_steps: (loss (0-dim tensor), None) -- or
    (None, bg) when the per-stage chain has to run this step instead: nothing has been modified then, and `bg` is the random background
    if it has already been drawn (the chain must use it rather than draw again: one draw per step, as in the reference)."""
    from . import fused
    lib = _lib.load()
    _check_layout(lib)
    dev = scene.device if isinstance(scene.device, torch.device) else torch.device(scene.device)
    P = scene.gaussian_params
    lin1, lin2 = scene.mlp_cov[0], scene.mlp_cov[2]
    lvl = keyframe.pyr_lvl
    scale = 2 ** lvl
    W, H = scene.width // scale, scene.height // scale
    xyz = P["xyz"]["val"]
    N, V = int(xyz.shape[0]), int(P["global_feat"]["val"].shape[0])
    tile_px = rasterizer.default_tile_px()
    if N == 0 or V == 0 or not lib.adk_bin_local_supported_t(W, H, tile_px[0], tile_px[1]):
        STATS["fallback_layout"] += 1
        return None, None
    r6, t, E = keyframe.rW2C, keyframe.tW2C, keyframe.exposure
    gt, mono = keyframe.image_pyr[lvl], keyframe.get_mono_idepth(lvl)
    f_dc, f_rest = P["f_dc"]["val"], P["f_rest"]["val"]
    leaves = (xyz, P["opacity"]["val"], P["scaling"]["val"], P["rotation"]["val"], P["local_feat"]["val"], P["global_feat"]["val"],
              lin1.weight, lin1.bias, lin2.weight, lin2.bias)
    cls_id, d_max = P["cls_id"]["val"], P["d_max"]["val"]
    sh_K = 1 + int(f_rest.shape[1]) if f_rest.dim() == 3 else 0
    deg = int(scene.active_sh_degree)
    # a test keyframe's optimiser holds no exposure entry (keyframe.py:119-123): its zero_grad leaves exposure.grad alone and autograd keeps
    # accumulating into it; only a parameter the keyframe steps must arrive with an empty .grad
    e_stepped = any(pd.get("val") is E for pd in getattr(getattr(keyframe, "optimizer", None), "params", {}).values())
    layout_ok = (_ok32(xyz, (N, 3)) and _ok32(leaves[1], (N, 1)) and _ok32(leaves[2], (N, 3)) and _ok32(leaves[3], (N, 4))
                 and _ok32(leaves[4], (N, 16)) and _ok32(leaves[5], (V, 16)) and _ok32(leaves[6], (32, 32)) and _ok32(leaves[7], (32,))
                 and _ok32(leaves[8], (7, 32)) and _ok32(leaves[9], (7,)) and _ok32(f_dc, (N, 1, 3)) and _ok32(f_rest, (N, sh_K - 1, 3))
                 and cls_id.is_cuda and cls_id.dtype == torch.int64 and cls_id.is_contiguous() and cls_id.numel() == N
                 and _ok32(d_max) and d_max.numel() == N and _ok32(r6, (3, 2)) and _ok32(t, (3,)) and _ok32(E, (3, 4))
                 and _ok32(gt, (3, H, W)) and _ok32(mono) and mono.numel() == H * W and 0 <= deg <= 3 and (deg + 1) ** 2 <= sh_K
                 # every Gaussian tensor but xyz / rotation trains in every step (those two lose requires_grad between rigid_transform_gs and
                 # the next add_and_prune, h3dgsv3.py:964-965: their `.grad` then stays None, as the reference's would)
                 and all(x.requires_grad for x in leaves[1:3] + leaves[4:]) and f_dc.requires_grad and f_rest.requires_grad
                 and all(x.grad is None for x in leaves) and f_dc.grad is None and f_rest.grad is None
                 and r6.grad is None and t.grad is None and (E.grad is None or not e_stepped))
    if not layout_ok:
        STATS["fallback_layout"] += 1
        return None, None
    plan = _plan_for(scene, lib, dev, N, V, W, H, tile_px)
    plan.calls += 1
    if plan.calls <= plan.skip_until:
        STATS["fallback_route"] += 1
        return None, None
    color_state = None if keyframe.is_test else fused._color_adam_state(scene.optimizer)
    if color_state is not None and not (color_state["f_dc"] is f_dc and color_state["f_rest"] is f_rest
                                        and all(_ok32(color_state[k]) and color_state[k].numel() == ref.numel()
                                                for k, ref in (("m_dc", f_dc), ("v_dc", f_dc), ("m_rest", f_rest), ("v_rest", f_rest)))):
        color_state = None
    K = fused._intrinsics(scene, W, H, dev)
    rdk = fused._rdk_cached(scene, H, W)
    if not _ok32(rdk, (H, W)) or not _ok32(K, (3, 3)):
        STATS["fallback_layout"] += 1
        return None, None
    eps2d = scene.args.low_pass_filter_eps if hasattr(scene, "args") else scene.eps2d
    A = plan.args
    # ---- pointers of the scene's / keyframe's own tensors: rewritten only when one of the objects changed
    cs = color_state
    bound = leaves + (cls_id, d_max, f_dc, f_rest, r6, t, E, gt, mono, rdk, K) + ((cs["m_dc"], cs["v_dc"], cs["m_rest"], cs["v_rest"], cs["lr_dc"], cs["lr_rest"]) if cs else ())
    ptrs = tuple(x.data_ptr() for x in bound)
    if ptrs != plan.bound_ptrs or len(bound) != len(plan.bound) or any(a is not b for a, b in zip(bound, plan.bound)):
        (A.xyz, A.opacity_raw, A.scaling_raw, A.rotation, A.local_feat, A.global_feat, A.W1, A.b1, A.W2, A.b2) = (x.data_ptr() for x in leaves)
        A.cls_id, A.d_max, A.f_dc, A.f_rest = cls_id.data_ptr(), d_max.data_ptr(), f_dc.data_ptr(), f_rest.data_ptr()
        A.r6, A.t, A.exposure, A.gt, A.mono, A.rdk, A.Kmat = (r6.data_ptr(), t.data_ptr(), E.data_ptr(), gt.data_ptr(), mono.data_ptr(),
                                                               rdk.data_ptr(), K.data_ptr())
        if cs:
            (A.exp_avg_dc, A.exp_avg_sq_dc, A.exp_avg_rest, A.exp_avg_sq_rest, A.lr_dc, A.lr_rest) = (
                cs["m_dc"].data_ptr(), cs["v_dc"].data_ptr(), cs["m_rest"].data_ptr(), cs["v_rest"].data_ptr(), cs["lr_dc"].data_ptr(),
                cs["lr_rest"].data_ptr())
        plan.bound, plan.bound_ptrs = bound, ptrs
    if cs:
        A.color_adam = 1
        A.adam_b1, A.adam_b2, A.adam_eps = float(cs["betas"][0]), float(cs["betas"][1]), float(cs["eps"])
    else:
        A.color_adam = 0
        v_dc, v_rest = plan.color_grads(f_dc, f_rest)
    pose_grad = bool(r6.requires_grad or t.requires_grad)
    lam, wd = float(scene.lambda_dssim), float(keyframe.depth_loss_weight)
    A.sh_K, A.sh_degree, A.mask_outliers, A.pose_grad = sh_K, deg, 0 if is_important else 1, int(pose_grad)
    A.eps2d, A.near_plane, A.far_plane, A.radius_clip = float(eps2d), 0.01, 1e10, 0.0
    A.lambda_dssim, A.depth_weight, A.ssim_grad_scale = lam, wd, -lam / float(3 * H * W)
    A.scaling_reg_factor = float(getattr(scene, "scaling_reg_factor", 0.0) or 0.0)     # h3dgsv3.py:443-449, inside the call since ABI v19
    timer = rasterizer._TIMER
    A.time_mask = 0 if timer is None else sum(1 << i for i, s in enumerate(_Lazy.stages) if timer.only is None or s in timer.only)
    with torch.no_grad(), _lib.on_device(dev):
        bg = torch.rand(3, device=dev)                                   # the reference's draw (h3dgsv3.py:421), same generator stream
        loss = torch.empty((), dtype=torch.float32, device=dev)
        invdepth = torch.empty(1, H, W, dtype=torch.float32, device=dev)  # Keyframe.latest_invdepth outlives the step: never a plan buffer
        A.bg, A.loss, A.invdepth = bg.data_ptr(), loss.data_ptr(), invdepth.data_ptr()
        stream = _lib.raw_stream(dev)
        lock = getattr(scene, "lock", None)
        if lock is not None:
            lock.acquire()
        try:
            rc = lib.adk_mapper_step(ctypes.byref(A), ctypes.byref(plan.out), stream)
            if rc == ADK_STEP_ECAPACITY:
                STATS["capacity_retries"] += 1
                plan.set_capacity(int(plan.out.n_isects * 1.25))
                rc = lib.adk_mapper_step(ctypes.byref(A), ctypes.byref(plan.out), stream)
        finally:
            if lock is not None:
                lock.release()
    if rc == ADK_STEP_EROUTE:
        # a tile list too long for the tile-local sort: this frame needs the global route, which the per-stage chain takes
        STATS["fallback_route"] += 1
        plan.route_miss += 1
        if plan.route_miss >= 3:
            plan.skip_until = plan.calls + 100   # do not pay a wasted forward per step while the view stays like this
            plan.route_miss = 0
        return None, bg
    if rc != 0:
        plan.t["cam_grad"].zero_()
        stage = _Lazy.stages[plan.out.stage] if 0 <= plan.out.stage < len(_Lazy.stages) else "?"
        _lib.check(rc, f"adk_mapper_step (stage {stage})")
    plan.route_miss = 0
    STATS["native"] += 1
    STATS["wait_ns"] += int(plan.out.wait_ns)
    _small_inverse.check()   # behind the call's own host wait: a singular 4x4 met by the wrapped inverse since the last step raises HERE (one pinned read)
    if plan.out.max_tile > 8192:
        STATS["long_list_steps"] += 1      # a tile list above 8 192 entries: sorted by the long-list kernel inside the call (round 5)
    n_isects = int(plan.out.n_isects)
    rasterizer._CAPACITY_HINT[(dev.index, W, H, tile_px[0])] = n_isects
    rasterizer.LAST_STATS.update(N=N, I=n_isects, width=W, height=H, tile_px=tile_px)
    if n_isects > 0.92 * A.isect_capacity:      # grow before the map does, not in the middle of a step
        plan.set_capacity(int(n_isects * 1.25))
    # ---- the gradients, where the optimisers look for them
    g = plan.grads
    for leaf, key in zip(leaves, ("xyz", "opacity", "scaling", "rotation", "local_feat", "global_feat", "W1", "b1", "W2", "b2")):
        if leaf.requires_grad:
            leaf.grad = g[key]
    if not cs:
        f_dc.grad, f_rest.grad = v_dc, v_rest
    if E.requires_grad:
        if e_stepped:
            E.grad = g["exposure"]
        elif E.grad is None:
            E.grad = g["exposure"].clone()     # never the plan's buffer: this one is accumulated into, step after step
        else:
            E.grad.add_(g["exposure"])
    if pose_grad:
        if r6.requires_grad:
            r6.grad = g["r6"]
        if t.requires_grad:
            t.grad = g["t"]
    fused._apply_steps(scene, keyframe, plan.vis, plan.gvis, invdepth)
    return loss, None


def drain_timings(lib=None):
    """{stage: (sum_ms, min_ms, count)} of the event pairs the native steps recorded since the last call (waits for them)."""
    if lib is None and not _lib.is_loaded():
        return {}                      # nothing can have been recorded: do not load (or require) the library for this
    if not _ensure_parsed():
        return {}
    lib = lib or _lib.load()
    stages = _Lazy.stages
    n = len(stages)
    S, M, C = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_int64 * n)()
    lib.adk_mapper_step_timings(S, M, C)
    return {stages[i]: (S[i], M[i], int(C[i])) for i in range(n) if C[i] > 0}
