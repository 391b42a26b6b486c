"""`gsplat.rendering.rasterization` on MI355X.

Signature and return contract follow gsplat >= 1.5 [UPSTREAM, not vendored]; ARTDECO's call is
Reconstruct/scene/scene_models/h3dgsv3.py:664-680 and it reads `colors[...,0:3]`, `colors[...,3:4]`,
`alphas` and `meta['radii']` (:682-689).  Differentiable w.r.t. means, quats, scales, opacities,
colors and viewmats.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from artdeco_amd import rasterizer as _rasterizer
from artdeco_amd.rasterizer import render_camera


class _Meta(dict):
    """gsplat's meta dictionary.  `flatten_ids` / `isect_offsets` are upstream's 16x16-tile lists; when the render used the wider
    internal tiles (artdeco_amd.rasterizer.default_tile_px) they are built on first access from the finished projection --
    ARTDECO itself only reads `radii` (h3dgsv3.py:689) and never pays for them."""

    def __init__(self, *a, lazy=None, **k):
        super().__init__(*a, **k)
        self._lazy = lazy

    def _materialise(self):
        if self._lazy is not None:
            fn, self._lazy = self._lazy, None
            flat, offs = fn()
            dict.__setitem__(self, "flatten_ids", flat)
            dict.__setitem__(self, "isect_offsets", offs)

    def __getitem__(self, key):
        if key in ("flatten_ids", "isect_offsets"):
            self._materialise()
        return super().__getitem__(key)

    def get(self, key, default=None):
        if key in ("flatten_ids", "isect_offsets"):
            self._materialise()
        return super().get(key, default)

    # every way of reading the values without naming a key sees the materialised lists, never the None placeholders
    def items(self):
        self._materialise()
        return super().items()

    def values(self):
        self._materialise()
        return super().values()

    def copy(self):
        self._materialise()
        return _Meta(super().copy())

    def pop(self, key, *default):
        if key in ("flatten_ids", "isect_offsets"):
            self._materialise()
        return super().pop(key, *default)

_RENDER_MODES = ("RGB", "D", "ED", "RGB+D", "RGB+ED")


def rasterization(
    means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Tensor, viewmats: Tensor, Ks: Tensor,
    width: int, height: int, near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0,
    eps2d: float = 0.3, sh_degree: Optional[int] = None, packed: bool = True, tile_size: int = 16,
    backgrounds: Optional[Tensor] = None, render_mode: str = "RGB", sparse_grad: bool = False,
    absgrad: bool = False, rasterize_mode: str = "classic", channel_chunk: int = 32, distributed: bool = False,
    camera_model: str = "pinhole", covars: Optional[Tensor] = None, **unsupported,
) -> Tuple[Tensor, Tensor, Dict]:
    """-> (render_colors [C,H,W,X], render_alphas [C,H,W,1], meta).

    X = 3 ("RGB"), 1 ("D"/"ED") or 4 ("RGB+D"/"RGB+ED": accumulated / expected depth in channel 3).
    meta carries radii [C,N,2] (int32), means2d, depths, conics, opacities, tiles_per_gauss,
    flatten_ids, isect_offsets, tile_width/height, ... like upstream (isect_ids on request via
    the extra keyword `return_isect_ids=True`).
    """
    want_isect_ids = bool(unsupported.pop("return_isect_ids", False))
    if unsupported:
        raise NotImplementedError(f"gsplat.rasterization (artdeco_amd): unsupported arguments {sorted(unsupported)}")
    if packed:
        raise NotImplementedError("artdeco_amd implements packed=False (what ARTDECO passes, h3dgsv3.py:677)")
    if tile_size != 16:
        raise NotImplementedError("tile_size must be 16")
    if rasterize_mode != "classic":
        raise NotImplementedError('rasterize_mode must be "classic" (h3dgsv3.py:675)')
    if camera_model != "pinhole" or covars is not None or sparse_grad or absgrad or distributed:
        raise NotImplementedError("only pinhole / quats+scales / dense gradients are implemented")
    if render_mode not in _RENDER_MODES:
        raise ValueError(f"render_mode must be one of {_RENDER_MODES}")
    if means.dim() != 2 or means.shape[-1] != 3:
        raise ValueError(f"means must be [N,3], got {tuple(means.shape)}")
    N = means.shape[0]
    if quats.shape != (N, 4) or scales.shape != (N, 3) or opacities.shape != (N,):
        raise ValueError("quats [N,4], scales [N,3], opacities [N] expected")
    if viewmats.dim() != 3 or viewmats.shape[1:] != (4, 4) or Ks.shape != (viewmats.shape[0], 3, 3):
        raise ValueError("viewmats [C,4,4] and Ks [C,3,3] expected")
    C = viewmats.shape[0]
    depth_only = render_mode in ("D", "ED")
    if not depth_only:
        if sh_degree is None:
            if not (colors.dim() == 2 or (colors.dim() == 3 and colors.shape[0] == C)):
                raise ValueError("colors must be [N,3] or [C,N,3] when sh_degree is None")
        elif not (colors.dim() == 3 or (colors.dim() == 4 and colors.shape[0] == C)):
            raise ValueError("SH colors must be [N,K,3] or [C,N,K,3]")
    if backgrounds is not None and backgrounds.shape[0] != C:
        raise ValueError("backgrounds must be [C, channels]")

    outs = []
    for c in range(C):
        cols = colors
        if not depth_only and ((sh_degree is None and colors.dim() == 3) or (sh_degree is not None and colors.dim() == 4)):
            cols = colors[c]
        bg4 = None
        if backgrounds is not None:
            b = backgrounds[c]
            bg4 = torch.zeros(4, dtype=torch.float32, device=b.device)
            bg4[: b.shape[0]] = b  # channel order matches the output (rgb[, depth] or depth)
        outs.append(render_camera(means, quats, scales, opacities, cols, viewmats[c], Ks[c], width, height,
                                  sh_degree=sh_degree, eps2d=eps2d, near_plane=near_plane, far_plane=far_plane,
                                  radius_clip=radius_clip, backgrounds=bg4, depth_only=depth_only,
                                  want_isect_ids=want_isect_ids))

    col4 = torch.stack([o[0] for o in outs])      # [C,H,W,4]
    alphas = torch.stack([o[1] for o in outs])    # [C,H,W,1]
    if render_mode == "RGB":
        render = col4[..., :3]
    elif render_mode in ("D", "ED"):
        render = col4[..., :1]
    else:
        render = col4
    if render_mode in ("ED", "RGB+ED"):
        render = torch.cat([render[..., :-1], render[..., -1:] / alphas.clamp(min=1e-10)], dim=-1)

    rec = torch.stack([o[3] for o in outs])       # [C,N,12] packed splat records
    dch = 8 if depth_only else 11
    # the lists in the outputs are gsplat's when they were asked for or when the internal tiles are 16x16 anyway
    # decided per CAMERA from the lists themselves (one offset per 16x16 tile?), not from process-global state: a camera whose wide lists
    # overflowed falls back to the 16x16 route on its own, and a viewer thread may render concurrently
    grid16 = ((height + 15) // 16, (width + 15) // 16)
    gsplat_lists = want_isect_ids or all(tuple(o[6].shape[-2:]) == grid16 for o in outs)
    meta = _Meta({
        "camera_ids": None, "gaussian_ids": None,
        "radii": torch.stack([o[2] for o in outs]),
        "means2d": rec[..., 0:2], "depths": rec[..., dch], "conics": rec[..., 4:7],
        "opacities": opacities[None].expand(C, -1),
        "tile_width": (width + 15) // 16, "tile_height": (height + 15) // 16,
        "tiles_per_gauss": torch.stack([o[4] for o in outs]),
        "isect_ids": (torch.cat([o[7] for o in outs]) if C > 1 else outs[0][7]) if want_isect_ids else None,
        "width": width, "height": height, "tile_size": 16, "n_cameras": C,
    })
    if gsplat_lists:
        meta["flatten_ids"] = torch.cat([o[5] + c * N for c, o in enumerate(outs)]) if C > 1 else outs[0][5]
        meta["isect_offsets"] = torch.stack([o[6] for o in outs])
    else:
        def build():
            per = [_rasterizer.gsplat_tile_lists(o[3], o[11], o[4], width, height) for o in outs]
            flat = torch.cat([p[0] + c * N for c, p in enumerate(per)]) if C > 1 else per[0][0]
            return flat, torch.stack([p[1] for p in per])
        dict.__setitem__(meta, "flatten_ids", None)
        dict.__setitem__(meta, "isect_offsets", None)
        meta._lazy = build
    return render, alphas, meta
