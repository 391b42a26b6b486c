"""GaussianRasterizationSettings / GaussianRasterizer of the on-the-fly-nvs rasteriser fork
[UPSTREAM, not vendored], as ARTDECO's web-viewer scene model drives them
(Reconstruct/webviewer/scene_models.py:559-605):

    settings  = GaussianRasterizationSettings(H, W, tanfovx, tanfovy, bg, scale_modifier, projmatrix,
                                              sh_degree, campos, prefiltered, debug)
    color, invdepth, mainGaussID, radii = GaussianRasterizer(settings)(
        means3D, means2D, opacities, dc, shs, scales, rotations, viewmatrix)

This is an ADAPTER over the same HIP kernels that serve gsplat.rendering.rasterization (SURVEY.md
8a6): the Inria-style conventions are translated at the boundary --
  * `viewmatrix` arrives transposed (row-vector convention, scene_models.py:518) -> viewmat = viewmatrix.T;
  * intrinsics come from tanfov with a centred principal point (what `projmatrix` encodes);
  * colours = SH(dc ++ shs) of the active degree, composited over `bg`;
  * invdepth  = alpha-composited 1/z  [1,H,W];  mainGaussID = id of the Gaussian with the largest
    alpha*T per pixel, -1 where nothing was drawn [1,H,W] int32;  radii = max(radius_x, radius_y) [N];
  * the 2D low-pass is the classic 0.3 px^2 of that rasteriser family.
Gradients flow to means3D, opacities, dc, shs, scales, rotations (and viewmatrix); `means2D` is
accepted for signature compatibility and receives no gradient.
"""
from __future__ import annotations

from typing import NamedTuple

import torch

from artdeco_amd.rasterizer import render_camera


class GaussianRasterizationSettings(NamedTuple):
    """11 positional fields, in the order used at Reconstruct/webviewer/scene_models.py:559-571."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(torch.nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, dc, shs, scales, rotations, viewmatrix):
        s = self.raster_settings
        H, W = int(s.image_height), int(s.image_width)
        dev = means3D.device
        N = means3D.shape[0]
        fx, fy = W / (2.0 * float(s.tanfovx)), H / (2.0 * float(s.tanfovy))
        K = torch.tensor([[fx, 0.0, W / 2.0], [0.0, fy, H / 2.0], [0.0, 0.0, 1.0]], dtype=torch.float32, device=dev)
        viewmat = viewmatrix.transpose(0, 1).contiguous()
        out = render_camera(means3D, rotations, scales * float(s.scale_modifier), opacities.reshape(N),
                            dc.reshape(N, 1, 3), viewmat, K, W, H, sh_degree=int(s.sh_degree), eps2d=0.3,
                            inv_depth=True, want_main_ids=True, sh_rest=shs.reshape(N, -1, 3))
        col4, alphas, radii2 = out[0], out[1], out[2]
        main_ids = out[9]
        bg = s.bg.to(dev).reshape(3)
        color = col4[..., :3].permute(2, 0, 1) + (1.0 - alphas.permute(2, 0, 1)) * bg[:, None, None]
        invdepth = col4[..., 3:4].permute(2, 0, 1)
        return color, invdepth, main_ids[None], radii2.max(dim=1).values
