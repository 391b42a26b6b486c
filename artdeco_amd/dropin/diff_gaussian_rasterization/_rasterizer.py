"""GaussianRasterizationSettings / GaussianRasterizer adapter (filled in once the gsplat path exists)."""
from __future__ import annotations

from typing import NamedTuple

import torch


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
        raise NotImplementedError("GaussianRasterizer adapter not wired yet")
