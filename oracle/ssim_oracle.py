"""CPU oracle for fused-SSIM -- TEST INFRASTRUCTURE, never on the product path.

Restates the reference's own pure-PyTorch SSIM, the function the reference's
test pins its CUDA kernel against:
  Reconstruct/submodules/fused-ssim/tests/test.py:14-54  (gaussian / create_window / ssim / _ssim)
plus the "valid" crop of fused_ssim/__init__.py:13-14 and the C1/C2 of :35-36.
Gradients come from autograd over this restatement (what tests/test.py:90-91 compares).

Parity: PINNED -- tests/golden/ssim_*.npz are produced by importing the reference's
test.py itself (tests/golden/make_golden.py) and this oracle is checked against them.
"""
from __future__ import annotations

from math import exp

import torch
import torch.nn.functional as F

C1 = 0.01 ** 2
C2 = 0.03 ** 2


def gaussian(window_size: int = 11, sigma: float = 1.5) -> torch.Tensor:
    # tests/test.py:14-16 -- built in fp32 from python floats
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def create_window(window_size: int, channel: int, dtype=torch.float32) -> torch.Tensor:
    # tests/test.py:18-22
    w1 = gaussian(window_size, 1.5).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous().to(dtype)


def ssim_map(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    """Per-pixel SSIM map with zero 'same' padding: tests/test.py:35-49."""
    channel = img1.size(-3)
    window = create_window(window_size, channel, img1.dtype).to(img1.device)  # tests/test.py:28-30 (window.cuda(...).type_as(img1))
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    return ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))


def fused_ssim_oracle(img1: torch.Tensor, img2: torch.Tensor, padding: str = "same") -> torch.Tensor:
    """Scalar mean SSIM with the reference's padding modes (fused_ssim/__init__.py:34-42)."""
    m = ssim_map(img1, img2)
    if padding == "valid":
        m = m[:, :, 5:-5, 5:-5]
    return m.mean()
