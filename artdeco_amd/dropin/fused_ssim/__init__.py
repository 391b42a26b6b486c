"""Drop-in for the `fused_ssim` package (Reconstruct/submodules/fused-ssim/fused_ssim/__init__.py).

Same public surface -- `fused_ssim(img1, img2, padding="same", train=True)`,
`FusedSSIMMap`, `allowed_padding` -- same autograd contract (gradient w.r.t.
img1 only, __init__.py:32), HIP kernels underneath.  Call sites in ARTDECO:
Reconstruct/scene/scene_models/h3dgsv3.py:441 (training loss) and :545 (eval).
"""
import torch

from fused_ssim_cuda import fusedssim, fusedssim_backward

allowed_padding = ["same", "valid"]


class FusedSSIMMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, C1, C2, img1, img2, padding="same", train=True):
        ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = fusedssim(C1, C2, img1, img2, train)
        if padding == "valid":
            ssim_map = ssim_map[:, :, 5:-5, 5:-5]
        ctx.save_for_backward(img1.detach(), img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)
        ctx.C1, ctx.C2, ctx.padding = C1, C2, padding
        return ssim_map

    @staticmethod
    def backward(ctx, opt_grad):
        img1, img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = ctx.saved_tensors
        dL_dmap = opt_grad
        if ctx.padding == "valid":
            dL_dmap = torch.zeros_like(img1)
            dL_dmap[:, :, 5:-5, 5:-5] = opt_grad
        grad = fusedssim_backward(ctx.C1, ctx.C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)
        return None, None, grad, None, None, None


def fused_ssim(img1, img2, padding="same", train=True):
    C1 = 0.01 ** 2
    C2 = 0.03 ** 2
    assert padding in allowed_padding
    img1 = img1.contiguous()
    ssim_map = FusedSSIMMap.apply(C1, C2, img1, img2, padding, train)
    return ssim_map.mean()
