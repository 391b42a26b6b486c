"""Generate tests/golden/ssim_*.npz by importing the REFERENCE's own pure-PyTorch SSIM.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_ssim.py
The reference module Reconstruct/submodules/fused-ssim/tests/test.py imports the CUDA
extension and pytorch_msssim at module level; both are stubbed so that only its
`ssim()` (test.py:24-54) -- the function its own asserts compare the kernel to -- runs.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/Reconstruct/submodules/fused-ssim/tests/test.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_ssim():
    for name, attrs in (("fused_ssim", {"fused_ssim": None}), ("pytorch_msssim", {"SSIM": None})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("ref_fused_ssim_test", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.ssim


def main():
    ref_ssim = load_reference_ssim()
    cases = {"ssim_small": (2, 3, 37, 45, 0), "ssim_ragged": (1, 2, 70, 131, 1), "ssim_tiny": (1, 1, 7, 9, 2)}
    for name, (B, CH, H, W, seed) in cases.items():
        g = torch.Generator().manual_seed(seed)
        img1 = torch.rand(B, CH, H, W, generator=g)
        img2 = torch.rand(B, CH, H, W, generator=g)
        x = img1.clone().requires_grad_(True)
        val = ref_ssim(x, img2)  # size_average=True -> mean over everything
        val.backward()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), img1=img1.numpy(), img2=img2.numpy(),
                            ssim=val.detach().numpy(), grad=x.grad.numpy())
        print(name, float(val))


if __name__ == "__main__":
    main()
