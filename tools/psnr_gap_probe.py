#!/usr/bin/env python
"""Diagnostic for tests/test_psnr_proxy.py: which part of the HIP path makes its training curve differ from the CPU-oracle one?
Runs the proxy with the GPU scene (a) unfused (drop-in natives, torch glue), (b) fused, (c) fused without the in-backward
colour Adam, and prints the held-out PSNR curves next to the CPU-oracle curve."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

torch.set_num_threads(1)
import artdeco_amd

artdeco_amd.install_dropins()
import test_psnr_proxy as T
from artdeco_amd import fused, rasterizer
from harness import mapper as gmap

dev = torch.device("cuda:0")
cmap = T._cpu_mapper()
c = gmap.synthetic_cloud(T.N, T.W, T.H, seed=3, sigma_px=2.5)
g = torch.Generator().manual_seed(5)
poses = []
for _ in range(T.NKF + T.NTEST):
    Rt = torch.eye(4); Rt[:3, 3] = 0.05 * torch.randn(3, generator=g); poses.append(Rt)
ts = cmap.MapperScene(T.W, T.H, c["fx"], "cpu")
with torch.no_grad():
    ts.mlp_cov[2].weight.zero_(); ts.mlp_cov[2].bias.copy_(torch.tensor([0., 0, 0, 1, 1, 1, 1]))
    for p in ts.mlp_cov[0].parameters():
        p.zero_()
op = c["opacities"].clamp(1e-4, 1 - 1e-4)
ts.set_gaussians(c["means"], c["quats"], torch.log(2.0 * c["scales"]), torch.log(op / (1 - op)), c["sh"], seed=0)
targets = []
with torch.no_grad():
    for Rt in poses:
        pkg = ts.render(T.W, T.H, Rt, torch.full((3,), 0.5))
        targets.append((pkg["render"].clamp(0, 1).contiguous(), pkg["invdepth"].contiguous()))
truth = (c, targets[:T.NKF], poses[:T.NKF])
tv = list(zip(poses[T.NKF:], [t[0] for t in targets[T.NKF:]]))
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bgs = torch.rand(STEPS, 3, generator=torch.Generator().manual_seed(9))
real = torch.rand


def run(kind):
    sc = T._build(cmap, "cpu", truth) if kind == "cpu" else T._build(gmap, dev, truth)
    if kind in ("fused", "fused_nocoloradam"):
        assert fused.patch_scene_model(sc)
    if kind == "fused_nocoloradam":
        fused._color_adam_state_saved = fused._color_adam_state
        fused._color_adam_state = lambda opt: None
    st = {"i": 0}

    def fake(*size, **kw):
        if size == (3,) and kw.get("generator") is None:
            return bgs[st["i"]].to(kw.get("device", "cpu"))
        return real(*size, **kw)
    out = []
    for i in range(STEPS):
        st["i"] = i
        torch.rand = fake
        sc.optimization_step(i % T.NKF, is_important=(i % 5 != 4))
        torch.rand = real
        if (i + 1) % 10 == 0:
            with torch.no_grad():
                bg = torch.full((3,), 0.5, device=sc.device)
                out.append(round(float(np.mean([T._psnr(sc.render(T.W, T.H, Rt.to(sc.device), bg)["render"].clamp(0, 1).cpu(), img) for Rt, img in tv])), 3))
    if kind == "fused_nocoloradam":
        fused._color_adam_state = fused._color_adam_state_saved
    return out


ref = run("cpu")
print("cpu              ", ref)
for kind in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("unfused", "fused", "fused_nocoloradam")):
    r = run(kind)
    print(f"{kind:17s}", r, " delta", [round(a - b, 3) for a, b in zip(r, ref)])
