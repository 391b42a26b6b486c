#!/usr/bin/env python
"""Diagnostic: ONE optimisation step of the PSNR-proxy scene on the HIP path vs the CPU-oracle path, element by element."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.set_num_threads(1)
import artdeco_amd
artdeco_amd.install_dropins()
import test_psnr_proxy as T
from harness import mapper as gmap
dev = torch.device("cuda:0")
cmap = T._cpu_mapper()
c = gmap.synthetic_cloud(T.N, T.W, T.H, seed=3, sigma_px=2.5)
g = torch.Generator().manual_seed(5)
poses = []
for _ in range(T.NKF):
    Rt = torch.eye(4); Rt[:3, 3] = 0.05 * torch.randn(3, generator=g); poses.append(Rt)
ts = cmap.MapperScene(T.W, T.H, c["fx"], "cpu")
with torch.no_grad():
    ts.mlp_cov[2].weight.zero_(); ts.mlp_cov[2].bias.copy_(torch.tensor([0., 0, 0, 1, 1, 1, 1]))
    for p in ts.mlp_cov[0].parameters(): p.zero_()
op = c["opacities"].clamp(1e-4, 1 - 1e-4)
ts.set_gaussians(c["means"], c["quats"], torch.log(2.0 * c["scales"]), torch.log(op / (1 - op)), c["sh"], seed=0)
targets = []
with torch.no_grad():
    for Rt in poses:
        pkg = ts.render(T.W, T.H, Rt, torch.full((3,), 0.5)); targets.append((pkg["render"].clamp(0, 1).contiguous(), pkg["invdepth"].contiguous()))
truth = (c, targets, poses)
cpu, gpu = T._build(cmap, "cpu", truth), T._build(gmap, dev, truth)
bg = torch.tensor([0.3, 0.6, 0.1])
real = torch.rand
torch.rand = lambda *s, **k: bg.to(k.get("device", "cpu")) if s == (3,) else real(*s, **k)
grads = {}
for name, sc in (("cpu", cpu), ("gpu", gpu)):
    orig = sc.optimizer.step
    def spy(vis, N, gvis, NG, _o=orig, _sc=sc, _n=name):
        grads[_n] = {k: v["val"].grad.detach().cpu().clone() for k, v in _sc.gaussian_params.items() if v["val"].is_floating_point() and v["val"].grad is not None}
        grads[_n]["vis"] = vis.cpu().clone()
        return _o(vis, N, gvis, NG)
    sc.optimizer.step = spy
    loss = sc.optimization_step(0, is_important=True)
    print(name, "loss", float(loss))
torch.rand = real
vc, vg = grads["cpu"]["vis"], grads["gpu"]["vis"]
print("visible rows cpu/gpu", int(vc.sum()), int(vg.sum()), "differ", int((vc != vg).sum()))
for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "local_feat"):
    if k not in grads["cpu"] or k not in grads["gpu"]:
        print(k, "grad missing on", [n for n in ("cpu", "gpu") if k not in grads[n]]); continue
    a, b = grads["cpu"][k].double(), grads["gpu"][k].double()
    rel = float((a - b).norm() / a.norm())
    za, zb = (a == 0), (b == 0)
    sign_flip = ((a * b) < 0)
    print(f"{k:10s} grad rel_l2 {rel:.2e}  zero-only-cpu {int((za & ~zb).sum())} zero-only-gpu {int((zb & ~za).sum())} sign-flips {int(sign_flip.sum())} of {a.numel()}"
          f"  |g| of flipped: max {float(a[sign_flip].abs().max()) if sign_flip.any() else 0:.2e} (max |g| {float(a.abs().max()):.2e})")
    pa, pb = cpu.gaussian_params[k]["val"].detach(), gpu.gaussian_params[k]["val"].detach().cpu()
    d = (pa - pb).abs()
    print(f"           param after step: elements differing > 1e-7: {int((d > 1e-7).sum())}, max diff {float(d.max()):.3e}")
