"""PSNR-delta proxy (BASELINE metric "... PSNR delta"; north star "PSNR within 0.1 dB of the CUDA reference") -- bench / test harness.

No dataset and no CUDA reference exist on the GPU box, so the proxy is the same small reconstruction problem -- a perturbed
2 500-Gaussian cloud optimised against images of the true cloud -- solved twice from the same state, with the same keyframe
order and the same random backgrounds: once by the HIP path (drop-in natives + fused glue, on the MI355X) and once by the CPU
ORACLE path (the same host code, `harness/mapper.py`, with every native bound to `oracle/`: fp32-autograd rasteriser, the
reference's torch SSIM, IEEE Adam).  PSNR is the reference's own (`Reconstruct/utils.py:86-87`: 10 log10(1 / mse)), on HELD-OUT
views rendered the way `SceneModel.evaluate` renders test frames (`h3dgsv3.py:523-558`), each model by its own renderer.

The oracle here is the CHECKER, never the thing measured: `tests/test_psnr_proxy.py` and `bench.py`'s `psnr_proxy` object are the
only callers; nothing under `artdeco_amd/` imports this module.

Noise floor (measured, `noise_floor=True`): the CPU path against ITSELF with the initial positions scaled by 1 + 1e-7.  At the
reference's learning rates this tiny problem is chaotic (0.69 dB between two CPU runs); at 0.2 x it is ~0.01 dB while the
reconstruction still climbs 17 -> 31 dB in ~100 steps, so the proxy runs at 0.2 x on both sides.
"""
from __future__ import annotations

import importlib.util
import math
import os
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, W, H, NKF, NTEST, LR_SCALE = 2500, 80, 56, 3, 6, 0.2


def oracle_natives():
    """The four natives of the mapper's step, bound to oracle/ (gsplat.rendering.rasterization, fused_ssim, adamUpdate, adamUpdateBasic)."""
    from oracle import adam_oracle, gsplat_oracle, ssim_oracle

    def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, render_mode, rasterize_mode, absgrad, packed,
                      sh_degree, eps2d):
        assert (render_mode, rasterize_mode, absgrad, packed) == ("RGB+D", "classic", False, False)   # h3dgsv3.py:673-677
        r, a, meta = gsplat_oracle.rasterization(means, quats, scales, opacities, colors, viewmats[0], Ks[0], width, height, sh_degree=sh_degree,
                                                 eps2d=eps2d, grad_dtype=torch.float32)
        return r[None], a[None], {"radii": meta["radii"][None]}

    def fused_ssim(img1, img2, padding="same", train=True):
        return ssim_oracle.fused_ssim_oracle(img1, img2, padding)

    def adam_update(param, grad, exp_avg, exp_avg_sq, visible, lr, b1, b2, eps, N_, M):
        lr_np = lr.detach().numpy() if torch.is_tensor(lr) else np.float32(lr)
        p, m, v = adam_oracle.adam_update_oracle(param.detach().numpy(), grad.numpy(), exp_avg.numpy(), exp_avg_sq.numpy(),
                                                 visible.numpy(), lr_np, b1, b2, eps, N_, M)
        param.data.copy_(torch.from_numpy(p)); exp_avg.copy_(torch.from_numpy(m)); exp_avg_sq.copy_(torch.from_numpy(v))

    def adam_update_basic(param, grad, exp_avg, exp_avg_sq, lr, b1, b2, eps):
        p, m, v = adam_oracle.adam_update_basic_oracle(param.detach().numpy(), grad.numpy(), exp_avg.numpy(), exp_avg_sq.numpy(), lr, b1, b2, eps)
        param.data.copy_(torch.from_numpy(p)); exp_avg.copy_(torch.from_numpy(m)); exp_avg_sq.copy_(torch.from_numpy(v))
    return rasterization, fused_ssim, adam_update, adam_update_basic


def cpu_mapper():
    """A second copy of harness/mapper.py whose natives are the CPU oracles."""
    spec = importlib.util.spec_from_file_location("harness_mapper_cpu_oracle", os.path.join(ROOT, "harness", "mapper.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    ras, ssim, adam, adam_basic = oracle_natives()
    m.gsplat = types.SimpleNamespace(rendering=types.SimpleNamespace(rasterization=ras))
    m.fused_ssim, m.adamUpdate, m.adamUpdateBasic = ssim, adam, adam_basic
    return m


def psnr(a, b):
    """Reconstruct/utils.py:86-87"""
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 10.0 * math.log10(1.0 / max(mse, 1e-20))


def build(mapper, device, truth, position_scale=1.0):
    """Scene whose keyframes observe `truth` (images + inverse depths rendered beforehand) from a PERTURBED copy of it."""
    c, targets, poses = truth
    sc = mapper.MapperScene(W, H, c["fx"], device)
    with torch.no_grad():
        last = sc.mlp_cov[2]
        last.weight.zero_()
        last.bias.copy_(torch.tensor([0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 1.0]))
        for p in sc.mlp_cov[0].parameters():
            p.zero_()
    g = torch.Generator().manual_seed(11)
    op = c["opacities"].clamp(1e-4, 1 - 1e-4)
    means = (c["means"] + 0.01 * torch.randn(N, 3, generator=g)) * position_scale
    sh = c["sh"] + 0.25 * torch.randn(c["sh"].shape, generator=g)
    logit = torch.log(op / (1 - op)) + 0.5 * torch.randn(N, generator=g)
    sc.set_gaussians(means, c["quats"], torch.log(2.0 * c["scales"]) + 0.15 * torch.randn(N, 3, generator=g), logit, sh, seed=0)
    for (img, idepth), Rt in zip(targets, poses):
        sc.add_keyframe(mapper.Keyframe(img.to(device), idepth.to(device), Rt.to(device), torch.device(device)))
    for k, pd in sc.optimizer.params.items():      # see the module docstring: the proxy's noise floor needs smaller steps
        if "lr" in pd and k not in ("cls_id", "d_max", "id"):
            pd["lr"] = pd["lr"] * LR_SCALE
    sc.lr_dict["xyz"]["lr_init"] *= LR_SCALE
    return sc


def run(dev, steps=120, every=20, noise_floor=False, cpu_threads=None):
    """Train the proxy problem `steps` optimisation steps on the HIP path and on the CPU-oracle path; held-out PSNR of both every
    `every` steps.  Returns dict(start_db, checkpoints=[{step, cpu_oracle_db, hip_db, delta_db}], max_abs_delta_db,
    noise_floor_db (CPU vs CPU from positions scaled by 1 + 1e-7; None unless noise_floor), seconds)."""
    from artdeco_amd import fused
    from harness import mapper as gmap
    t_start = time.perf_counter()
    threads0 = torch.get_num_threads()
    if cpu_threads:     # the CPU-oracle side is ~0.3 s per step single-threaded; a handful of threads is where its small tensors stop gaining
        torch.set_num_threads(int(cpu_threads))
    cmap = cpu_mapper()
    rng_state = torch.get_rng_state()
    torch.manual_seed(0)
    c = gmap.synthetic_cloud(N, W, H, seed=3, sigma_px=2.5)
    g = torch.Generator().manual_seed(5)
    poses = []
    for _ in range(NKF + NTEST):   # the last NTEST poses are held out
        Rt = torch.eye(4)
        Rt[:3, 3] = 0.05 * torch.randn(3, generator=g)
        poses.append(Rt)
    # ground-truth observations: the true cloud rendered by the CPU oracle path (identical targets for both runs)
    truth_scene = cmap.MapperScene(W, H, c["fx"], "cpu")
    with torch.no_grad():
        truth_scene.mlp_cov[2].weight.zero_()
        truth_scene.mlp_cov[2].bias.copy_(torch.tensor([0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 1.0]))
        for p in truth_scene.mlp_cov[0].parameters():
            p.zero_()
    op = c["opacities"].clamp(1e-4, 1 - 1e-4)
    truth_scene.set_gaussians(c["means"], c["quats"], torch.log(2.0 * c["scales"]), torch.log(op / (1 - op)), c["sh"], seed=0)
    targets = []
    with torch.no_grad():
        for Rt in poses:
            pkg = truth_scene.render(W, H, Rt, torch.full((3,), 0.5))
            targets.append((pkg["render"].clamp(0, 1).contiguous(), pkg["invdepth"].contiguous()))
    truth = (c, targets[:NKF], poses[:NKF])
    test_views = list(zip(poses[NKF:], [t[0] for t in targets[NKF:]]))

    runs = {"cpu": build(cmap, "cpu", truth), "hip": build(gmap, dev, truth)}
    if not fused.patch_scene_model(runs["hip"]):
        raise RuntimeError("the fused HIP path did not install on the proxy scene")
    if noise_floor:
        runs["cpu_perturbed"] = build(cmap, "cpu", truth, position_scale=1.0 + 1e-7)

    def evaluate(sc):
        """mean PSNR over the held-out views (SceneModel.evaluate, h3dgsv3.py:523-558, renders test frames the same way)"""
        with torch.no_grad():
            bg = torch.full((3,), 0.5, device=sc.device)
            return float(np.mean([psnr(sc.render(W, H, Rt.to(sc.device), bg)["render"].clamp(0, 1).cpu(), img) for Rt, img in test_views]))

    start = {k: evaluate(sc) for k, sc in runs.items()}
    # same random backgrounds on every side: the step draws torch.rand(3, device=...) (h3dgsv3.py:422), whose stream depends on the device
    bgs = torch.rand(steps, 3, generator=torch.Generator().manual_seed(9))
    real_rand = torch.rand
    state = {"i": 0}

    def fake_rand(*size, **kw):
        if size == (3,) and kw.get("generator") is None:
            return bgs[state["i"]].to(kw.get("device", "cpu"))
        return real_rand(*size, **kw)
    curve = []
    try:
        for i in range(steps):
            state["i"] = i
            torch.rand = fake_rand
            for sc in runs.values():
                sc.optimization_step(i % NKF, is_important=(i % 5 != 4))
            torch.rand = real_rand
            if (i + 1) % every == 0:
                curve.append((i + 1, {k: evaluate(sc) for k, sc in runs.items()}))
    finally:
        torch.rand = real_rand
        torch.set_rng_state(rng_state)
        torch.set_num_threads(threads0)
    cps = [{"step": s, "cpu_oracle_db": round(v["cpu"], 4), "hip_db": round(v["hip"], 4), "delta_db": round(v["hip"] - v["cpu"], 4)} for s, v in curve]
    out = {"start_db": round(start["cpu"], 4), "start_delta_db": round(start["hip"] - start["cpu"], 5), "checkpoints": cps,
           "max_abs_delta_db": max(abs(cp["delta_db"]) for cp in cps) if cps else None,
           "noise_floor_db": (max(abs(v["cpu_perturbed"] - v["cpu"]) for _, v in curve) if noise_floor and curve else None),
           "learning_rate_scale": LR_SCALE, "gaussians": N, "width": W, "height": H, "train_views": NKF, "held_out_views": NTEST, "steps": steps,
           "seconds": None}
    out["seconds"] = round(time.perf_counter() - t_start, 2)
    return out
