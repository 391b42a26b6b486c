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
    from harness import ref_env
    m.scatter_max = ref_env.cpu_natives()[4]     # update_voxel's scatter_max on the CPU oracle too (the large proxy densifies)
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


# ------------------------------------------------------------------------------------------------------------------------------------------
# The LARGE proxy (round 6): a problem that can FAIL.  An 80 k-Gaussian true scene at 256x192 (a 52 k-Gaussian start map) is reconstructed the way
# run_system.py reconstructs a sequence -- frames arrive one at a time; each becomes a Keyframe, `add_new_gaussians` densifies from it (the
# start map is missing a third of the scene, so the densified Gaussians are what the held-out views of that region see), `add_and_prune`
# re-allocates the map and both Adam moments inside it, and a burst of optimisation steps follows -- once on the HIP path and once on the
# CPU-oracle path, from the same state, the same frames, the same backgrounds and the same uniform draws for the densification masks.
# Held-out PSNR (the reference's, Reconstruct/utils.py:86-87; test views rendered as h3dgsv3.py:523-558 renders them) at five checkpoints.
# ~4 s per CPU-oracle step at this size: twenty-odd minutes, so this is NOT in bench.py's default tail (python harness/psnr_proxy.py --large,
# or bench.py --psnr-large); the small proxy above stays there.
def run_large(dev, n_true=80_000, width=256, height=192, n_frames=10, steps_per_frame=32, n_test=6, checkpoints=5, lr_scale=LR_SCALE,
              cpu_threads=None, sides=("cpu", "hip"), log=None):
    from harness import mapper as gmap
    import torch.nn.functional as F
    t_start = time.perf_counter()
    threads0 = torch.get_num_threads()
    if cpu_threads:
        torch.set_num_threads(int(cpu_threads))
    say = log or (lambda *a: None)
    cmap = cpu_mapper()
    rng_state = torch.get_rng_state()
    torch.manual_seed(0)
    W, H = width, height
    c = gmap.synthetic_cloud(n_true, W, H, seed=4, sigma_px=2.0)
    g = torch.Generator().manual_seed(6)
    poses = []
    for _ in range(n_frames + n_test):
        Rt = torch.eye(4)
        Rt[:3, 3] = 0.04 * torch.randn(3, generator=g)
        poses.append(Rt)

    def neutral_mlp(sc):
        with torch.no_grad():
            sc.mlp_cov[2].weight.zero_()
            sc.mlp_cov[2].bias.copy_(torch.tensor([0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 1.0]))
            for p in sc.mlp_cov[0].parameters():
                p.zero_()

    # the true scene's observations (CPU oracle; identical inputs for both sides): image, inverse depth -> point map + confidence
    truth_scene = cmap.MapperScene(W, H, c["fx"], "cpu")
    neutral_mlp(truth_scene)
    op = c["opacities"].clamp(1e-4, 1 - 1e-4)
    truth_scene.set_gaussians(c["means"], c["quats"], torch.log(2.0 * c["scales"]), torch.log(op / (1 - op)), c["sh"], seed=0)
    frames = []
    with torch.no_grad():
        for Rt in poses:
            pkg = truth_scene.render(W, H, Rt, torch.full((3,), 0.5))
            img, inv = pkg["render"].clamp(0, 1).contiguous(), pkg["invdepth"]
            depth = (1.0 / inv[0].clamp(1e-3, 1e3))
            depth = torch.where(torch.isfinite(depth), depth, torch.full_like(depth, 4.0))
            ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
            pm = torch.stack([(xs - (W - 1) / 2) / c["fx"] * depth, (ys - (H - 1) / 2) / c["fx"] * depth, depth], -1).contiguous()
            conf = 0.4 + 0.5 * torch.rand(H, W, generator=g)
            frames.append(dict(image=img, point_map=pm, conf=conf, Rt=Rt))
    say(f"truth rendered: {n_frames + n_test} views in {time.perf_counter() - t_start:.1f} s")
    test_views = [(fr["Rt"], fr["image"]) for fr in frames[n_frames:]]

    # the start map: the true scene WITHOUT the Gaussians that project into the right third of the image, the rest perturbed
    u = c["means"][:, 0] / c["means"][:, 2] * c["fx"] + W / 2
    keep = u < 0.66 * W
    gq = torch.Generator().manual_seed(12)
    n0 = int(keep.sum())
    means0 = c["means"][keep] + 0.01 * torch.randn(n0, 3, generator=gq)
    sh0 = c["sh"][keep] + 0.25 * torch.randn(c["sh"][keep].shape, generator=gq)
    logit0 = torch.log(op[keep] / (1 - op[keep])) + 0.5 * torch.randn(n0, generator=gq)
    lsc0 = torch.log(2.0 * c["scales"][keep]) + 0.15 * torch.randn(n0, 3, generator=gq)

    def build_side(mapper, device, position_scale=1.0):
        sc = mapper.MapperScene(W, H, c["fx"], device)
        neutral_mlp(sc)
        sc.set_gaussians(means0 * position_scale, c["quats"][keep], lsc0, logit0, sh0, seed=0)
        for k, pd in sc.optimizer.params.items():
            if "lr" in pd and k not in ("cls_id", "d_max", "id"):
                pd["lr"] = pd["lr"] * lr_scale
        sc.lr_dict["xyz"]["lr_init"] *= lr_scale
        return sc

    runs = {}
    if "cpu" in sides:
        runs["cpu"] = (cmap, build_side(cmap, "cpu"))
    if "cpu_perturbed" in sides:     # the proxy's own noise floor: the CPU oracle against ITSELF from positions scaled by 1 + 1e-7
        runs["cpu_perturbed"] = (cmap, build_side(cmap, "cpu", position_scale=1.0 + 1e-7))
    if "hip" in sides:
        from artdeco_amd import fused
        sc = build_side(gmap, dev)
        if not fused.patch_scene_model(sc):
            raise RuntimeError("the fused HIP path did not install on the proxy scene")
        runs["hip"] = (gmap, sc)

    def evaluate(sc):
        with torch.no_grad():
            bg = torch.full((3,), 0.5, device=sc.device)
            return float(np.mean([psnr(sc.render(W, H, Rt.to(sc.device), bg)["render"].clamp(0, 1).cpu(), img) for Rt, img in test_views]))

    total = n_frames * steps_per_frame
    cp_frames = {max(1, round(n_frames * (k + 1) / checkpoints)) for k in range(checkpoints)}     # checkpoints fall on frame boundaries
    bgs = torch.rand(total + 64, 3, generator=torch.Generator().manual_seed(9))
    real_rand, real_rand_like = torch.rand, torch.rand_like
    st = {"i": 0, "side": None, "draw": {}}

    def fake_rand(*size, **kw):
        if size == (3,) and kw.get("generator") is None:
            return bgs[st["i"]].to(kw.get("device", "cpu"))
        return real_rand(*size, **kw)

    def fake_rand_like(t, **kw):
        # the densification's sample masks (`rand_like(p) < ...`, h3dgsv3.py:819): the SAME uniform field on both sides, keyed by the side's own
        # call count (both sides densify the same frames in the same LoD-level order) -- a different draw is a different map, not a numerics error
        k = st["draw"].get(st["side"], 0)
        st["draw"][st["side"]] = k + 1
        field = real_rand(tuple(t.shape), generator=torch.Generator().manual_seed(100_000 + k))
        return field.to(device=t.device, dtype=t.dtype)

    start = {k: evaluate(sc) for k, (_, sc) in runs.items()}
    curve, sizes = [], []
    try:
        for f in range(n_frames):
            for side, (mapper, sc) in runs.items():
                st["side"] = side
                torch.rand, torch.rand_like = fake_rand, fake_rand_like
                fr = frames[f]
                prev = sc.keyframes[-1] if sc.keyframes else None
                to = lambda x: x.to(sc.device)
                kf = mapper.StreamKeyframe(to(fr["image"]), to(fr["Rt"]), to(fr["point_map"]), to(fr["conf"]), torch.tensor([c["fx"]], device=sc.device),
                                           sc.device, index=f, prev_kf=prev, is_test=False, pyr_levels=1)
                sc.add_keyframe(kf)
                n_before = int(sc.xyz.shape[0])
                sc.add_new_gaussians()                     # densify from the new frame (+ add_and_prune inside it)
                n_after = int(sc.xyz.shape[0])
                for s in range(steps_per_frame):
                    st["i"] = f * steps_per_frame + s
                    kid = (f if s % 3 == 0 else (s * 7 + f) % (f + 1))     # the newest keyframe a third of the time, the others in a fixed order
                    sc.optimization_step(kid, is_important=(s % 5 != 4))
                torch.rand, torch.rand_like = real_rand, real_rand_like
                sizes.append((f, side, n_before, n_after))
            done = (f + 1) * steps_per_frame
            if (f + 1) in cp_frames or f == n_frames - 1:
                curve.append((done, {k: evaluate(sc) for k, (_, sc) in runs.items()}, {k: int(sc.xyz.shape[0]) for k, (_, sc) in runs.items()}))
                say(f"frame {f}: step {done}: " + "  ".join(f"{k} {v:.3f} dB" for k, v in curve[-1][1].items()) + f"  map {curve[-1][2]}  ({time.perf_counter() - t_start:.0f} s)")
    finally:
        torch.rand, torch.rand_like = real_rand, real_rand_like
        torch.set_rng_state(rng_state)
        torch.set_num_threads(threads0)
    both = "cpu" in runs and "hip" in runs
    floor = "cpu" in runs and "cpu_perturbed" in runs
    cps = [{"step": s, **{f"{k}_db": round(x, 4) for k, x in v.items()}, **({"delta_db": round(v["hip"] - v["cpu"], 4)} if both else {}),
            **({"noise_floor_db": round(v["cpu_perturbed"] - v["cpu"], 4)} if floor else {}), "map_size": m} for s, v, m in curve]
    return {"start_db": {k: round(v, 4) for k, v in start.items()}, "checkpoints": cps,
            "max_abs_delta_db": max(abs(cp["delta_db"]) for cp in cps) if both and cps else None,
            "max_abs_noise_floor_db": max(abs(cp["noise_floor_db"]) for cp in cps) if floor and cps else None,
            "gaussians_true": n_true, "gaussians_start": n0, "width": W, "height": H, "frames": n_frames, "steps": total, "held_out_views": n_test,
            "densified": [{"frame": f, "side": s, "before": a, "after": b} for f, s, a, b in sizes], "learning_rate_scale": lr_scale,
            "seconds": round(time.perf_counter() - t_start, 1)}


if __name__ == "__main__":
    import argparse
    import json
    import sys
    sys.path.insert(0, ROOT)
    ap = argparse.ArgumentParser()
    ap.add_argument("--large", action="store_true")
    ap.add_argument("--cpu-only", action="store_true", help="the CPU-oracle side alone (no GPU needed): a dry run of the harness")
    ap.add_argument("--gaussians", type=int, default=80_000)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--steps-per-frame", type=int, default=32)
    ap.add_argument("--threads", type=int, default=None)
    ap.add_argument("--noise-floor", action="store_true", help="--large: also the CPU oracle against itself from positions scaled by 1 + 1e-7 (doubles the CPU time)")
    a = ap.parse_args()
    if not a.cpu_only:
        import artdeco_amd
        artdeco_amd.install_dropins()
    devc = None if a.cpu_only else torch.device("cuda:0")
    if a.large:
        r = run_large(devc, a.gaussians, a.width, a.height, a.frames, a.steps_per_frame, cpu_threads=a.threads,
                      sides=(("cpu",) if a.cpu_only else ("cpu", "hip")) + (("cpu_perturbed",) if a.noise_floor else ()), log=lambda *x: print(*x, file=sys.stderr, flush=True))
    else:
        r = run(devc, cpu_threads=a.threads)
    print(json.dumps(r))
