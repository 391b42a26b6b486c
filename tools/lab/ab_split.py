#!/usr/bin/env python
"""Same-box A/B: one wave per 16x16 tile (mode 0) against one wave per 8x8 quadrant (1) or per 16x8 half (2) of it
(ADK_RASTER_SPLIT_FWD / _BWD) on the stationary optimisation step, alternating in one process; also checks that the forward is bit-identical and how far the backward's sums move.
    python tools/lab/ab_split.py [N W H]..."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd import fused, rasterizer
from harness import mapper

dev = torch.device("cuda:0")
cfgs = [(200_000, 648, 486), (1_000_000, 512, 384), (1_000_000, 800, 600), (1_000_000, 1024, 768), (1_000_000, 1280, 720), (1_000_000, 1440, 810),
        (1_000_000, 1600, 900), (1_000_000, 1920, 1080), (4_000_000, 2592, 1944)]
if len(sys.argv) > 3:
    cfgs = [tuple(int(x) for x in sys.argv[1:4])]


def setmode(f, b):
    os.environ["ADK_RASTER_SPLIT_FWD"], os.environ["ADK_RASTER_SPLIT_BWD"] = f, b


def render_and_grads(scene):
    from gsplat.rendering import rasterization
    g = torch.Generator(device="cpu").manual_seed(3)
    N = scene.xyz.shape[0]
    means = scene.xyz.detach().clone().requires_grad_(True)
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=g).to(dev), dim=-1)
    scales = (0.01 + 0.02 * torch.rand(N, 3, generator=g)).to(dev).requires_grad_(True)
    opac = (0.05 + 0.9 * torch.rand(N, generator=g)).to(dev).requires_grad_(True)
    cols = torch.rand(N, 3, generator=g).to(dev).requires_grad_(True)
    W, H = scene.width, scene.height
    K = torch.tensor([[scene.f, 0, W / 2], [0, scene.f, H / 2], [0, 0, 1]], device=dev)[None]
    out, alpha, meta = rasterization(means, quats, scales, opac, cols, torch.eye(4, device=dev)[None], K, W, H, packed=False)
    wgt = torch.rand(out.shape, generator=g).to(dev)
    (out * wgt).sum().backward()
    return out.detach(), alpha.detach(), [means.grad.clone(), scales.grad.clone(), opac.grad.clone(), cols.grad.clone()]


out = {}
for N, W, H in cfgs:
    scene = mapper.build_synthetic_mapper(N, W, H, dev, seed=0, targets="render")
    fused.patch_scene_model(scene)
    res = {}
    setmode("0", "0")
    o0, a0, g0 = render_and_grads(scene)
    setmode("1", "1")
    o1, a1, g1 = render_and_grads(scene)
    res["fwd_bit_identical"] = bool(torch.equal(o0, o1) and torch.equal(a0, a1))
    res["bwd_max_rel"] = [float((x - y).abs().max() / x.abs().max().clamp_min(1e-30)) for x, y in zip(g0, g1)]
    for rep in range(2):
        for mode in ("00", "11", "22"):
            setmode(mode[0], mode[1])
            for i in range(5):
                scene.optimization_step(i % 4)
            t = rasterizer.StageTimer()
            rasterizer.set_stage_timer(t)
            for i in range(20):
                scene.optimization_step(i % 4)
            rasterizer.set_stage_timer(None)
            sm = t.summary_ms()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(30):
                scene.optimization_step(i % 4)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 30 * 1e3
            r = {k: round(v["mean_ms"], 4) for k, v in sm.items() if k in ("raster_fwd", "raster_bwd")}
            r["step_ms"] = round(ms, 4)
            res.setdefault("fwd%s_bwd%s" % (mode[0], mode[1]), []).append(r)
    res["I"] = rasterizer.LAST_STATS["I"]
    res["tiles"] = ((W + 15) // 16) * ((H + 15) // 16)
    out[f"{N}/{W}x{H}"] = res
    print(f"{N}/{W}x{H}", json.dumps(res), flush=True)
    del scene
    torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_ab_split.json"), "w"), indent=1)
