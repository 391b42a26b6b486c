#!/usr/bin/env python
"""DEV TOOL: the loop of tests/test_native_step.py::test_every_reuse_of_the_plan_matches_the_per_stage_chain with the forward intermediates of
both paths compared bit for bit at every iteration (which buffer differs first?)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd import fused, rasterizer
from test_fused_glue import _scene, _sync_state

dev = torch.device("cuda:0")
a, b = _scene(dev, N=6000, seed=4), _scene(dev, N=6000, seed=4)
fused.patch_scene_model(a); fused.patch_scene_model(b)
stash = {}
lod_fwd, ras_fwd = fused.FusedLodParams.forward, rasterizer.RasterizeGaussians.forward


def lod_spy(ctx, *args):
    out = lod_fwd(ctx, *args)
    stash["opac"], stash["scale"], stash["quat"] = out[0].clone(), out[1].clone(), out[2].clone()
    return out


def ras_spy(ctx, *args):
    out = ras_fwd(ctx, *args)
    stash["render_colors"], stash["radii"], stash["rec"] = out[0].clone(), out[2].clone(), out[3].clone()
    return out


fused.FusedLodParams.forward = staticmethod(lod_spy)
rasterizer.RasterizeGaussians.forward = staticmethod(ras_spy)
for i in range(8):
    _sync_state(a, b)
    imp, kid = i % 3 != 0, i % 2
    os.environ["ARTDECO_AMD_NATIVE_STEP"] = "1"
    torch.manual_seed(20 + i)
    la = a.optimization_step(kid, is_important=imp)
    plan = next(iter(a.__dict__["_adk_step_plans"].values()))
    n = plan.n
    na = {k: plan.t[k][:n].clone() for k in ("opac", "scale", "quat", "rec", "radii")}
    na["render_colors"] = plan.t["render_colors"].clone()
    inv_a = a.keyframes[kid].latest_invdepth.clone()
    os.environ["ARTDECO_AMD_NATIVE_STEP"] = "0"
    torch.manual_seed(20 + i)
    lb = b.optimization_step(kid, is_important=imp)
    inv_b = b.keyframes[kid].latest_invdepth.clone()
    line = [f"step {i} important={imp} kid={kid} loss_equal={bool(torch.equal(la, lb))} invdepth_equal={bool(torch.equal(inv_a, inv_b))}"]
    for k, x in na.items():
        y = stash[k]
        ne = int((x.view(torch.int32) != y.view(torch.int32)).sum()) if x.dtype == torch.float32 else int((x != y).sum())
        if ne:
            line.append(f"{k}:{ne}")
    print(" ".join(line), flush=True)
