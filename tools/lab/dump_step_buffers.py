#!/usr/bin/env python
"""DEV TOOL: the plan buffers of ONE native optimisation step (fixed seeds) -> a .pt file, for comparing two builds of the library.
    ARTDECO_HIP_LIB=... python tools/lab/dump_step_buffers.py out.pt   |   python tools/lab/dump_step_buffers.py --compare a.pt b.pt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

if sys.argv[1] == "--compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        x, y = a[k], b[k]
        if x.dtype.is_floating_point:
            ne = (x.view(torch.int32) != y.view(torch.int32)) if x.dtype == torch.float32 else (x != y)
        else:
            ne = x != y
        n = int(ne.sum())
        extra = ""
        if n and x.dtype.is_floating_point:
            d = (x.double() - y.double()).abs()
            extra = f" max abs diff {float(d.max()):.3e} rel {float((d / (y.double().abs() + 1e-30)).max()):.3e} first rows {ne.reshape(ne.shape[0], -1).any(1).nonzero().flatten()[:5].tolist()}"
        print(f"{k:18s} {tuple(x.shape)} differing elements {n}{extra}")
    sys.exit(0)

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd import fused
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_fused_glue import _scene

dev = torch.device("cuda:0")
sc = _scene(dev, N=6000, seed=4)
fused.patch_scene_model(sc)
torch.manual_seed(20)
sc.optimization_step(0, is_important=True)
torch.cuda.synchronize()
plan = next(iter(sc.__dict__["_adk_step_plans"].values()))
n = plan.n
keep = {k: plan.t[k][:n].detach().cpu().clone() for k in ("opac", "scale", "quat", "sel", "rec", "radii", "depth_keys", "tiles_per_gauss")}
keep["render_colors"] = plan.t["render_colors"].cpu().clone()
torch.save(keep, sys.argv[1])
print("saved", sys.argv[1], {k: tuple(v.shape) for k, v in keep.items()})
