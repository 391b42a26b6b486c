#!/usr/bin/env python
"""DEV TOOL (round 6, finding 62): binning stage times on (a) SURVEY 8(d)'s cloud, (b) the same with a few screen-filling Gaussians,
(c) a close-up (every Gaussian grown x12: the typical rectangle above the wave-cooperative threshold).
    ARTDECO_HIP_LIB=artdeco_amd/lib/libartdeco_hip.<variant>.so python tools/lab/bin_big_lab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd import rasterizer
from gsplat.rendering import rasterization
from oracle import gsplat_oracle as go

dev = torch.device("cuda:0")
W, H = 1920, 1080
for name, N, n_big, grow in (("8(d) cloud", 1_000_000, 0, 1.0), ("+ 16 screen-filling", 1_000_000, 16, 400.0), ("close-up x12", 60_000, 60_000, 12.0)):
    sc = go.synthetic_scene(N, W, H, seed=0)
    if n_big:
        idx = torch.randperm(N, generator=torch.Generator().manual_seed(1))[:n_big]
        sc["scales"] = sc["scales"].clone(); sc["scales"][idx] *= grow
    t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}
    def run():
        return rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], t["viewmat"][None], t["K"][None], W, H, render_mode="RGB+D",
                             rasterize_mode="classic", absgrad=False, packed=False, sh_degree=3, eps2d=0.01)
    for _ in range(3):
        run()
    tm = rasterizer.StageTimer()
    rasterizer.set_stage_timer(tm)
    for _ in range(10):
        run()
    rasterizer.set_stage_timer(None)
    sm = tm.summary_ms()
    print(f"{os.path.basename(os.environ.get('ARTDECO_HIP_LIB', 'default')):28s} {name:22s} I = {rasterizer.LAST_STATS.get('I'):9d}  " +
          "  ".join(f"{k} {sm[k]['mean_ms']:.4f}" for k in ("bin_count", "bin_scatter", "bin_sort", "raster_fwd") if k in sm), flush=True)
