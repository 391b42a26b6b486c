#!/usr/bin/env python
"""How much of the binning scatter's write amplification (3.9x: isolated 8-byte stores) is the ORDER of the
Gaussians?  The bench cloud is in random order (worst case).  A map built by add_new_gaussians is appended in image raster order
per LoD level, i.e. already spatially coherent.  Stage times of the optimisation step for the same 1 M / 1080p cloud in
 (a) random order, (b) raster order of the creating view (what add_new_gaussians produces), (c) Morton order of the screen tile.
    python tools/lab/morton_lab.py [N W H]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd import fused, rasterizer
from harness import mapper

dev = torch.device("cuda:0")
N, W, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1_000_000, 1920, 1080)


def part1by1(v):
    v = v & 0xFFFF
    v = (v | (v << 8)) & 0x00FF00FF
    v = (v | (v << 4)) & 0x0F0F0F0F
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v


out = {}
for order in ("random", "raster", "morton_tile"):
    scene = mapper.build_synthetic_mapper(N, W, H, dev, seed=0, targets="render")
    if order != "random":
        xyz = scene.xyz.detach()
        fx = scene.f
        px = (fx * xyz[:, 0] / xyz[:, 2] + W / 2).clamp(0, W - 1)
        py = (fx * xyz[:, 1] / xyz[:, 2] + H / 2).clamp(0, H - 1)
        if order == "raster":
            key = py.long() * W + px.long()
        else:
            key = (part1by1((py.long() >> 4)) << 1) | part1by1((px.long() >> 4))
        perm = torch.argsort(key)
        with torch.no_grad():
            for k, pd in scene.gaussian_params.items():
                if k == "global_feat":
                    continue
                pd["val"].data = pd["val"].data[perm].contiguous()
        # keyframe targets were rendered before the permutation: a permutation does not change any render
    fused.patch_scene_model(scene)
    for i in range(8):
        scene.optimization_step(i % 4)
    t = rasterizer.StageTimer()
    rasterizer.set_stage_timer(t)
    for i in range(20):
        scene.optimization_step(i % 4)
    rasterizer.set_stage_timer(None)
    sm = t.summary_ms()
    out[order] = {k: round(v["mean_ms"], 4) for k, v in sm.items()}
    out[order]["sum"] = round(sum(v["mean_ms"] for v in sm.values()), 4)
    del scene
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
