#!/usr/bin/env python
"""What the natives under ONE `add_new_gaussians` call cost (h3dgsv3.py:766-953, every important frame): 4 x render_from_id (one per
LoD level, :790), 4 x update_voxel (:884), 1 x optimizer.add_and_prune with the new Gaussians (:938), 1 x weed_out_gaussians (:940)
-- with the fused paths installed (what the drop-ins' post-import hook does) and without (drop-in natives under ARTDECO's torch
glue / torch chains).  The image-space torch ops in between (bilinear resizes, two 2-D convolutions per level, grid_sample) are
ARTDECO's own in both cases and not timed here.

    python tools/bench_important_frame.py [N W H new_points keyframes]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd import fused
from bench_update_voxel import torch_chain
from harness import mapper


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    a = [int(x) for x in sys.argv[1:]]
    N, W, H, E, NKF = (a + [1_000_000, 1296, 972, 30_000, 40][len(a):])[:5]
    dev = torch.device("cuda:0")
    res = {"config": {"gaussians": N, "width": W, "height": H, "new_points_per_level": E, "keyframes": NKF}}
    for mode in ("fused", "unfused"):
        sc = mapper.build_synthetic_mapper(N, W, H, dev, seed=0, targets="render", lod=True)
        g = torch.Generator().manual_seed(1)
        for k in range(NKF - len(sc.keyframes)):
            Rt = torch.eye(4); Rt[:3, 3] = 0.3 * torch.randn(3, generator=g)
            sc.add_keyframe(type(sc.keyframes[0])(sc.keyframes[0].image_pyr[0], sc.keyframes[0].idepth_pyr[0], Rt.to(dev), dev))
        if mode == "fused":
            assert fused.patch_scene_model(sc)
        new_xyz = (sc.xyz.detach()[torch.randint(0, N, (E,), generator=g).to(dev)] + 0.05 * torch.randn(E, 3, generator=g).to(dev)).contiguous()
        r = {}
        with torch.no_grad():
            r["render_from_id_x4"] = 4 * timed(lambda: sc.render_from_id(0))
            if mode == "fused":
                r["update_voxel_x4"] = 4 * timed(lambda: fused.update_voxel_device(new_xyz, sc.xyz.detach(), sc.cls_id, 0.1))
            else:
                r["update_voxel_x4"] = 4 * timed(lambda: torch_chain(new_xyz, sc.xyz.detach(), sc.cls_id, 0.1))

            def weed():
                # weed_out with a threshold nobody fails (visible_threshold = 0 and every Gaussian is in range of its creating view):
                # the full count + mask + add_and_prune path runs, the scene stays the same size for the next repetition
                sc.weed_out_gaussians()
            r["weed_out_gaussians"] = timed(weed, reps=3, warm=1)
            n0 = sc.xyz.shape[0]

            def add():
                P = sc.gaussian_params
                ext = {k: P[k]["val"].detach()[:E].clone() for k in ("cls_id", "d_max", "xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "local_feat")}
                ext["global_feat"] = P["global_feat"]["val"].detach()[:0]
                keep = torch.ones(sc.xyz.shape[0], dtype=torch.bool, device=dev)
                keep[-E:] = sc.xyz.shape[0] == n0       # drop what the previous repetition appended
                sc.optimizer.add_and_prune(ext, keep)
            r["add_and_prune"] = timed(add, reps=3, warm=1)
        r["total_ms"] = sum(r.values())
        res[mode] = r
        del sc
        torch.cuda.empty_cache()
    res["speedup"] = res["unfused"]["total_ms"] / res["fused"]["total_ms"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
