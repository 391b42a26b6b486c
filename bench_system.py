#!/usr/bin/env python
"""Mapper and frontend SHARING one MI355X -- the thing "on-the-fly frames/sec" actually is (BASELINE metric; run.sh:17-20 puts
--device_frontend, --device_backend and --device_mapper on the same GPU, run_system.py:99-110 starts them as separate processes).

Two processes on cuda:0, started together:
  mapper    the bench.py step (1 M Gaussians, 512x384 -- the north-star target configuration -- fused glue), 10 optimisation
            steps per frame (run.sh --num_common_iterations 10);
  frontend  one tracked frame per iteration: 1 MASt3R ViT-L encode (keyframe embedding cached) + decoder + 2 heads + iter_proj +
            refine_matches as one hipGraph replay, then the Sim(3) tracker (adk_track_frame), TF32-class precision
            (bench_frontend.py --dtype tf32eq).
Each is first timed ALONE (the other process idles at a barrier), then both run concurrently for the same number of frontend frames (default 300, BASELINE configs[1]);
the system rate is min(frontend frames/s, mapper frames/s) under contention.  Prints one JSON line.

    python bench_system.py [--frames 300] [--gaussians 1000000] [--width 512] [--height 384]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.multiprocessing as mp

STEPS_PER_FRAME = 10
ROUNDS = ("mapper_alone", "frontend_alone", "together")


def mapper_proc(args, barrier, stop, out_q):
    import artdeco_amd
    artdeco_amd.install_dropins()
    from artdeco_amd import fused
    from harness import mapper
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    scene = mapper.build_synthetic_mapper(args.gaussians, args.width, args.height, dev, seed=0, targets="render")
    fused.patch_scene_model(scene)
    fused.freeze_gc()
    nkf = len(scene.keyframes)
    for i in range(10):
        scene.optimization_step(i % nkf)
    torch.cuda.synchronize()
    for phase in ROUNDS:
        barrier.wait()
        if phase != "frontend_alone":
            steps, t0 = 0, time.perf_counter()
            while (time.perf_counter() - t0 < args.alone_seconds) if phase == "mapper_alone" else (not stop.is_set()):
                scene.optimization_step(steps % nkf)
                steps += 1
                if steps % 20 == 0:
                    torch.cuda.synchronize()   # keep the launch queue bounded, as the real loop's per-frame host work does
            torch.cuda.synchronize()
            out_q.put(("mapper", phase, steps, time.perf_counter() - t0))
        barrier.wait()


def frontend_proc(args, barrier, stop, out_q):
    import artdeco_amd
    artdeco_amd.install_dropins()
    import bench_frontend as BF
    from artdeco_amd.mast3r_model import vit_large
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    net = vit_large().to(dev).eval().to_inference_dtype(torch.float16, fp32_stream=True, heads=True)
    img_f = torch.rand(1, 3, 384, 512, device=dev) * 2 - 1
    img_k = torch.rand(1, 3, 384, 512, device=dev) * 2 - 1
    with torch.inference_mode():
        kf_feat, kf_pos, _ = net._encode_image(img_k, torch.tensor(img_k.shape[-2:])[None])
    for _ in range(3):
        BF.tracking_frame_match(net, img_f, kf_feat, kf_pos)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        BF.tracking_frame_match(net, img_f, kf_feat, kf_pos)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        BF.tracking_frame_match(net, img_f, kf_feat, kf_pos)
    track = BF.make_tracker_step(dev)
    for _ in range(3):
        graph.replay(); track()
    torch.cuda.synchronize()
    for phase in ROUNDS:
        barrier.wait()
        if phase != "mapper_alone":
            n = args.frames if phase == "together" else max(args.frames // 3, 50)
            t0 = time.perf_counter()
            for _ in range(n):
                graph.replay()
                track()                      # includes the tracker's one host read per frame
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if phase == "together":
                stop.set()
            out_q.put(("frontend", phase, n, dt))
        barrier.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--alone-seconds", type=float, default=3.0)
    args = ap.parse_args()
    ctx = mp.get_context("spawn")
    barrier, stop, q = ctx.Barrier(3), ctx.Event(), ctx.Queue()
    procs = [ctx.Process(target=mapper_proc, args=(args, barrier, stop, q)), ctx.Process(target=frontend_proc, args=(args, barrier, stop, q))]
    for p in procs:
        p.start()
    res = {}
    for phase in ROUNDS:
        barrier.wait()
        for _ in range(2 if phase == "together" else 1):
            who, ph, n, dt = q.get(timeout=1200)
            res[(who, ph)] = (n, dt)
        barrier.wait()
    for p in procs:
        p.join(timeout=120)
    m_n, m_dt = res[("mapper", "together")]
    f_n, f_dt = res[("frontend", "together")]
    mapper_fps = m_n / STEPS_PER_FRAME / m_dt
    frontend_fps = f_n / f_dt
    out = {"metric": "on-the-fly frames/sec with mapper and frontend sharing one MI355X (min of the two rates under contention)",
           "value": min(mapper_fps, frontend_fps), "unit": "frames/s", "n_gpus": 1, "data": "synthetic, random-init MASt3R weights",
           "config": {"workload": f"{args.frames} tracked frames; mapper {args.gaussians} Gaussians {args.width}x{args.height}, {STEPS_PER_FRAME} steps/frame; "
                                  "frontend MASt3R ViT-L 512x384 tracked frame (TF32-class) + Sim(3) tracker; two processes, same device"},
           "together": {"mapper_frames_per_s": mapper_fps, "mapper_ms_per_step": m_dt / m_n * 1e3, "frontend_frames_per_s": frontend_fps,
                        "frontend_ms_per_frame": f_dt / f_n * 1e3},
           "alone": {"mapper_frames_per_s": res[("mapper", "mapper_alone")][0] / STEPS_PER_FRAME / res[("mapper", "mapper_alone")][1],
                     "frontend_frames_per_s": res[("frontend", "frontend_alone")][0] / res[("frontend", "frontend_alone")][1]}}
    out["slowdown_under_contention"] = {"mapper": out["alone"]["mapper_frames_per_s"] / mapper_fps,
                                        "frontend": out["alone"]["frontend_frames_per_s"] / frontend_fps}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
