#!/usr/bin/env python
"""Per-step trace of the bench.py workload: GPU time between step boundaries (HIP events), host enqueue time per step, and
the raster_bwd kernel time of every step.  Answers "why does ms_per_step depend on --steps": a kernel time that creeps up is the
clock settling under sustained load, a flat kernel time with growing step time is host-side (allocator, GC).

    python tools/step_trace.py [--steps 300] [--gc-off]
"""
import argparse
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--gc-off", action="store_true")
    ap.add_argument("--profile", action="store_true", help="cProfile the host side of the steps instead of tracing them")
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    a = ap.parse_args()
    import artdeco_amd
    artdeco_amd.install_dropins()
    from artdeco_amd import fused, rasterizer
    from harness import mapper
    dev = torch.device("cuda:0")
    scene = mapper.build_synthetic_mapper(a.gaussians, a.width, a.height, dev, seed=0, targets="render")
    fused.patch_scene_model(scene)
    nkf = len(scene.keyframes)
    for i in range(10):
        scene.optimization_step(i % nkf)
    if a.gc_off:
        gc.disable()
    if a.profile:
        import cProfile
        import pstats
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for i in range(a.steps):
            scene.optimization_step(i % nkf)
        torch.cuda.synchronize()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(28)
        st.sort_stats("cumulative").print_stats(30)
        return
    timer = rasterizer.StageTimer(only=("raster_bwd",))
    rasterizer.set_stage_timer(timer)
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    host, isects, mem = [], [], []
    marks[0].record()
    t_all = time.perf_counter()
    for i in range(a.steps):
        t0 = time.perf_counter()
        scene.optimization_step(i % nkf)
        marks[i + 1].record()
        host.append((time.perf_counter() - t0) * 1e3)
        isects.append(rasterizer.LAST_STATS["I"])
        mem.append(torch.cuda.memory_reserved() >> 20)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t_all) * 1e3 / a.steps
    rasterizer.set_stage_timer(None)
    gpu = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    bwd = [e0.elapsed_time(e1) for e0, e1 in timer.events["raster_bwd"]]
    print(f"wall {wall:.3f} ms/step over {a.steps} steps; reserved {mem[0]} -> {mem[-1]} MiB; distinct I {len(set(isects))}")
    print("steps      gpu_mean gpu_max  host_mean host_max  bwd_mean  reserved")
    for s in range(0, a.steps, 20):
        e = min(s + 20, a.steps)
        g, h, b = gpu[s:e], host[s:e], bwd[s:e]
        print(f"{s:4d}-{e:4d}  {sum(g)/len(g):7.3f} {max(g):7.3f}  {sum(h)/len(h):8.3f} {max(h):8.3f}  {sum(b)/len(b):7.3f}  {mem[e-1]}")
    slow = sorted(range(a.steps), key=lambda i: -gpu[i])[:8]
    print("slowest steps (idx, gpu ms, host ms, I):", [(i, round(gpu[i], 2), round(host[i], 2), isects[i]) for i in sorted(slow)])


if __name__ == "__main__":
    main()
