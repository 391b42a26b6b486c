#!/usr/bin/env python
"""Eight ranks' HOST path without the eight GPUs ("host-side contention ... is the scaling risk, not xGMI").

    python tools/lab/host_ranks.py [--ranks 1,2,4,8] [--steps 400]

For every rank count R: R processes, each pinned to its own 1/R of the cores exactly as bench.py pins its ranks (`bench.pin_rank_cpus`),
each with its own scene on the ONE GPU of the box, each running the default optimisation step (`adk_mapper_step`) on a scene so small that
the host, not the device, bounds a lone process (2 000 Gaussians, 96x64: tools/lab/host_cost_step.py).  All ranks start their timed steps
at the same wall-clock instant.  Reported per rank count: wall ms per step, the part of it spent INSIDE the step's one device wait (adk_mapper_step reports it: out->wait_ns,
the host blocked on the intersection count while R processes' kernels share the one device), and the difference = the host's own work per
step (Python, ctypes, the HIP runtime's launch path), mean and worst rank.  Process CPU time is listed too but says little: the runtime
spins inside the wait.  HOST ms/step flat from 1 to 8 ranks = the eight host loops do not get in each other's way (driver locks, allocator,
core migration), which is the part of the 8-GPU run this box can measure; the wall time of R > 1 is the shared device.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(rank, world, steps, start_at, gaussians, width, height):
    import bench
    bench.pin_rank_cpus(rank, world)
    import torch
    import artdeco_amd
    artdeco_amd.install_dropins()
    from artdeco_amd import fused, native_step
    from harness import mapper
    dev = torch.device("cuda:0")
    torch.manual_seed(rank)
    scene = mapper.build_synthetic_mapper(gaussians, width, height, dev, seed=rank, targets="render")
    fused.patch_scene_model(scene)
    for i in range(60):
        scene.optimization_step(i % 4)
    torch.cuda.synchronize()
    late = time.time() > start_at
    while time.time() < start_at:
        pass
    c0, t0, w0 = time.process_time(), time.perf_counter(), native_step.STATS["wait_ns"]
    for i in range(steps):
        scene.optimization_step(i % 4)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    wall, cpu = time.perf_counter() - t0, time.process_time() - c0
    wait = (native_step.STATS["wait_ns"] - w0) * 1e-9
    print(json.dumps({"rank": rank, "world": world, "wall_ms_per_step": wall / steps * 1e3, "cpu_ms_per_step": cpu / steps * 1e3,
                      "wait_ms_per_step": wait / steps * 1e3, "host_ms_per_step": (t1 - t0 - wait) / steps * 1e3,
                      "cores": len(os.sched_getaffinity(0)), "native_steps": native_step.STATS["native"], "started_late": late}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--gaussians", type=int, default=2000)
    ap.add_argument("--width", type=int, default=96)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--child", nargs=3, type=float, default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.child is not None:
        return child(int(a.child[0]), int(a.child[1]), a.steps, a.child[2], a.gaussians, a.width, a.height)
    print(f"# scene per rank: {a.gaussians} Gaussians, {a.width}x{a.height}; {a.steps} timed steps per rank; {len(os.sched_getaffinity(0))} host cores")
    print("# ranks | wall ms/step mean (worst) | of which inside the step's one device wait | HOST ms/step = wall - wait, mean (worst) | cores per rank")
    table = []
    for R in [int(x) for x in a.ranks.split(",")]:
        start_at = time.time() + 20.0 + 3.0 * R          # imports + scene build + warm-up of R concurrent processes
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--steps", str(a.steps), "--gaussians", str(a.gaussians),
                                   "--width", str(a.width), "--height", str(a.height), "--child", str(r), str(R), repr(start_at)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for r in range(R)]
        rows = []
        for p in procs:
            out, _ = p.communicate(timeout=600)
            rows += [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]
        assert len(rows) == R, (R, rows)
        wall = [r["wall_ms_per_step"] for r in rows]
        wait = [r["wait_ms_per_step"] for r in rows]
        host = [r["host_ms_per_step"] for r in rows]
        late = any(r["started_late"] for r in rows)
        print(f"{R:7d} | {sum(wall) / R:.4f} ({max(wall):.4f}) | {sum(wait) / R:.4f} | {sum(host) / R:.4f} ({max(host):.4f}) | {rows[0]['cores']}" + ("  [a rank missed the common start]" if late else ""), flush=True)
        table.append({"ranks": R, "rows": rows})
    print(json.dumps(table))


if __name__ == "__main__":
    main()
