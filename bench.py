#!/usr/bin/env python
"""bench.py -- mapper hot path on MI355X (see DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU, one independent scene per rank.  Under torch.distributed.run the ranks are already there;
    a bare `python bench.py --gpus N` re-executes itself under torch.distributed.run with N local ranks on 127.0.0.1.
    `--backend gloo --cpu-dry-run` drives the same launch / barrier / all-reduce / report path on CPU for the tests.)

Workload (BASELINE.json metric "on-the-fly frames/sec + raster fwd+bwd ms @1M Gaussians 1080p"):
config[2] -- 1 M Gaussians, 1920x1080, SH degree 3, RGB+D, L1 + fused-SSIM + inverse-depth loss,
sparse Adam -- as a seeded synthetic cloud.  One "step" = one mapper optimisation
step (render -> loss -> backward -> keyframe Adam -> sparse Gaussian Adam), exactly the body of
SceneModel.optimization_step (h3dgsv3.py:401-469) driven through the reference's binding surface.
frames/s = steps/s / steps_per_frame with steps_per_frame = 10 (run.sh --num_common_iterations 10).

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
STEPS_PER_FRAME = 10   # run.sh: --num_common_iterations 10 (key frames use 20)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage-detail", action="store_true", help="print per-stage timings to stderr")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the secondary measurements (other BASELINE configs, unfused glue) reported next to `value`")
    ap.add_argument("--unfused-glue", action="store_true",
                    help="time ARTDECO's render() glue as stock torch ops instead of artdeco_amd.fused (SURVEY 8 f-1)")
    ap.add_argument("--no-frontend", action="store_true",
                    help="skip the bounded frontend / whole-system measurements reported next to `value` (N = 1 only)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--cpu-dry-run", action="store_true",
                    help="no GPU work: a constant CPU step through the same launch / barrier / all-reduce / report code "
                         "(tests/test_multigpu.py runs `bench.py --gpus 2 --backend gloo --cpu-dry-run`)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    (one process per GPU, rendezvous on 127.0.0.1, a free port).  Never returns."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def cpu_dry_run(args, rank, world):
    """The N > 1 control path on CPU (gloo): rendezvous, barrier, K timed `steps`, MAX / SUM all-reduce, one JSON line."""
    from artdeco_amd import multigpu
    dev = torch.device("cpu")
    multigpu.init(args.backend, None)
    x = torch.ones(1 << 16)
    step = lambda: float((x * 1.0001).sum())
    for _ in range(args.warmup):
        step()
    multigpu.barrier(None)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    multigpu.barrier(None)
    elapsed = time.perf_counter() - t0
    elapsed_max, sums = multigpu.aggregate(elapsed, {"steps": float(args.steps)}, dev)
    if rank == 0:
        print(json.dumps({"metric": "cpu-dry-run steps/sec (control path only)", "value": sums["steps"] / elapsed_max, "unit": "steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "cpu-dry-run",
                          "config": {"workload": "constant CPU step (no kernels): launch + barrier + metric all-reduce path",
                                     "parallelism": f"scene-per-rank x{world}", "backend": args.backend}}))
    multigpu.shutdown()


def measure_hbm_copy_gbs(lib, dev):
    """Achievable HBM bandwidth of THIS box: adk_stream_copy (float4 grid-stride copy) of 1 GiB, read + write bytes over the
    best of 5 launches.  BASELINE.md asks for the roofline denominator to be re-measured next to the 8 TB/s spec."""
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=dev).fill_(1)
    dst = torch.empty_like(src)
    stream = torch.cuda.current_stream(dev).cuda_stream
    best = float("inf")
    for i in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.adk_stream_copy(dst.data_ptr(), src.data_ptr(), n, stream)
        e1.record()
        assert rc == 0, rc
        torch.cuda.synchronize(dev)
        if i >= 2:
            best = min(best, e0.elapsed_time(e1))
    return 2.0 * n / (best * 1e-3) / 1e9


def cpu_baseline(args):
    """The oracle ("port") timed on this box's host cores on a bounded sample of the same workload:
    a 1/36-area window (320x180, N/36 Gaussians => same splat density per pixel), one render +
    fp32 autograd backward + SSIM + Adam in torch-CPU (about 20 s on the GPU box's host).  Scaled by 36 to the full frame."""
    from oracle import gsplat_oracle as go
    from oracle import ssim_oracle
    DIV = 6
    W, H, N = args.width // DIV, args.height // DIV, args.gaussians // (DIV * DIV)
    torch.manual_seed(0)
    sc = go.synthetic_scene(N, W, H, seed=0)
    leaves = {k: sc[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    gt = torch.rand(3, H, W)
    t0 = time.time()
    r, a, _ = go.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                               sc["viewmat"], sc["K"], W, H, eps2d=0.01, grad_dtype=torch.float32)
    img = r[..., :3].permute(2, 0, 1)
    loss = 0.8 * (img - gt).abs().mean() + 0.2 * (1 - ssim_oracle.fused_ssim_oracle(img[None], gt[None]))
    loss.backward()
    with torch.no_grad():
        for v in leaves.values():  # dense torch Adam-style update as the CPU stand-in for adamUpdate
            g = v.grad
            m, s = 0.5 * g, 0.01 * g * g
            v -= 1e-3 * m / (s.sqrt() + 1e-15)
    dt = time.time() - t0
    step_s_full = dt * float(DIV * DIV)
    return {"value": 1.0 / (step_s_full * STEPS_PER_FRAME), "unit": "frames/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"oracle (torch-CPU) render+loss+backward+update of a 1/{DIV * DIV}-area window ({W}x{H}, {N} Gaussians, "
                      f"same density) = {dt:.1f} s, x{DIV * DIV} to the full frame, /{STEPS_PER_FRAME} steps per frame"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                     # one rank per GPU; this process is replaced
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks", file=sys.stderr)
        sys.exit(2)
    if args.cpu_dry_run:
        return cpu_dry_run(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from artdeco_amd import _lib, multigpu, rasterizer
    from harness import mapper
    multigpu.init(args.backend, dev)  # nccl = RCCL over xGMI; used for the barrier + the metric all-reduce only
    lib = _lib.load()
    torch.manual_seed(rank)
    scene = mapper.build_synthetic_mapper(args.gaussians, args.width, args.height, dev, seed=rank, targets="render")
    nkf = len(scene.keyframes)
    from artdeco_amd import fused
    glue = "torch (unchanged host code)"
    if not args.unfused_glue and fused.patch_scene_model(scene):
        glue = "artdeco_amd.fused (one HIP kernel per direction)"
    fused.freeze_gc()   # the host program's own choice (DESIGN finding 5b); the library never does it on its own

    def sync_all():
        multigpu.barrier(dev)

    for i in range(args.warmup):
        scene.optimization_step(i % nkf)
    # timed region: HIP events around the roofline kernel only (raster_bwd); every event pair costs a few
    # microseconds of stream bubble, so the full per-stage breakdown is taken in a second, untimed pass
    timer = rasterizer.StageTimer(only=("raster_bwd",))
    rasterizer.set_stage_timer(timer)
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        scene.optimization_step(i % nkf)
    sync_all()
    elapsed = time.perf_counter() - t0
    stages_timed = timer.summary_ms()
    detail = rasterizer.StageTimer()
    rasterizer.set_stage_timer(detail)
    for i in range(min(args.steps, 10)):
        scene.optimization_step(i % nkf)
    rasterizer.set_stage_timer(None)
    stages = detail.summary_ms()
    stages["raster_bwd"] = stages_timed["raster_bwd"]

    # workload size seen by the kernels (last step): I intersections, V visible, P pixels
    with torch.no_grad():
        pkg = scene.render_from_id(0)
    I, P = rasterizer.LAST_STATS["I"], args.width * args.height
    V = int(pkg["visibility_filter"].sum())

    elapsed_max, sums = multigpu.aggregate(elapsed, {"steps": float(args.steps)}, dev)  # MAX time, SUM steps
    total_steps = sums["steps"]

    if rank == 0:
        frames_per_s = total_steps / STEPS_PER_FRAME / elapsed_max
        bwd_ms = stages["raster_bwd"]["mean_ms"]
        alg_bytes_bwd = 44.0 * I + 28.0 * P + 40.0 * V  # raster bwd
        achieved = alg_bytes_bwd / (bwd_ms * 1e-3) / 1e9
        hbm_measured = measure_hbm_copy_gbs(lib, dev)
        prof = _profile_counters(args)
        out = {
            "metric": "on-the-fly frames/sec (mapper hot path; raster fwd+bwd + L1 + fused-SSIM + sparse Adam) @1M Gaussians 1080p",
            "value": frames_per_s, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: 1M-Gaussian map, 1920x1080 render, RGB+D SH3, L1+fused-SSIM+invdepth loss, sparse Adam; keyframes observe the cloud itself (stationary workload); one independent scene per GPU",
                       "gaussians": args.gaussians, "width": args.width, "height": args.height,
                       "steps_per_frame": STEPS_PER_FRAME, "intersections_I": I, "visible_V": V, "pixels_P": P,
                       "render_glue": glue, "parallelism": f"scene-per-gpu x{world}"},
            "raster_fwd_ms": stages["raster_fwd"]["mean_ms"], "raster_bwd_ms": bwd_ms,
            "stage_ms": {k: round(v["mean_ms"], 4) for k, v in stages.items()},
            "roofline": {"bound": "hbm", "kernel": "raster_bwd_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": prof.get("raster_bwd_hbm_bytes"),
                         "algorithmic_bytes": alg_bytes_bwd, "avg_launch_ms": bwd_ms,
                         "peak_measured_stream_copy": hbm_measured, "frac_of_measured_peak": achieved / hbm_measured,
                         "traffic_source": prof.get("source")},
        }
        if prof.get("raster_bwd_valu_wave_insts"):
            # What actually bounds the kernel the metric prices against HBM is vector-ALU time.  Two
            # numbers, both from the committed PMC pass of THIS kernel source (null when the source changed since):
            #  * issue fraction: wave64 VALU instructions per launch / launch time against the guide's issue rate, 1024 SIMDs
            #    x 2.4 GHz / 2 cycles per instruction (MI355X_MICROARCH.md "issues each VALU instruction over 2 cycles");
            #  * VALU-busy: SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (1024 SIMDs x launch cycles at 2.4 GHz).
            # busy / issue = average cycles per VALU instruction / 2: DPP, transcendental and packed ops cost more than 2.
            valu = prof["raster_bwd_valu_wave_insts"]
            peak_valu = 1024 * 2.4e9 / 2.0 / 1e9
            ach = valu / (bwd_ms * 1e-3) / 1e9
            out["roofline_valu"] = {"bound": "valu", "kernel": "raster_bwd_kernel", "achieved": ach, "peak": peak_valu,
                                    "unit": "G wave-instr/s", "frac": ach / peak_valu, "wave_insts_per_launch": valu}
            if prof.get("raster_bwd_active_inst_valu_quadcycles"):
                busy = prof["raster_bwd_active_inst_valu_quadcycles"] * 4.0 / (1024 * bwd_ms * 1e-3 * 2.4e9)
                out["roofline_valu"]["valu_busy"] = busy
                out["roofline_valu"]["cycles_per_valu_inst"] = prof["raster_bwd_active_inst_valu_quadcycles"] * 4.0 / valu
        if not args.no_extra_configs and world == 1:
            out["other_configs"] = extra_configs(args, dev)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        if not args.no_frontend and world == 1:
            del scene
            torch.cuda.empty_cache()
            out["frontend"] = frontend_summary(args, dev, cpu=not args.no_cpu_baseline)
            out["system"] = system_summary()
        if args.stage_detail:
            print(json.dumps(stages, indent=1), file=sys.stderr)
        print(json.dumps(out))
    multigpu.shutdown()


def frontend_summary(args, dev, cpu=True):
    """BASELINE configs[0]/[1], bounded: the MASt3R ViT-L 512x384 frontend on this GPU (TF32-class precision = the reference's
    allow_tf32 setting, see bench_frontend.py) -- one asymmetric pair match and one tracked frame, each as a hipGraph replay --
    next to the reference's PyTorch-CPU MASt3R path (fp32, model only, ONE pair) on this box's host cores."""
    try:
        import bench_frontend as BF
        from artdeco_amd.mast3r_model import vit_large
        torch.manual_seed(0)
        cpu_net = vit_large().eval()
        img1, img2 = torch.rand(1, 3, 384, 512) * 2 - 1, torch.rand(1, 3, 384, 512) * 2 - 1
        res = {"precision": "tf32eq: fp16 GEMM/conv operands (TF32's 10-bit mantissa), fp32 accumulate / residual stream / LayerNorm / softmax",
               "data": "synthetic images, random-init weights"}
        if cpu:
            with torch.inference_mode():
                t0 = time.perf_counter()
                cpu_net({"img": img1}, {"img": img2})
                cdt = time.perf_counter() - t0
            res["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": f"ONE fp32 pair inference (2 encodes + decoder + 2 heads, no matching kernels) of the same module on the host: {cdt:.2f} s"}
        net = cpu_net.to(dev).to_inference_dtype(torch.float16, fp32_stream=True, heads=True)
        g1, g2 = img1.to(dev), img2.to(dev)
        res["tracking_frame"] = BF.tracking_frame_bench(net, g1, g2, 20)
        for _ in range(3):
            BF.pair_match(net, g1, g2, None)
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            BF.pair_match(net, g1, g2, None)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(graph):
            BF.pair_match(net, g1, g2, None)
        graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            graph.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        res["pair_match"] = {"ms_per_pair": dt * 1e3, "pairs_per_s": 1.0 / dt,
                             "roofline": {"bound": "mfma", "achieved": BF.PAIR_TFLOP / dt, "peak": BF.MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "frac": BF.PAIR_TFLOP / dt / BF.MFMA_BF16_PEAK_TFLOPS}}
        del net, graph
        torch.cuda.empty_cache()
        return res
    except Exception as e:  # report, never hide
        return {"error": repr(e)[:300]}


def system_summary():
    """Mapper and frontend as two processes on this one GPU (bench_system.py, 120 tracked frames): min of the two rates."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_system.py"), "--frames", "120", "--alone-seconds", "2"],
                           capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        return json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-300:]}
    except Exception as e:
        return {"error": repr(e)[:300]}


def _time_steps(scene, steps=10, warmup=3, repeats=2):
    """Seconds per step: best of `repeats` runs of `steps` steps (a one-off allocator stall in a 10-step run otherwise
    shows up as a 10x slower configuration)."""
    nkf = len(scene.keyframes)
    for i in range(warmup):
        scene.optimization_step(i % nkf)
    best = float("inf")
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            scene.optimization_step(i % nkf)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    return best


def extra_configs(args, dev):
    """Secondary single-GPU measurements, OUTSIDE the timed region of `value` (10 steps each): the other
    BASELINE.json configs that fit one GPU, and the headline config with ARTDECO's render() glue left as stock
    torch ops (what an unchanged run_system.py gets without the one-line artdeco_amd.fused patch)."""
    from artdeco_amd import fused
    from harness import mapper
    res = {}
    cases = [("configs[1] 200k Gaussians 512x384", 200_000, 512, 384, True, False),
             ("north-star target 1M Gaussians 512x384", 1_000_000, 512, 384, True, False),
             ("run.sh training resolution (map 1296x972, pyr_lvl 1): 1M Gaussians 648x486", 1_000_000, 648, 486, True, False),
             ("configs[3] 4M Gaussians 2592x1944, d_max = creation depth x LoD level (LoD culling and fading active)",
              4_000_000, 2592, 1944, True, True),
             ("configs[3] 4M Gaussians 2592x1944, no LoD culling (d_max = inf)", 4_000_000, 2592, 1944, True, False),
             (f"headline config, UNCHANGED host code: ARTDECO's torch glue, natives swapped only ({args.gaussians} Gaussians {args.width}x{args.height})",
              args.gaussians, args.width, args.height, False, False),
             ("north-star target 1M Gaussians 512x384, UNCHANGED host code", 1_000_000, 512, 384, False, False)]
    for name, n, w, h, use_fused, lod in cases:
        try:
            scene = mapper.build_synthetic_mapper(n, w, h, dev, seed=0, targets="render", lod=lod)
            if use_fused:
                fused.patch_scene_model(scene)
            dt = _time_steps(scene)
            res[name] = {"ms_per_step": dt * 1e3, "frames_per_s": 1.0 / (dt * STEPS_PER_FRAME)}
            if lod:
                with torch.no_grad():
                    pkg = scene.render_from_id(0)
                res[name]["lod_selected_frac"] = float(scene.last_selected_frac) if hasattr(scene, "last_selected_frac") else None
                res[name]["visible_frac"] = float(pkg["visibility_filter"].float().mean())
            del scene
            torch.cuda.empty_cache()
        except Exception as e:  # report, never hide
            res[name] = {"error": repr(e)}
    return res


def kernel_source_sha():
    """Hash of the sources raster_bwd_kernel is compiled from: profiles/traffic.json is only valid for the kernel it profiled."""
    h = hashlib.sha256()
    for f in ("raster_tiles.hip", "adk_common.hpp"):
        h.update(open(os.path.join(ROOT, "artdeco_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def _profile_counters(args):
    """Per-launch PMC figures of raster_bwd_kernel from the committed rocprofv3 passes (profiles/traffic.json, written by
    tools/profile_round.sh on this exact workload).  Stamped with the hash of the kernel's sources: after any edit of
    raster_tiles.hip the old counters are NOT reported (traffic: null) until the profile is re-taken."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        return {}
    if (t.get("gaussians"), t.get("width"), t.get("height")) != (args.gaussians, args.width, args.height):
        return {}
    if t.get("kernel_source_sha") != kernel_source_sha():
        return {"source": f"profiles/traffic.json is stale (kernel sources changed since {t.get('tag')}): re-run tools/profile_round.sh"}
    t["source"] = f"profiles/traffic.json ({t.get('tag')}; rocprofv3 --pmc passes, per launch)"
    return t


if __name__ == "__main__":
    main()
