#!/usr/bin/env python
"""bench.py -- mapper hot path on MI355X (see DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU, one independent scene per rank.  Under torch.distributed.run the ranks are already there;
    a bare `python bench.py --gpus N` re-executes itself under torch.distributed.run with N local ranks on 127.0.0.1.
    `--backend gloo --cpu-dry-run` drives the same launch / barrier / all-reduce / report path on CPU for the tests.)

Workload (BASELINE.json metric "on-the-fly frames/sec + raster fwd+bwd ms @1M Gaussians 1080p"):
configs[2] -- a 1 M-Gaussian map rendered at 1920x1080, SH degree 3, RGB+D, L1 + fused-SSIM + inverse-depth loss, sparse Adam,
densification with LoG multi-resolution initialisation -- as a seeded synthetic cloud and a synthetic frame stream.

One "step" = ONE MAPPED FRAME = one pass of the mapper loop body of run_system.py:143-234 (harness/stream.py): Keyframe
construction (pyramids), [SLAM keyframes: pose re-read + rigid_transform_gs], add_keyframe, [important frames: add_new_gaussians
= 4 LoD levels of probability maps + render + sampling + update_voxel, add_and_prune, weed_out_gaussians], then 20 (important) or
10 optimisation steps (run.sh --num_key_iterations 20 --num_common_iterations 10), each = SceneModel.optimization_step
(h3dgsv3.py:401-469: render -> loss -> backward -> keyframe Adam -> sparse Gaussian Adam).  `value` = frames / wall-seconds of
that loop: the reference's own FPS definition (h3dgsv3.py:1129-1132).  Frame cadence (no dataset here; stated in the output):
mapper keyframe every 5th frame, SLAM keyframe every 15th, test frame every 8th (run.sh --test_hold 8), --use_all_frames.

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

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
STEPS_PER_FRAME = 10   # run.sh: --num_common_iterations 10 (key frames use 20); only for the secondary steps-only figures


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed FRAMES (one step = one mapped frame)")
    ap.add_argument("--warmup", type=int, default=8, help="untimed frames before them")
    ap.add_argument("--sequence-frames", type=int, default=1000, help="length of the sequence the timed window is taken from (BASELINE configs[2]: 1 000 frames)")
    ap.add_argument("--start-frame", type=int, default=-1,
                    help="sequence position of the first WARM-UP frame; the keyframes of the frames before it are built untimed (Keyframe objects + "
                         "add_keyframe, no optimisation).  Default -1: the timed window is centred on the MIDDLE of the sequence, where a cost that "
                         "grows linearly with the position (the reference script's SLAM-keyframe loop) equals its whole-sequence mean.  0: a fresh sequence")
    ap.add_argument("--texture", type=float, default=0.05, help="amplitude of what the synthetic frames show and the map does not explain yet "
                                                                 "(drives the number of Gaussians add_new_gaussians creates)")
    ap.add_argument("--kf-every", type=int, default=5)
    ap.add_argument("--slam-every", type=int, default=15)
    ap.add_argument("--test-hold", type=int, default=8)
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
    ap.add_argument("--psnr-large", action="store_true",
                    help="also run the LARGE PSNR proxy in the untimed tail (harness/psnr_proxy.run_large: an 80 k-Gaussian scene reconstructed from streamed "
                         "frames with add_new_gaussians + add_and_prune in the loop, HIP vs CPU oracle, 5 checkpoints; ~10 minutes, most of it the CPU oracle)")
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


def pin_rank_cpus(local_rank, local_world):
    """One scene per GPU means one Python host loop per GPU (plus, in the full system, a frontend and a backend process each): give
    every rank its own equal share of the node's cores so that eight host loops do not migrate over each other (the
    host, not xGMI, is the scaling risk).  A no-op for a single rank or where the affinity call is unavailable."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = len(cpus) // local_world
        if per >= 1:
            os.sched_setaffinity(0, cpus[local_rank * per:(local_rank + 1) * per])
    except OSError:
        pass


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


def cpu_baseline(args, steps_per_frame=STEPS_PER_FRAME):
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
    samples = []
    for _ in range(3):   # three samples of the same bounded workload; the median is reported, all three are listed
        for v in leaves.values():
            v.grad = None
        t0 = time.time()
        r, a, _ = go.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                                   sc["viewmat"], sc["K"], W, H, eps2d=0.01, grad_dtype=torch.float32)
        img = r[..., :3].permute(2, 0, 1)
        loss = 0.8 * (img - gt).abs().mean() + 0.2 * (1 - ssim_oracle.fused_ssim_oracle(img[None], gt[None]))
        loss.backward()
        with torch.no_grad():
            for v in leaves.values():  # dense torch Adam-style update as the CPU stand-in for adamUpdate
                g = v.grad
                m, s_ = 0.5 * g, 0.01 * g * g
                v -= 1e-3 * m / (s_.sqrt() + 1e-15)
        samples.append(time.time() - t0)
    dt = sorted(samples)[1]
    step_s_full = dt * float(DIV * DIV)
    return {"value": 1.0 / (step_s_full * steps_per_frame), "unit": "frames/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"oracle (torch-CPU) render+loss+backward+update of a 1/{DIV * DIV}-area window ({W}x{H}, {N} Gaussians, "
                      f"same density): median of 3 samples {dt:.1f} s ({', '.join(f'{x:.1f}' for x in samples)}), x{DIV * DIV} to the full frame, x{steps_per_frame:.1f} optimisation steps per frame of "
                      f"the timed stream (the per-frame stages other than the steps are not in the CPU figure)"}


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
    pin_rank_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    # ARTDECO_BENCH_SHARE_GPU=1 (development only, with --backend gloo): ranks beyond the visible GPUs share them, so that the N > 1
    # code path can be driven on a one-GPU box; the driver's multi-GPU runs never set it
    dev_index = local_rank % torch.cuda.device_count() if os.environ.get("ARTDECO_BENCH_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    from artdeco_amd import _lib, multigpu, rasterizer
    from harness import mapper
    multigpu.init(args.backend, dev)  # nccl = RCCL over xGMI; used for the barrier + the metric all-reduce only
    lib = _lib.load()
    torch.manual_seed(rank)
    np.random.seed(rank)
    from artdeco_amd import fused
    from harness import stream
    scene = mapper.build_synthetic_mapper(args.gaussians, args.width, args.height, dev, seed=rank, n_keyframes=0, targets="random")
    glue = "torch (unchanged host code)"
    if not args.unfused_glue and fused.patch_scene_model(scene):
        glue = "artdeco_amd.fused (HIP kernels behind render / optimization_step / add_new_gaussians / add_and_prune / update_voxel)"
    # `value` is measured with the interpreter as an unmodified run_system.py leaves it: NO gc.freeze() (the library never freezes on its
    # own; DESIGN finding 5b prices what a generation-2 collection costs).  `with_gc_freeze` below times the next
    # frames of the same sequence after fused.freeze_gc(), the host program's one-line opt-in (ARTDECO_AMD_GC_FREEZE=1 does the same).
    cadence = dict(kf_every=args.kf_every, slam_every=args.slam_every, test_hold=args.test_hold)
    n_detail = 10 if world == 1 else 0
    n_frozen = args.steps if world == 1 else 0
    n_torch_inv = args.steps if world == 1 and not args.unfused_glue else 0
    frames = stream.synthetic_frames(scene, args.warmup + args.steps + n_frozen + n_torch_inv + n_detail, seed=rank, texture=args.texture)  # resident in HBM
    # Where in the sequence the timed window sits (round 6).  run_system.py's frame cost grows with the sequence position (its SLAM-keyframe
    # loop walks every keyframe so far, :194-227), so frames 5-25 of a fresh sequence -- what rounds 3-5 timed -- are its CHEAPEST 20 frames and
    # overstate the sequence's frames/s by a third.  Default now: the window is centred on the sequence's middle frame, with the keyframes of
    # every earlier frame present (built untimed by stream.fast_forward: Keyframe construction + add_keyframe, the map keeps its size).
    F0 = args.start_frame if args.start_frame >= 0 else max(0, args.sequence_frames // 2 - args.warmup - args.steps // 2)

    def sync_all():
        multigpu.barrier(dev)

    stream.warm_process(dev, use_fused=not args.unfused_glue)   # one-off costs of the process, not of a frame (harness/stream.py)
    if F0 > 0:
        stream.fast_forward(scene, frames, F0, start_index=0, **cadence)
    stream.run_stream(scene, frames[:args.warmup], start_index=F0, **cadence)
    # timed region: HIP events around the roofline kernel only (raster_bwd); every event pair costs a few
    # microseconds of stream bubble, so the full per-stage breakdown is taken in a second, untimed pass
    timer = rasterizer.StageTimer(only=("raster_bwd",))
    rasterizer.set_stage_timer(timer)
    from artdeco_amd import native_step
    sync_all()
    native_before = dict(native_step.STATS)
    t0 = time.perf_counter()
    timed = stream.run_stream(scene, frames[args.warmup:args.warmup + args.steps], start_index=F0 + args.warmup, **cadence)
    sync_all()
    elapsed = time.perf_counter() - t0
    native_stats_timed = {k: native_step.STATS[k] - native_before[k] for k in native_before}
    native_stats_timed["one_call_step_enabled"] = native_step.enabled() and not args.unfused_glue
    native_stats_timed["rank"] = rank   # this rank's own steps (every rank runs its own scene)
    stages_timed = timer.summary_ms()
    rasterizer.set_stage_timer(None)
    frame_stages, stages = {}, {}
    frozen = None
    if n_frozen:
        fused.freeze_gc()
        k0 = args.warmup + args.steps
        frozen = stream.run_stream(scene, frames[k0:k0 + n_frozen], start_index=F0 + k0, **cadence)
        fused.unfreeze_gc()   # everything reported after this runs with the interpreter's default again
    torch_inv = None
    if n_torch_inv:
        # the next frames with torch's own 4x4 inverse back in place (ARTDECO_AMD_FAST_INV4=0): what the wrapped inverse is worth to run_system.py's
        # SLAM-keyframe loop (a LATER window of the sequence: more keyframes in that loop than `value`'s window had)
        from artdeco_amd import small_inverse
        was = small_inverse.installed()
        small_inverse.uninstall()
        k0 = args.warmup + args.steps + n_frozen
        torch_inv = stream.run_stream(scene, frames[k0:k0 + n_torch_inv], start_index=F0 + k0, **cadence)
        if was:
            small_inverse.install(force=True)
    if n_detail:
        # untimed: the loop's stages bracketed by device synchronisations, then the kernels of the optimisation step by HIP events
        k0 = args.warmup + args.steps + n_frozen + n_torch_inv
        frame_stages = stream.run_stream(scene, frames[k0:], start_index=F0 + k0, breakdown=True,
                                         **cadence)["stage_ms"]
        detail = rasterizer.StageTimer()
        rasterizer.set_stage_timer(detail)
        for i in range(10):
            stream.optimization_step(scene, True)
        rasterizer.set_stage_timer(None)
        stages = detail.summary_ms()
    stages["raster_bwd"] = stages_timed["raster_bwd"]
    if "project_fwd" in stages and "lod_params_fwd" not in stages:
        # the one-call step runs the LoD / mlp_cov forward and the projection forward as ONE kernel (timed as its projection stage)
        stages["lod_project_fwd"] = stages.pop("project_fwd")

    # workload size seen by the kernels (a render of the newest keyframe): N Gaussians, I intersections, V visible, P pixels
    with torch.no_grad():
        pkg = scene.render_from_id(-1)
    N_end = int(scene.xyz.shape[0])
    I, P = rasterizer.LAST_STATS["I"], args.width * args.height
    V = int(pkg["visibility_filter"].sum())

    elapsed_max, sums = multigpu.aggregate(elapsed, {"frames": float(args.steps), "steps": float(timed["steps"])}, dev)  # MAX time, SUMs
    total_frames = sums["frames"]

    if rank == 0:
        frames_per_s = total_frames / elapsed_max
        bwd_ms = stages["raster_bwd"]["mean_ms"]
        alg_bytes_bwd = 44.0 * I + 28.0 * P + 40.0 * V  # raster bwd
        achieved = alg_bytes_bwd / (bwd_ms * 1e-3) / 1e9
        hbm_measured = measure_hbm_copy_gbs(lib, dev)
        prof = _profile_counters(args)
        flags = [stream.frame_flags(i, **cadence) for i in range(F0 + args.warmup, F0 + args.warmup + args.steps)]
        out = {
            "metric": "on-the-fly frames/sec of the mapper loop (reference definition: frames / wall-seconds, run_system.py:139-276, h3dgsv3.py:1129-1132) @1M Gaussians 1080p",
            "value": frames_per_s, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: 1M-Gaussian map, 1920x1080 render, RGB+D SH3, L1+fused-SSIM+invdepth loss, sparse Adam, "
                                   "simple densify + LoG multi-res init; one step = one mapped FRAME of run_system.py's loop (Keyframe build, "
                                   "rigid_transform_gs on SLAM keyframes, add_keyframe, add_new_gaussians on important frames, 20 / 10 "
                                   "optimisation steps); frames observe the map itself plus unexplained texture; one independent scene per GPU",
                       "gaussians_start": int(timed["gaussians_start"]), "gaussians_end": int(timed["gaussians_end"]),
                       "width": args.width, "height": args.height, "pyr_levels": 1,
                       "keyframes_at_start": F0 + args.warmup,
                       "sequence_position": (f"frames {F0 + args.warmup}-{F0 + args.warmup + args.steps} of a {args.sequence_frames}-frame sequence: the window is centred on "
                                             f"the sequence's middle frame, with the {F0} earlier frames' keyframes present (built untimed: Keyframe + add_keyframe, no "
                                             "optimisation).  The reference script's per-keyframe SLAM pose loop (run_system.py:194-227) grows linearly with the position, "
                                             "so the middle of the sequence costs what the whole sequence costs on average: `full_sequence` runs BASELINE's 300- and "
                                             "1 000-frame sequences start to finish for comparison (`value_over_full_sequence`)") if F0 > 0 else
                                            (f"frames {args.warmup}-{args.warmup + args.steps} of a FRESH sequence (--start-frame 0): its cheapest frames, see `full_sequence`"),
                       "python_gc": "default (no gc.freeze()): what an unmodified run_system.py gets; see `with_gc_freeze`",
                       "cadence": {**cadence, "use_all_frames": True, "num_key_iterations": 20, "num_common_iterations": 10},
                       "important_frame_fraction": sum(f["is_important"] for f in flags) / len(flags),
                       "densified_frames": sum(f["is_important"] and not f["is_test"] for f in flags),
                       "slam_keyframes": sum(f["is_slam_keyframe"] for f in flags),
                       "optimisation_steps": int(sums["steps"]), "steps_per_frame": sums["steps"] / total_frames,
                       "new_gaussians_per_densified_frame": timed["gaussians_added"] / max(timed["densified_frames"], 1),
                       "gaussians_pruned_in_timed_region": int(timed["gaussians_start"] + timed["gaussians_added"] - timed["gaussians_end"]),
                       "intersections_I": I, "visible_V": V, "pixels_P": P, "gaussians_N": N_end,
                       "render_glue": glue, "parallelism": f"scene-per-gpu x{world}",
                       # how the optimisation steps of the timed region were issued: adk_mapper_step (forward + loss + backward as ONE native call,
                       # DESIGN finding 36) vs steps handed back to the per-stage chain (another layout / a frame that needs the global binning route)
                       "optimisation_step_host_path": dict(native_stats_timed)},
            "ms_per_optimisation_step_incl_frame_overheads": elapsed_max / max(timed["steps"], 1) * 1e3,
            "raster_fwd_ms": stages.get("raster_fwd", {}).get("mean_ms"), "raster_bwd_ms": bwd_ms,
            "frame_stage_ms": {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in frame_stages.items()},
            "stage_ms": {k: round(v["mean_ms"], 4) for k, v in stages.items()},
            "roofline": {"bound": "hbm", "kernel": bwd_kernel_name(args.width, args.height), "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": prof.get("raster_bwd_hbm_bytes"),
                         "algorithmic_bytes": alg_bytes_bwd, "avg_launch_ms": bwd_ms,
                         "peak_measured_stream_copy": hbm_measured, "frac_of_measured_peak": achieved / hbm_measured,
                         "traffic_source": prof.get("source")},
            "roofline_stages": roofline_stages(stages, N_end, V, I, P, args.width, args.height),
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
            out["roofline_valu"] = {"bound": "valu", "kernel": bwd_kernel_name(args.width, args.height), "achieved": ach, "peak": peak_valu,
                                    "unit": "G wave-instr/s", "frac": ach / peak_valu, "wave_insts_per_launch": valu}
            if prof.get("raster_bwd_active_inst_valu_quadcycles"):
                busy = prof["raster_bwd_active_inst_valu_quadcycles"] * 4.0 / (1024 * bwd_ms * 1e-3 * 2.4e9)
                out["roofline_valu"]["valu_busy"] = busy
                out["roofline_valu"]["cycles_per_valu_inst"] = prof["raster_bwd_active_inst_valu_quadcycles"] * 4.0 / valu
                # round 6 (FINDINGS 49, 52): SQ_ACTIVE_INST_VALU is 1.00 - 1.08 x SQ_INSTS_VALU in quad-cycles on EVERY kernel of this library, stream_copy included -- it ticks per
                # ISSUED instruction, so "valu_busy" ~ 1 does not show a saturated port.  Measured issue costs (tools/valu_cost_bench2.hip): 2.6
                # cycles for VGPR-only mul / add / fma, 4.2 for compares / selects / min / DPP / SGPR operands, 8.2 for v_exp / v_rcp / permlane
                # swaps; this kernel's mix issues in ~0.42 ms at full occupancy and the rest is latency at 7 waves per SIMD.
                out["roofline_valu"]["note"] = ("valu_busy / cycles_per_valu_inst are derived from SQ_ACTIVE_INST_VALU, which counts issued instructions "
                                                "(1.00 - 1.08 x SQ_INSTS_VALU on every kernel, the HBM-bound ones included), not port-busy time: see profiles/FINDINGS.md 49 and 52 for the measured "
                                                "per-class issue costs and the occupancy / padding probes")
        if frozen is not None:
            out["with_gc_freeze"] = {"frames_per_s": frozen["frames"] / frozen["seconds"], "frames": frozen["frames"],
                                     "sequence_position": f"frames {args.warmup + args.steps}-{args.warmup + args.steps + n_frozen} of the same sequence",
                                     "note": "after fused.freeze_gc() (the host program's opt-in, ARTDECO_AMD_GC_FREEZE=1): no generation-2 "
                                             "collection walks the process's long-lived objects between optimisation steps"}
        if torch_inv is not None:
            k0 = args.warmup + args.steps + n_frozen
            out["with_torch_inverse"] = {"frames_per_s": torch_inv["frames"] / torch_inv["seconds"], "frames": torch_inv["frames"],
                                         "sequence_position": f"frames {k0}-{k0 + n_torch_inv} of the same sequence",
                                         "note": "ARTDECO_AMD_FAST_INV4=0: torch.linalg.inv / Tensor.inverse as torch ships them (batched LU + a blocking "
                                                 "read of `info` per call; run_system.py:221-223 calls three per mapper keyframe on a SLAM keyframe) "
                                                 "instead of artdeco_amd.small_inverse (one launch, no read-back), default GC"}
        del scene, frames
        torch.cuda.empty_cache()
        if not args.no_extra_configs and world == 1:
            out["full_sequence"] = full_sequence(args, dev)
            fs = [v for k, v in out["full_sequence"].items() if k.startswith("configs[2]") and "BATCHED" not in k and "frames_per_s" in v]
            if fs:   # how representative the timed window is: `value` / the same configuration run start to finish with the unmodified host script
                out["value_over_full_sequence"] = out["value"] / fs[0]["frames_per_s"]
            out["other_configs"] = extra_configs(args, dev)
        if world == 1:
            out["frame_delivery"] = frame_delivery(args, dev, elapsed_max / args.steps * 1e3)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, timed["steps"] / args.steps)
            out["psnr_proxy"] = psnr_proxy(dev)
            if args.psnr_large:
                try:
                    from harness import psnr_proxy as PP
                    out["psnr_proxy_large"] = PP.run_large(dev, cpu_threads=32)
                except Exception as e:  # report, never hide
                    out["psnr_proxy_large"] = {"error": repr(e)[:300]}
            else:   # the last measured large proxy travels with the line (it takes ten minutes: not in the default run)
                lp = os.path.join(ROOT, "profiles", "r06_psnr_proxy_large.json")
                if os.path.exists(lp):
                    with open(lp) as f:
                        d = json.load(f)
                    out["psnr_proxy_large_last_measured"] = {"file": "profiles/r06_psnr_proxy_large.json", "max_abs_delta_db": d.get("max_abs_delta_db"),
                                                             "checkpoints": [{k: cp[k] for k in ("step", "cpu_db", "hip_db", "delta_db")} for cp in d.get("checkpoints", [])],
                                                             "gaussians_true": d.get("gaussians_true"), "steps": d.get("steps"),
                                                             "how": "python bench.py --psnr-large (or python harness/psnr_proxy.py --large)"}
        if not args.no_frontend and world == 1:
            out["frontend"] = frontend_summary(args, dev, cpu=not args.no_cpu_baseline)
            out["system"] = system_summary()
            sf = system_summary("fp32", frames=60)
            out["system_strict_fp32_frontend"] = {k: sf.get(k) for k in ("value", "unit", "pipeline_frames_per_s", "alone_frames_per_s", "error") if k in sf}
        if args.stage_detail:
            print(json.dumps(stages, indent=1), file=sys.stderr)
        print(json.dumps(out))
    multigpu.shutdown()


def bwd_kernel_name(W, H):
    """The form of raster_bwd_kernel a frame of this size is served by (raster_tiles.hip:split_parts, same thresholds)."""
    t = ((W + 15) // 16) * ((H + 15) // 16)
    if t < 1600:
        return "raster_bwd_kernel<1, 1, true> (one wave per 8x8 quadrant of a 16x16 list tile)"
    if t < 20000:
        return "raster_bwd_kernel<2, 1, true> (one wave per 16x8 half of a 16x16 list tile: the form a 1080p frame uses since round 4)"
    return "raster_bwd_kernel<2, 2, false> (one wave per 16x16 tile)"


def roofline_stages(stages, N, V, I, P, W, H):
    """Per-stage HBM roofline of one optimisation step from SURVEY.md 8(d)'s algorithmic bytes (N Gaussians, V visible, I tile
    intersections, P pixels, T 16x16 tiles) and the HIP-event time of the stage: {stage: {alg_bytes, ms, GBps, frac}} + the whole
    step.  Stages SURVEY gives no formula for carry the bytes DESIGN.md section 2 states for them."""
    T = ((W + 15) // 16) * ((H + 15) // 16)
    alg = {
        "project_fwd": 76.0 * N + 216.0 * V,                        # projection fwd + SH fwd (one kernel)
        "binning": 36.0 * I + 4.0 * T,                              # emit + sort minimum + offsets
        "raster_fwd": 44.0 * I + 24.0 * P,
        "raster_bwd": 44.0 * I + 28.0 * P + 40.0 * V,
        # projection bwd + SH bwd + the colours' Adam in ONE kernel.  SURVEY's SH-bwd term 420 V = 12 (direction) + 204 (192 B of
        # coefficients + 12 B of v_rgb read) + 192 (colour gradients WRITTEN) + 12 (v_dir); the fused colour Adam never writes or
        # re-reads those 192 B of gradients and reads the coefficients once for both purposes, so what the kernel has to move per visible
        # Gaussian is 36 B + the Adam triple p, m, v read and written = 36 + 6 x 192 = 1188 B (round 3 priced 420 + 1152 = 1572 B here,
        # which counted the removed traffic; counters say 1.24 GB per launch at 1 M / 1080p against 1.20 GB by this formula)
        "project_bwd": 148.0 * N + (36.0 + 6.0 * 192.0) * V,
        "ssim_fwd": 24.0 * P * 3, "ssim_bwd": 28.0 * P * 3,
        "adam_multi": 28.0 * 27 * V + N,                            # the remaining 27 floats per visible Gaussian + the mask
        "lod_params_fwd": 190.0 * N, "photometric_fwd": 64.0 * P, "photometric_bwd": 72.0 * P,
        # LoD forward + projection forward as one kernel: the two formulas minus the 32 B per Gaussian (opacity, scale, quaternion) the
        # projection no longer reads back
        "lod_project_fwd": 190.0 * N + 76.0 * N + 216.0 * V - 32.0 * N,
    }
    ms = {k: v["mean_ms"] for k, v in stages.items()}
    if all(k in ms for k in ("bin_count", "bin_scatter", "bin_sort")):
        ms["binning"] = ms["bin_count"] + ms["bin_scatter"] + ms["bin_sort"]
    out = {}
    for k, b in alg.items():
        if k in ms and ms[k] > 0:
            gbs = b / (ms[k] * 1e-3) / 1e9
            out[k] = {"alg_bytes": b, "ms": round(ms[k], 4), "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    counted = [k for k in out]
    tot_b = sum(alg[k] for k in counted)
    tot_ms = sum(v["mean_ms"] for k, v in stages.items() if k not in ("binning",))
    if tot_ms > 0 and "raster_fwd" in ms:   # N > 1 runs time raster_bwd only: no whole-step line from one stage
        out["whole_step"] = {"alg_bytes": tot_b, "ms_sum_of_stages": round(tot_ms, 4), "GBps": round(tot_b / (tot_ms * 1e-3) / 1e9, 1),
                             "frac": round(tot_b / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "note": "algorithmic bytes of the stages listed here / the sum of ALL stage times of one optimisation step"}
    return out


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


def system_summary(precision="tf32eq", frames=120):
    """Frontend -> backend -> mapper as a pipeline of three processes on this one GPU (bench_system.py, 120 frames): frames / wall until the
    mapper has finished the last one.  precision "fp32": the same pipeline with strict-fp32 MASt3R networks (a shorter run: it puts the
    precision dependence of the >= 30 frames/s figure on the record)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_system.py"), "--frames", str(frames), "--alone-seconds", "2",
                            "--frontend-precision", precision], capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        return json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-300:]}
    except Exception as e:
        return {"error": repr(e)[:300]}


def _time_steps(scene, steps=10, warmup=3, repeats=2):
    """Seconds per optimisation step on a stationary scene: best of `repeats` runs of `steps` steps (a one-off allocator stall in
    a 10-step run otherwise shows up as a 10x slower configuration)."""
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


def _stream_fps(n, w, h, dev, use_fused, lod, args, pyr_levels=1, warm=6, timed=16, slam_hw=(384, 512)):
    """frames/s of the mapper loop (the reference's FPS definition) on a fresh scene of this configuration: `timed` frames after
    `warm`, the headline's cadence.  (w, h) = the MAP resolution; training renders at (w, h) / 2^(pyr_levels - 1)."""
    from artdeco_amd import fused
    from harness import mapper, stream
    scene = mapper.build_synthetic_mapper(n, w, h, dev, seed=0, n_keyframes=0, targets="random", lod=lod)
    if use_fused:
        fused.patch_scene_model(scene)
    cadence = dict(kf_every=args.kf_every, slam_every=args.slam_every, test_hold=args.test_hold)
    frames = stream.synthetic_frames(scene, warm + timed, seed=0, texture=args.texture, slam_hw=slam_hw)
    np.random.seed(0)
    stream.warm_process(dev, use_fused=use_fused)
    stream.run_stream(scene, frames[:warm], start_index=0, pyr_levels=pyr_levels, **cadence)
    r = stream.run_stream(scene, frames[warm:], start_index=warm, pyr_levels=pyr_levels, **cadence)
    res = {"frames_per_s": r["frames"] / r["seconds"], "ms_per_frame": r["seconds"] / r["frames"] * 1e3, "frames": r["frames"],
           "optimisation_steps": r["steps"], "important_frames": r["important_frames"], "densified_frames": r["densified_frames"],
           "gaussians_start": r["gaussians_start"], "gaussians_end": r["gaussians_end"], "gaussians_added": r["gaussians_added"]}
    del scene, frames
    torch.cuda.empty_cache()
    return res


def _full_sequence(n, w, h, dev, args, n_frames, batched, distinct=48):
    """The reference's FPS over a WHOLE sequence (run_system.py:139,276 + h3dgsv3.py:1129-1132: len(dataset) / wall-seconds of the mapper
    loop), frame 0 to frame n_frames - 1 of a fresh scene, nothing fast-forwarded: every Keyframe build, every SLAM-keyframe pose re-read
    over ALL keyframes so far (run_system.py:194-227, linear in the sequence position), every densification, 20 / 10 optimisation steps per
    frame.  The synthetic stream cycles through `distinct` pre-rendered frames (resident in HBM; building 1 000 distinct ones would take longer
    than the run).  batched = the SLAM-keyframe loop as ONE batched call (artdeco_amd/keyframe_poses.py, the INTEGRATION section 3c edit)."""
    from artdeco_amd import fused
    from harness import mapper, stream
    scene = mapper.build_synthetic_mapper(n, w, h, dev, seed=0, n_keyframes=0, targets="random")
    fused.patch_scene_model(scene)
    cadence = dict(kf_every=args.kf_every, slam_every=args.slam_every, test_hold=args.test_hold)
    base = stream.synthetic_frames(scene, min(distinct, n_frames), seed=0, texture=args.texture)
    frames = [base[j % len(base)] for j in range(n_frames)]
    np.random.seed(0)
    stream.warm_process(dev)
    r = stream.run_stream(scene, frames, start_index=0, batched_slam_update=batched, marks_every=20, **cadence)
    marks = [0.0] + r["marks"]
    per_100 = [round(100 / (marks[i + 5] - marks[i]), 2) for i in range(0, len(marks) - 5, 5)]
    res = {"frames": r["frames"], "seconds": r["seconds"], "frames_per_s": r["frames"] / r["seconds"],
           "frames_per_s_per_100_frames": per_100, "frames_per_s_last_20_frames": 20 / (marks[-1] - marks[-2]),
           "optimisation_steps": r["steps"], "important_frames": r["important_frames"], "densified_frames": r["densified_frames"],
           "gaussians_start": r["gaussians_start"], "gaussians_end": r["gaussians_end"], "gaussians_added": r["gaussians_added"],
           "keyframes_at_end": len(scene.keyframes), "python_gc": "default (no gc.freeze())" if not fused._GC_FROZEN else "frozen earlier in this process",
           "slam_pose_update": "batched (requires the INTEGRATION section 3c edit of run_system.py)" if batched else "run_system.py:194-227 as written (per-keyframe loop; the new pose of each keyframe, which the script reads from the pypose SLAM graph with five LieTensor calls, is a clone + a small shift here: pypose is the script's own dependency and costs it more host time than this stand-in)"}
    del scene, frames, base
    torch.cuda.empty_cache()
    return res


def full_sequence(args, dev):
    """`value` times frames 8-48 of a fresh sequence; BASELINE's configs name 300 (PINGPONG, configs[1]) and 1 000 frames (configs[2]).  Here
    both sequences run START TO FINISH, once with the reference script's per-keyframe SLAM loop as written and once with it batched."""
    res = {}
    cases = [("configs[1] / north-star: PINGPONG's 300 frames, 1M Gaussians 512x384", 1_000_000, 512, 384, 300),
             (f"configs[2]: 1 000 frames, {args.gaussians} Gaussians {args.width}x{args.height}", args.gaussians, args.width, args.height, 1000)]
    for name, n, w, h, nf in cases:
        for batched in (False, True):
            key = name + (" -- SLAM-keyframe loop BATCHED (requires the INTEGRATION 3c edit)" if batched else "")
            try:
                res[key] = _full_sequence(n, w, h, dev, args, nf, batched)
            except Exception as e:  # report, never hide
                res[key] = {"error": repr(e)[:300]}
    return res


def frame_delivery(args, dev, ms_per_frame):
    """What `value` leaves out by contract (inputs resident in HBM): in run_system.py every mapped frame arrives from the backend process
    through a host queue and is copied to the mapper's device inside the timed loop (run_system.py:162-177: `densePoint.to(device_mapper)`,
    `dataset.transform.to_map(original_img, device=device_mapper)`).  Measured here: the frame's image (3 x H x W fp32) and dense point map
    + confidence at the SLAM resolution (384 x 512 x 4 fp32) from PINNED host memory to the device, mean of 10 copies; and the frame rate
    with that copy serialised in front of every frame (the PCIe-inclusive rate: never `value`)."""
    try:
        img = torch.empty(3, args.height, args.width, dtype=torch.float32).pin_memory()
        pts = torch.empty(384, 512, 4, dtype=torch.float32).pin_memory()
        for _ in range(2):
            img.to(dev, non_blocking=True); pts.to(dev, non_blocking=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(10):
            a = img.to(dev, non_blocking=True); b = pts.to(dev, non_blocking=True)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / 10 * 1e3
        nbytes = img.numel() * 4 + pts.numel() * 4
        del a, b
        return {"bytes_per_frame": nbytes, "h2d_ms_per_frame": ms, "h2d_GBps": nbytes / (ms * 1e-3) / 1e9,
                "frames_per_s_with_delivery_serialised": 1e3 / (ms_per_frame + ms),
                "note": "pinned host -> device copy of one frame's image + point map + confidence; inside the reference's wall clock "
                        "(run_system.py:162-177), outside `value` (inputs resident in HBM by the bench contract)"}
    except Exception as e:  # report, never hide
        return {"error": repr(e)[:300]}


def psnr_proxy(dev):
    """BASELINE's metric has a third component, "PSNR delta" (north star: within 0.1 dB of the reference).  No dataset and no CUDA
    reference exist on the GPU box; the bounded proxy (harness/psnr_proxy.py) trains the same small reconstruction from the same state
    with the HIP path and with the CPU-ORACLE path (the oracle as the checker, as in cpu_baseline) and reports the held-out PSNR of
    both at three checkpoints, the largest |delta| and the proxy's own noise floor (CPU vs CPU from positions scaled by 1 + 1e-7)."""
    try:
        from harness import psnr_proxy as PP
        out = PP.run(dev, steps=30, every=10, noise_floor=True, cpu_threads=8)
        out["reference"] = "PSNR = 10 log10(1 / mse) (Reconstruct/utils.py:86-87) on held-out views rendered like SceneModel.evaluate (h3dgsv3.py:523-558)"
        out["criterion"] = "max_abs_delta_db <= 0.1 (north star: PSNR within 0.1 dB)"
        return out
    except Exception as e:  # report, never hide
        return {"error": repr(e)[:300]}


def extra_configs(args, dev):
    """Secondary single-GPU measurements, OUTSIDE the timed region of `value`: the other BASELINE.json configs that fit one GPU,
    each as (a) the mapper FRAME loop (`frames_per_s`, the reference's FPS definition, same cadence as the headline) and (b) the
    bare optimisation step on a stationary scene (`ms_per_step`; `steps_only_frames_per_s` = steps/s / 10 is the UPPER BOUND
    earlier rounds reported as frames/s: no Keyframe build, no important frame); and the headline config with ARTDECO's own torch
    host code around the natives (what an unchanged run_system.py gets with ARTDECO_AMD_AUTOFUSE=0)."""
    from artdeco_amd import fused
    from harness import mapper
    res = {}
    # name, N, map W, map H, fused, lod, pyr_levels, stream?
    cases = [(f"headline cloud in RASTER order (the order add_new_gaussians appends in; SURVEY 8d's cloud is in random order): {args.gaussians} Gaussians "
              f"{args.width}x{args.height}, optimisation step only", args.gaussians, args.width, args.height, True, "raster", 1, False),
             ("configs[1] 200k Gaussians 512x384", 200_000, 512, 384, True, False, 1, True),
             ("north-star target 1M Gaussians 512x384", 1_000_000, 512, 384, True, False, 1, True),
             ("run.sh geometry: 1M Gaussians, map 1296x972, --pyr_levels 2 (training renders 648x486, densification renders 1296x972)",
              1_000_000, 1296, 972, True, False, 2, True),
             ("configs[3] 4M Gaussians 2592x1944, d_max = creation depth x LoD level (LoD culling and fading active)",
              4_000_000, 2592, 1944, True, True, 1, True),
             ("configs[3] 4M Gaussians 2592x1944, no LoD culling (d_max = inf)", 4_000_000, 2592, 1944, True, False, 1, False),
             (f"headline config, UNCHANGED host code: ARTDECO's torch glue, natives swapped only ({args.gaussians} Gaussians {args.width}x{args.height})",
              args.gaussians, args.width, args.height, False, False, 1, True),
             ("north-star target 1M Gaussians 512x384, UNCHANGED host code", 1_000_000, 512, 384, False, False, 1, True)]
    # A frame on the other side of what used to be the 8 192-entries-per-tile cliff (round 5): the north-star size with splats twice as wide
    # (sigma ~ 4 px: I / N ~ 8, every tile list 10-20 k entries).  Until round 4 such a frame left the tile-local sort AND the one-call step.
    try:
        from artdeco_amd import native_step as _ns
        scene = mapper.build_synthetic_mapper(1_000_000, 512, 384, dev, seed=0, targets="render", sigma_px=4.0)
        fused.patch_scene_model(scene)
        before = dict(_ns.STATS)
        dt = _time_steps(scene)
        d = {k: _ns.STATS[k] - before[k] for k in ("native", "fallback_route", "fallback_layout", "long_list_steps")}
        from artdeco_amd import rasterizer as _r
        res["north-star size, DENSE frame: 1M Gaussians 512x384 with sigma ~ 4 px (tile lists of 10-20 k entries: the long-list sort inside the one-call step)"] = {
            "ms_per_step": dt * 1e3, "steps_only_frames_per_s": 1.0 / (dt * STEPS_PER_FRAME), "intersections_I": _r.LAST_STATS.get("I"),
            "steps_native": d["native"], "steps_with_a_long_list": d["long_list_steps"], "fallback_route": d["fallback_route"],
            "fallback_layout": d["fallback_layout"]}
        # the way round 4 ran such a frame: the per-stage chain with the global radix route (adk_mapper_step returned ADK_STEP_EROUTE)
        saved = {k: os.environ.get(k) for k in ("ARTDECO_AMD_NATIVE_STEP", "ADK_BIN_LONG")}   # a user's own setting survives this block
        os.environ["ARTDECO_AMD_NATIVE_STEP"], os.environ["ADK_BIN_LONG"] = "0", "0"
        try:
            res[next(reversed(res))]["ms_per_step_round4_path (per-stage chain, global radix route)"] = _time_steps(scene) * 1e3
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        del scene
        torch.cuda.empty_cache()
    except Exception as e:  # report, never hide
        res["north-star size, DENSE frame"] = {"error": repr(e)[:300]}
    for name, n, w, h, use_fused, lod, pyr, do_stream in cases:
        try:
            tw, th = w >> (pyr - 1), h >> (pyr - 1)
            raster = lod == "raster"
            lod = bool(lod) and not raster
            scene = mapper.build_synthetic_mapper(n, tw, th, dev, seed=0, targets="render", lod=lod, order="raster" if raster else "random")
            if use_fused:
                fused.patch_scene_model(scene)
            dt = _time_steps(scene)
            res[name] = {"ms_per_step": dt * 1e3, "steps_only_frames_per_s": 1.0 / (dt * STEPS_PER_FRAME)}
            if lod:
                with torch.no_grad():
                    pkg = scene.render_from_id(0)
                res[name]["visible_frac"] = float(pkg["visibility_filter"].float().mean())
            del scene
            torch.cuda.empty_cache()
            if do_stream:
                big = n >= 4_000_000 or not use_fused
                res[name].update(_stream_fps(n, w, h, dev, use_fused, lod, args, pyr_levels=pyr, warm=3 if big else 6, timed=8 if big else 16))
        except Exception as e:  # report, never hide
            res.setdefault(name, {})["error"] = repr(e)[:300]
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
