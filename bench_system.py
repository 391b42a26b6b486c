#!/usr/bin/env python
"""Mapper, frontend AND backend sharing one MI355X -- the thing "on-the-fly frames/sec" actually is (BASELINE metric; run.sh:17-20
puts --device_frontend, --device_backend and --device_mapper on the same GPU, run_system.py:99-110 starts them as separate processes).

Three processes on cuda:0, started together:
  mapper    run_system.py's frame loop (harness/stream.py: Keyframe build, rigid_transform_gs on SLAM keyframes, add_keyframe,
            add_new_gaussians on important frames, 20 / 10 optimisation steps) on 1 M Gaussians at 512x384 (the north-star
            configuration), fused host paths; rate = frames / wall (the reference's FPS definition, h3dgsv3.py:1129-1132);
  frontend  VSLAM/Frontend.py:55-124 per frame: one tracked frame = 1 MASt3R ViT-L encode (keyframe embedding cached) + decoder +
            2 heads + iter_proj + refine_matches as one hipGraph replay, then the Sim(3) tracker (adk_track_frame), TF32-class;
  backend   VSLAM/Backend.py:42-115 per frame with --use_all_frames: the SECOND mast3r_match_asymmetric of the frame against the last
            keyframe (style 2, :99-115: the same tracked-frame graph) and, on SLAM keyframes (every --slam-every frames, style 1 ->
            gloabla_optimization :196-265), a symmetric re-match (two more decoder + head passes) and gauss_newton_rays over the
            keyframe graph (16 keyframes, 512x384 points per factor, 10 iterations).
Each is first timed ALONE (the others idle at a barrier), then all three run as the PIPELINE they are in run_system.py: the frontend
free-runs over --frames frames (no --sync_hard), the backend takes frame k when the frontend has delivered it (states.msgFromFrontend,
polled with sleep(0.001) as Backend.py:60 does), the mapper takes frame k when the backend has (states.msgFromBackend, run_system.py:146-150).
The system rate is frames / wall from the common start until the MAPPER has finished the last frame -- what run_system.py prints.  (Until
round 3 the three ran unthrottled and the rate was the minimum of the three: a mapper that got faster then took GPU time from the backend
for frames nobody had delivered yet, and the minimum went DOWN.)  `--cu-mask A B`: HSA_CU_MASK for the frontend / backend processes
(e.g. "0:0-63"), the A/B of confining the two 768-token workloads to a slice of the chip.  Prints one JSON line.

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

ROUNDS = ("mapper_alone", "frontend_alone", "backend_alone", "together")


def _wait_for(counter, k, patience_s=300.0):
    """Block until the upstream stage has delivered frame k (polled like Backend.py:60 / run_system.py:148); a stage that died must not
    leave its consumers spinning on the GPU box for ever."""
    t0 = time.perf_counter()
    while counter.value <= k:
        time.sleep(0.001)
        if time.perf_counter() - t0 > patience_s:
            raise TimeoutError(f"no frame {k} after {patience_s:.0f} s: the upstream process is gone")


def mapper_proc(args, barrier, flow, out_q):
    import numpy as np
    import artdeco_amd
    artdeco_amd.install_dropins()
    from artdeco_amd import fused
    from harness import mapper, stream
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    scene = mapper.build_synthetic_mapper(args.gaussians, args.width, args.height, dev, seed=0, n_keyframes=0, targets="random")
    fused.patch_scene_model(scene)
    fused.freeze_gc()
    cadence = dict(kf_every=5, slam_every=args.slam_every, test_hold=8)
    frames = stream.synthetic_frames(scene, 48, seed=0, texture=0.05)       # recycled: the loop only reads them
    np.random.seed(0)
    stream.warm_process(dev)
    stream.run_stream(scene, frames[:16], start_index=0, **cadence)   # incl. the first SLAM keyframe
    idx = 16
    for phase in ROUNDS:
        barrier.wait()
        if phase in ("mapper_alone", "together"):
            n, t0 = 0, time.perf_counter()
            while (time.perf_counter() - t0 < args.alone_seconds) if phase == "mapper_alone" else (n < args.frames):
                if phase == "together":
                    _wait_for(flow["backend"], n)
                stream.run_stream(scene, [frames[idx % len(frames)]], start_index=idx, **cadence)   # synchronises once per frame
                idx += 1
                n += 1
            out_q.put(("mapper", phase, n, time.perf_counter() - t0))
        barrier.wait()


def _frontend_graph(dev, precision="tf32eq"):
    import bench_frontend as BF
    from artdeco_amd.mast3r_model import vit_large
    torch.manual_seed(0)
    net = vit_large().to(dev).eval()
    if precision != "fp32":   # fp32: every GEMM / convolution operand stays fp32 (no TF32 MFMA on gfx950: the strict reading of the reference's arithmetic)
        net = net.to_inference_dtype(torch.float16, fp32_stream=True, heads=True)
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
    return net, graph, (img_f, kf_feat, kf_pos)


def frontend_proc(args, barrier, flow, out_q):
    if args.cu_mask and args.cu_mask[0] != "-":
        os.environ["HSA_CU_MASK"] = args.cu_mask[0]
    import artdeco_amd
    artdeco_amd.install_dropins()
    import bench_frontend as BF
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    net, graph, _ = _frontend_graph(dev, args.frontend_precision)
    track = BF.make_tracker_step(dev)
    for _ in range(3):
        graph.replay(); track()
    torch.cuda.synchronize()
    for phase in ROUNDS:
        barrier.wait()
        if phase in ("frontend_alone", "together"):
            n = args.frames if phase == "together" else max(args.frames // 3, 50)
            t0 = time.perf_counter()
            for k in range(n):
                graph.replay()
                track()                      # includes the tracker's one host read per frame: the frame is finished when it returns
                if phase == "together":
                    flow["frontend"].value = k + 1
            torch.cuda.synchronize()
            out_q.put(("frontend", phase, n, time.perf_counter() - t0))
        barrier.wait()


def backend_proc(args, barrier, flow, out_q):
    if args.cu_mask and args.cu_mask[1] != "-":
        os.environ["HSA_CU_MASK"] = args.cu_mask[1]
    import numpy as np
    import artdeco_amd
    artdeco_amd.install_dropins()
    import bench_frontend as BF
    import mast3r_slam_backends as B
    from artdeco_amd import synthetic as S
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    net, graph, (img_f, kf_feat, kf_pos) = _frontend_graph(dev, args.frontend_precision)
    g = S.keyframe_graph(num_poses=16, n=512 * 384, seed=0, extra_edges=12, coherent=True)
    T0 = S.perturb_poses(g["T_gt"], np.random.default_rng(1), 0.01)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    Xs, Cs, ii, jj, idx, valid, Q = t(g["Xs"]), t(g["Cs"]), t(g["ii"]), t(g["jj"]), t(g["idx"]), t(g["valid"]), t(g["Q"])
    T0d = t(T0)

    def slam_keyframe():
        # gloabla_optimization: symmetric re-match against the previous keyframe (both embeddings cached: two decoder + head +
        # matching passes), then the global Gauss-Newton over the keyframe graph
        with torch.inference_mode():
            for _ in range(2):
                BF.tracking_frame_match(net, img_f, kf_feat, kf_pos)
        B.gauss_newton_rays(T0d.clone(), Xs, Cs, ii, jj, idx, valid, Q, 0.003, 10.0, 0.0, 1.5, 10, 1e-8)

    for _ in range(2):
        graph.replay(); slam_keyframe()
    torch.cuda.synchronize()
    for phase in ROUNDS:
        barrier.wait()
        if phase in ("backend_alone", "together"):
            n, t0 = 0, time.perf_counter()
            limit = max(args.frames // 3, 50)
            while n < (limit if phase == "backend_alone" else args.frames):
                if phase == "together":
                    _wait_for(flow["frontend"], n)
                graph.replay()                                   # style 2: the frame's second asymmetric match
                if n % args.slam_every == 0:
                    slam_keyframe()                              # style 1
                torch.cuda.synchronize()                         # the result goes to the mapper through a host queue (Backend.py:147)
                n += 1
                if phase == "together":
                    flow["backend"].value = n
            out_q.put(("backend", phase, n, time.perf_counter() - t0))
        barrier.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--alone-seconds", type=float, default=3.0)
    ap.add_argument("--slam-every", type=int, default=15)
    ap.add_argument("--cu-mask", nargs=2, default=None, metavar=("FRONTEND", "BACKEND"),
                    help='HSA_CU_MASK for the frontend and backend processes, e.g. "0:0-63" "0:64-127"; "-" leaves one unmasked')
    ap.add_argument("--frontend-precision", default="tf32eq", choices=["tf32eq", "fp32"],
                    help="tf32eq: fp16 GEMM / conv operands, fp32 everything else (what the reference's allow_tf32 amounts to, the default); "
                         "fp32: strict fp32 in the frontend and backend networks")
    args = ap.parse_args()
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(4), ctx.Queue()
    flow = {"frontend": ctx.Value("i", 0), "backend": ctx.Value("i", 0)}   # frames delivered so far
    procs = [ctx.Process(target=f, args=(args, barrier, flow, q), daemon=True) for f in (mapper_proc, frontend_proc, backend_proc)]   # die with the parent
    for p in procs:
        p.start()
    res = {}
    for phase in ROUNDS:
        barrier.wait()
        for _ in range(3 if phase == "together" else 1):
            who, ph, n, dt = q.get(timeout=1800)
            res[(who, ph)] = (n, dt)
        barrier.wait()
    for p in procs:
        p.join(timeout=120)
    rate = lambda who, ph: res[(who, ph)][0] / res[(who, ph)][1]
    tog = {w: rate(w, "together") for w in ("mapper", "frontend", "backend")}
    alone = {w: rate(w, w + "_alone") for w in ("mapper", "frontend", "backend")}
    finish = {w: res[(w, "together")][1] for w in tog}
    out = {"metric": "on-the-fly frames/sec with frontend -> backend -> mapper running as a pipeline of three processes on one MI355X "
                     "(frames / wall until the mapper has finished the last frame)",
           "value": args.frames / max(finish.values()), "unit": "frames/s", "n_gpus": 1, "data": "synthetic, random-init MASt3R weights",
           "config": {"workload": f"{args.frames} tracked frames; mapper: run_system.py's frame loop on {args.gaussians} Gaussians {args.width}x{args.height} "
                                  f"(20 / 10 iterations, add_new_gaussians on important frames); frontend: MASt3R ViT-L 512x384 tracked frame ({'TF32-class' if args.frontend_precision == 'tf32eq' else 'strict fp32'}) + "
                                  f"Sim(3) tracker; backend: the frame's second asymmetric match + on every {args.slam_every}th frame a symmetric re-match and "
                                  "gauss_newton_rays over a 16-keyframe graph; three processes, same device",
                      "cu_mask": args.cu_mask, "frontend_precision": args.frontend_precision},
           "pipeline_finish_s": finish, "pipeline_frames_per_s": tog, "alone_frames_per_s": alone,
           "gpu_ms_per_frame_alone_sum": sum(1e3 / v for v in alone.values())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
