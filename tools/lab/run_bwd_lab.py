#!/usr/bin/env python
"""DEV TOOL: build the raster-backward lab variants (tools/lab/raster_bwd_lab.hip, -DLAB_VARIANT=n [extra flags]) and time
them on the bench workload (1 M Gaussians, 1920x1080, stationary cloud).  `--build-only` cross-compiles in the CPU
container (the .so files travel to the GPU box); without it the script times what is built.

    python tools/lab/run_bwd_lab.py --build-only
    gpurun -- python tools/lab/run_bwd_lab.py
"""
import ctypes
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
BUILD = os.path.join(HERE, "_build")
SRC = os.path.join(HERE, "raster_bwd_lab.hip")
# name -> (LAB_VARIANT, extra hipcc flags)
NS = ["-fno-slp-vectorize"]
VARIANTS = {
    "v0_product": (0, NS),
    "v1_no_atomics": (1, NS),
    "v2_no_reduction": (2, NS),
    "v3_cull_exp_only": (3, NS),
    "v4_lds_transpose": (4, NS),
    "v5_lds_table_dsadd": (5, NS),
    "v6_lds_table_swap": (6, NS),
    "v7_v6_exp2_body": (7, NS),
    "v8_v7_6waves": (8, NS),
    "v9_count_atomics": (1, NS + ["-DLAB_COUNT"]),
    "v10_v0_exp2_body": (0, NS + ["-DLAB_EXP2"]),
}


def build():
    os.makedirs(BUILD, exist_ok=True)
    procs = []
    for name, (v, extra) in VARIANTS.items():
        out = os.path.join(BUILD, f"lab_{name}.so")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-DLAB_VARIANT={v}", *extra,
               "-I" + os.path.join(ROOT, "artdeco_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), SRC, "-o", out]
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for name, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise SystemExit(f"{name}: build failed\n{out}")
    print("built", len(VARIANTS), "variants in", BUILD)


def main():
    if "--build-only" in sys.argv:
        return build()
    import torch
    import artdeco_amd
    artdeco_amd.install_dropins()
    from artdeco_amd import _lib, rasterizer
    from harness import mapper
    dev = torch.device("cuda:0")
    N, W, H = 1_000_000, 1920, 1080
    if len(sys.argv) >= 4 and sys.argv[1].isdigit():
        N, W, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    if "--bench-workload" in sys.argv:
        # the tensors raster_bwd sees inside a real mapper step (stationary bench scene, step 12): captured from the
        # autograd node instead of synthesised
        from artdeco_amd import fused
        scene = mapper.build_synthetic_mapper(N, W, H, dev, seed=0, targets="render")
        fused.patch_scene_model(scene)
        cap = {}
        orig = rasterizer.RasterizeGaussians.backward

        def spy(ctx, v_colors, v_alphas, *rest):
            sv = ctx.saved_tensors
            cap.update(rec=sv[8].clone(), flat=sv[10].clone(), offs=sv[11].clone(), final_T=sv[12].clone(), last_ids=sv[13].clone(),
                       vr=v_colors.contiguous().clone(), va=v_alphas.contiguous().clone())
            return orig(ctx, v_colors, v_alphas, *rest)
        rasterizer.RasterizeGaussians.backward = staticmethod(spy)
        for i in range(12):
            scene.optimization_step(i % len(scene.keyframes))
        torch.cuda.synchronize()
        rec, flat, offs, final_T, last_ids, vr, va = (cap[k] for k in ("rec", "flat", "offs", "final_T", "last_ids", "vr", "va"))
        N = rec.shape[0]
        if "--random-v" in sys.argv:
            g = torch.Generator(device=dev).manual_seed(0)
            vr = torch.randn(H, W, 4, device=dev, generator=g).contiguous()
            va = torch.randn(H, W, 1, device=dev, generator=g).contiguous()
        if "--scaled-v" in sys.argv:   # the bench's gradients at the magnitude of the synthetic ones
            vr = (vr / vr.abs().mean()).contiguous()
            va = (va / va.abs().mean().clamp_min(1e-30)).contiguous()
    else:
        c = mapper.synthetic_cloud(N, W, H, 0)
        t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in c.items()}
        K = torch.tensor([[c["fx"], 0, W / 2], [0, c["fx"], H / 2], [0, 0, 1.0]], device=dev)
        with torch.no_grad():
            out = rasterizer.render_camera(t["means"], t["quats"], t["scales"], t["opacities"], t["sh"], torch.eye(4, device=dev), K,
                                           W, H, sh_degree=3, eps2d=0.01)
        rc, ra, radii, rec, tpg, flat, offs, isect_ids, last_ids, main_ids, final_T = out
        g = torch.Generator(device=dev).manual_seed(0)
        vr = torch.randn(H, W, 4, device=dev, generator=g).contiguous()
        va = torch.randn(H, W, 1, device=dev, generator=g).contiguous()
    sink = torch.zeros(((W + 15) // 16) * ((H + 15) // 16) * 64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    res, ref = {}, None
    print(f"N={N} {W}x{H} I={flat.numel()} workload={'bench step' if '--bench-workload' in sys.argv else 'synthetic, random v_render'}")
    P = ctypes.c_void_p
    libdir = os.path.join(ROOT, "artdeco_amd", "lib")
    products = {f"product[{f[len('libartdeco_hip.'):-3] or 'current'}]": os.path.join(libdir, f) for f in sorted(os.listdir(libdir))
                if f.startswith("libartdeco_hip") and f.endswith(".so")}
    for name in list(products) + list(VARIANTS):
        is_product = name in products
        so = products[name] if is_product else os.path.join(BUILD, f"lab_{name}.so")
        if not os.path.exists(so):
            continue
        lib = ctypes.CDLL(so)
        if is_product:
            lib.adk_raster_bwd.restype = ctypes.c_int
            lib.adk_raster_bwd.argtypes = [ctypes.c_int, ctypes.c_int, P, P, P, ctypes.c_int64, P, P, P, P, P, P, P]
            lib.lab_variant = lambda: 0
        else:
            lib.lab_raster_bwd.restype = ctypes.c_int
            lib.lab_raster_bwd.argtypes = [ctypes.c_int, ctypes.c_int, P, P, P, ctypes.c_int64, P, P, P, P, P, P, P, P]

        def run(v_rec):
            if is_product:
                rc_ = lib.adk_raster_bwd(W, H, rec.data_ptr(), flat.data_ptr(), offs.data_ptr(), flat.numel(), None, final_T.data_ptr(),
                                         last_ids.data_ptr(), vr.data_ptr(), va.data_ptr(), v_rec.data_ptr(), stream)
            else:
                rc_ = lib.lab_raster_bwd(W, H, rec.data_ptr(), flat.data_ptr(), offs.data_ptr(), flat.numel(), None, final_T.data_ptr(),
                                         last_ids.data_ptr(), vr.data_ptr(), va.data_ptr(), v_rec.data_ptr(), sink.data_ptr(), stream)
            assert rc_ == 0, rc_
        v_rec = torch.zeros(N, 12, device=dev)
        for _ in range(3):
            run(v_rec)
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            v_rec.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(v_rec); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        res[name] = {"median_ms": ts[len(ts) // 2], "min_ms": ts[0]}
        if is_product or lib.lab_variant() in (0, 4, 5, 6, 7, 8, 9):
            if ref is None:
                ref = v_rec.clone()
            else:
                res[name]["rel_diff_vs_v0"] = float((v_rec - ref).norm() / ref.norm())
        if name == "v9_count_atomics":
            sk = sink.view(-1, 64)
            pairs = float((sk[:, 63] // 1024).sum())
            atom = float((sk[:, :63].sum()) + (sk[:, 63] % 1024).sum())
            print(f"   reduced (splat, tile) pairs {pairs:.0f} of I={flat.numel()}  lane-atomics {atom:.0f} ({atom / max(pairs, 1):.2f} per pair)"
                  f"  v_render zeros: {float((vr == 0).float().mean()):.3f}  v_alpha zeros: {float((va == 0).float().mean()):.3f}")
        print(f"{name:22s} median {res[name]['median_ms']:.4f} ms  min {res[name]['min_ms']:.4f} ms  {res[name].get('rel_diff_vs_v0', '')}")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
