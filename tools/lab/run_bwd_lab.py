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
VARIANTS = {
    "v0_product": (0, []),
    "v1_no_atomics": (1, []),
    "v2_no_reduction": (2, []),
    "v3_cull_exp_only": (3, []),
    "v4_lds_transpose": (4, []),
    "v0_no_slp": (0, ["-fno-slp-vectorize"]),
    "v2_no_slp": (2, ["-fno-slp-vectorize"]),
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
    from artdeco_amd import mapper
    dev = torch.device("cuda:0")
    N, W, H = 1_000_000, 1920, 1080
    if len(sys.argv) >= 4 and sys.argv[1].isdigit():
        N, W, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
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
    print(f"N={N} {W}x{H} I={flat.numel()}")
    for name in VARIANTS:
        so = os.path.join(BUILD, f"lab_{name}.so")
        if not os.path.exists(so):
            continue
        lib = ctypes.CDLL(so)
        lib.lab_raster_bwd.restype = ctypes.c_int
        P = ctypes.c_void_p
        lib.lab_raster_bwd.argtypes = [ctypes.c_int, ctypes.c_int, P, P, P, ctypes.c_int64, P, P, P, P, P, P, P, P]

        def run(v_rec):
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
        if lib.lab_variant() in (0, 4):
            if ref is None:
                ref = v_rec.clone()
            else:
                res[name]["rel_diff_vs_v0"] = float((v_rec - ref).norm() / ref.norm())
        print(f"{name:22s} median {res[name]['median_ms']:.4f} ms  min {res[name]['min_ms']:.4f} ms  {res[name].get('rel_diff_vs_v0', '')}")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
