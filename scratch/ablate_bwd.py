import ctypes, os, subprocess, sys, torch
sys.path.insert(0, '/root/repo')
import artdeco_amd; artdeco_amd.install_dropins()
from artdeco_amd import _lib, mapper, rasterizer
from artdeco_amd.rasterizer import render_camera
dev = torch.device('cuda:0')
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
variants = {"base": [], "no_flush": ["-DADK_ABLATE_NO_FLUSH"], "no_ldsadd": ["-DADK_ABLATE_NO_LDSADD"],
            "no_reduce": ["-DADK_ABLATE_NO_REDUCE"], "no_reduce_ldsadd_flush": ["-DADK_ABLATE_NO_REDUCE", "-DADK_ABLATE_NO_LDSADD", "-DADK_ABLATE_NO_FLUSH"]}
libs = {}
for name, flags in variants.items():
    out = f"/tmp/rt_{name}.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-I{ROOT}/artdeco_amd/csrc", f"-I{ROOT}/include", *flags, f"{ROOT}/artdeco_amd/csrc/raster_tiles.hip", "-o", out], check=True)
    libs[name] = ctypes.CDLL(out)
c = mapper.synthetic_cloud(1_000_000, 1920, 1080, 0)
W, H = 1920, 1080
K = torch.tensor([[c["fx"], 0, W / 2], [0, c["fx"], H / 2], [0, 0, 1]], device=dev)
t = {k: v.to(dev) for k, v in c.items() if torch.is_tensor(v)}
out = render_camera(t["means"], t["quats"], t["scales"], t["opacities"], t["sh"], torch.eye(4, device=dev), K, W, H, sh_degree=3, eps2d=0.01)
col, alpha, radii, rec, tpg, flat, offs, _, last, _ = out
I = flat.numel()
vc = torch.randn(H, W, 4, device=dev); va = torch.randn(H, W, 1, device=dev)
s = torch.cuda.current_stream().cuda_stream
for name, lib in libs.items():
    f = lib.adk_raster_bwd
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int64] + [ctypes.c_void_p] * 7
    vrec = torch.zeros(1_000_000, 12, device=dev)
    def run():
        return f(W, H, rec.data_ptr(), flat.data_ptr(), offs.data_ptr(), I, None, alpha.data_ptr(), last.data_ptr(), vc.data_ptr(), va.data_ptr(), vrec.data_ptr(), s)
    for _ in range(3): assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1)/10:.3f} ms   (I={I})")
