#!/usr/bin/env python
"""DEV TOOL: tools/lab/gemm_lab.hip against torch's F.linear (hipBLASLt) on the MASt3R block shapes at 768 tokens."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "gemm_lab.so")
if "--build-only" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", os.path.join(HERE, "gemm_lab.hip"), "-o", SO], check=True)
    sys.exit(0)
import torch
import torch.nn.functional as F
lib = ctypes.CDLL(SO)
dev = torch.device("cuda:0")
P = ctypes.c_void_p


def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M = 768
shapes = [("enc qkv", 1024, 3072), ("enc proj", 1024, 1024), ("enc fc1", 1024, 4096), ("enc fc2", 4096, 1024),
          ("dec qkv", 768, 2304), ("dec proj/q", 768, 768), ("dec kv", 768, 1536), ("dec fc1", 768, 3072), ("dec fc2", 3072, 768)]
g = torch.Generator(device=dev).manual_seed(0)
for name, K, N in shapes:
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).half()
    b = torch.randn(N, device=dev, generator=g).half()
    ref = F.linear(A.float(), W.float(), b.float())
    t_ref = timeit(lambda: F.linear(A, W, b))
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    ws = torch.empty(M, N, device=dev, dtype=torch.float32)
    res = []
    for variant, vn in ((0, "128x128"), (1, "64x128"), (2, "128x64"), (3, "64x64")):
        for sk in (1, 2, 4):
            if sk > 1 and K < 2048 and (N // (128 if variant in (0, 1) else 64)) * (M // (128 if variant in (0, 2) else 64)) >= 192:
                continue
            run = lambda: lib.gemm_lab(variant, sk, P(A.data_ptr()), P(W.data_ptr()), P(b.data_ptr()), P(out.data_ptr()), P(ws.data_ptr()), M, N, K,
                                       P(torch.cuda.current_stream().cuda_stream))
            out.zero_()
            assert run() == 0
            torch.cuda.synchronize()
            err = float((out.float() - ref).abs().max() / ref.abs().max())
            res.append((timeit(run), f"{vn}/k{sk}", err))
    res.sort()
    best = res[0]
    print(f"{name:11s} K={K:4d} N={N:4d}: torch {t_ref:6.2f} us | best {best[1]} {best[0]:6.2f} us (err {best[2]:.1e}) | " +
          " ".join(f"{n}:{t:.1f}" for t, n, e in res[:6]) + (f"  MAXERR {max(e for _, _, e in res):.1e}"), flush=True)
