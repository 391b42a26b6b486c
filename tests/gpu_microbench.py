"""Ad-hoc GPU micro-benchmarks (not a pytest file): python tests/gpu_microbench.py [names...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import artdeco_amd  # noqa: E402

artdeco_amd.install_dropins()
from artdeco_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_copy():
    lib = _lib.load()
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    t = timeit(lambda: lib.adk_stream_copy(b.data_ptr(), a.data_ptr(), n, s))
    print(f"stream_copy 1 GiB: {t*1e6:.1f} us  {2*n/t/1e12:.2f} TB/s (read+write)")


def bench_ssim():
    from fused_ssim_cuda import fusedssim, fusedssim_backward
    for shape in [(1, 3, 1080, 1920), (5, 5, 1080, 1920), (1, 3, 486, 648)]:
        a, b = torch.rand(shape, device=dev), torch.rand(shape, device=dev)
        n = a.numel()
        t = timeit(lambda: fusedssim(1e-4, 9e-4, a, b, True))
        m, d1, d2, d3 = fusedssim(1e-4, 9e-4, a, b, True)
        dl = torch.rand_like(a)
        tb = timeit(lambda: fusedssim_backward(1e-4, 9e-4, a, b, dl, d1, d2, d3))
        ti = timeit(lambda: fusedssim(1e-4, 9e-4, a, b, False))
        print(f"ssim {shape}: fwd(train) {t*1e6:.1f} us = {24*n/t/1e12:.2f} TB/s | bwd {tb*1e6:.1f} us = {28*n/tb/1e12:.2f} TB/s | infer {ti*1e6:.1f} us = {12*n/ti/1e12:.2f} TB/s")


def bench_adam():
    from diff_gaussian_rasterization import adamUpdate
    N = 1_000_000
    for M, frac in [(45, 1.0), (45, 0.3), (3, 1.0), (16, 0.3), (1, 1.0)]:
        p, g, m, v = (torch.randn(N, M, device=dev) for _ in range(4))
        v.abs_()
        vis = torch.rand(N, device=dev) < frac
        lr = torch.tensor(1e-3, device=dev)
        t = timeit(lambda: adamUpdate(p, g, m, v, vis, lr, 0.5, 0.99, 1e-15, N, M))
        nv = int(vis.sum())
        print(f"adam N=1M M={M} vis={frac}: {t*1e6:.1f} us = {(28*nv*M+N)/t/1e12:.2f} TB/s algorithmic")


if __name__ == "__main__":
    names = sys.argv[1:] or ["copy", "ssim", "adam"]
    for n in names:
        globals()["bench_" + n]()
