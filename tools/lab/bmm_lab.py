#!/usr/bin/env python
"""Is one batched GEMM (torch.baddbmm over the two decoder branches' stacked weights, M = 2 x 768) faster than the two per-branch
GEMMs the decoder issues today on two HIP streams?  hipGraph replays of 12 levels x the decoder's seven GEMM shapes, fp16."""
import sys
import time

import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
C = 768
shapes = [("qkv", C, 3 * C), ("proj", C, C), ("projq", C, C), ("kv", C, 2 * C), ("proj2", C, C), ("fc1", C, 4 * C), ("fc2", 4 * C, C)]
N = 768
g = torch.Generator().manual_seed(0)
W = {n: [torch.randn(o, i, generator=g).half().to(dev) * 0.02 for _ in range(2)] for n, i, o in shapes}
Bv = {n: [torch.randn(o, generator=g).half().to(dev) for _ in range(2)] for n, i, o in shapes}
Ws = {n: torch.stack(W[n]).transpose(1, 2).contiguous() for n in W}    # [2, in, out]
Bs = {n: torch.stack(Bv[n])[:, None, :].contiguous() for n in W}
X = {n: torch.randn(2, N, i, generator=g).half().to(dev) for n, i, o in shapes}
side = torch.cuda.Stream()


def two_streams():
    main = torch.cuda.current_stream()
    for _ in range(12):
        for n, i, o in shapes:
            side.wait_stream(main)
            y0 = F.linear(X[n][0], W[n][0], Bv[n][0])
            with torch.cuda.stream(side):
                y1 = F.linear(X[n][1], W[n][1], Bv[n][1])
            main.wait_stream(side)


def batched():
    for _ in range(12):
        for n, i, o in shapes:
            y = torch.baddbmm(Bs[n], X[n], Ws[n])


def serial():
    for _ in range(12):
        for n, i, o in shapes:
            y0 = F.linear(X[n][0], W[n][0], Bv[n][0])
            y1 = F.linear(X[n][1], W[n][1], Bv[n][1])


def graph_time(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(gr):
        fn()
    gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for name, fn in (("two per-branch GEMMs, serial", serial), ("two per-branch GEMMs on two streams (today)", two_streams), ("one batched GEMM (baddbmm)", batched)):
    print(f"{name:48s} {graph_time(fn):7.3f} ms per 12 levels x 7 shapes")
