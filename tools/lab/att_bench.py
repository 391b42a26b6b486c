#!/usr/bin/env python
"""DEV TOOL: time adk_attention_fwd_f16 against torch's scaled_dot_product_attention at the frontend's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from artdeco_amd import attention as att
dev = torch.device("cuda:0")


def timeit(fn, n=200):  # noqa: E306
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (B, H, N) in [(1, 16, 768), (1, 12, 768), (2, 16, 768)]:
    qkv5 = torch.randn(B, N, 3, H, 64, device=dev).half()
    qkv = qkv5.transpose(1, 3)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    t_new = timeit(lambda: att.attention(q, k, v))
    t_ref = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, H * 64))
    flops = 4.0 * B * H * N * N * 64
    print(f"B{B} H{H} N{N}: adk {t_new:.2f} us ({flops / t_new / 1e6:.0f} TFLOP/s)   torch sdpa + layout {t_ref:.2f} us", flush=True)
if "--sweep" in sys.argv:
    B, H, Nq = 1, 16, 768
    for Nk in (64, 128, 192, 384, 768, 1536, 3072):
        q = torch.randn(B, H, Nq, 64, device=dev).half()
        k = torch.randn(B, H, Nk, 64, device=dev).half()
        v = torch.randn(B, H, Nk, 64, device=dev).half()
        print(f"Nq {Nq} Nk {Nk}: {timeit(lambda: att.attention(q, k, v)):.2f} us", flush=True)
