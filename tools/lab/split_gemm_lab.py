#!/usr/bin/env python
"""DEV TOOL (round 5): what a TF32-class GEMM with fp32 RANGE costs on gfx950, which has no TF32 MFMA.

The frontend's headline mode narrows GEMM operands to fp16 (TF32's 10-bit mantissa, NOT its 8-bit exponent: |x| <= 65504).  The
range-safe alternative is a bf16 hi + lo split: x = x_hi + x_lo, W = W_hi + W_lo (bf16 each), x W^T ~= x_hi W_hi^T + x_hi W_lo^T +
x_lo W_hi^T (>= 16 mantissa bits, fp32 exponent range), issued as ONE bf16 GEMM with the contraction tripled -- [x_hi | x_hi | x_lo]
against [W_hi | W_lo | W_hi] -- and fp32 output (torch.mm(..., out_dtype=torch.float32)).  This times, as hipGraph replays of the
MASt3R trunk's own GEMM shapes at 768 tokens (encoder: 24 blocks x {qkv, proj, fc1, fc2} at width 1024; decoder: 2 x 12 blocks x
{qkv, proj, q, kv, proj, fc1, fc2} at width 768):
    fp16 operands, fp16 output            (the headline "tf32eq" mode)
    bf16x3: the split (one elementwise kernel per activation) + the 3K GEMM with fp32 output
    fp32 operands                          (strict mode)
and checks the split product's error against float64.

    python tools/lab/split_gemm_lab.py
"""
import torch

dev = torch.device("cuda:0")
M = 768
ENC = [("enc qkv", 1024, 3072), ("enc proj", 1024, 1024), ("enc fc1", 1024, 4096), ("enc fc2", 4096, 1024)]
DEC = [("dec qkv", 768, 2304), ("dec proj", 768, 768), ("dec q", 768, 768), ("dec kv", 768, 1536), ("dec xproj", 768, 768),
       ("dec fc1", 768, 3072), ("dec fc2", 3072, 768)]
COUNTS = {**{n: 24 for n, _, _ in ENC}, **{n: 24 for n, _, _ in DEC}}   # 24 encoder blocks (per image); 2 x 12 decoder blocks


def split3(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, hi, lo], dim=-1)


def split3w(w):
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo, hi], dim=-1).contiguous()


def graph_time(fn, reps=20, inner=10):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * inner) * 1e3     # us per call


def main():
    torch.manual_seed(0)
    print(f"# M = {M} tokens; us per call as hipGraph replays (10 calls per graph, 20 replays)")
    print(f"# {'shape':12s} {'K':>5s} {'N':>5s} | {'fp16':>7s} | {'bf16x3 gemm':>11s} {'+split':>7s} | {'fp32':>7s} | rel err vs fp64: fp16 / bf16 / bf16x3 / fp32(=TF32-free)")
    tot = {"fp16": 0.0, "bf16x3": 0.0, "fp32": 0.0}
    for name, K, N in ENC + DEC:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        ref = (x.double() @ w.double().t())
        x16, w16 = x.half(), w.half()
        xb, wb = x.bfloat16(), w.bfloat16()
        w3 = split3w(w)
        t16 = graph_time(lambda: torch.mm(x16, w16.t()))
        x3 = split3(x)
        tg = graph_time(lambda: torch.mm(x3, w3.t(), out_dtype=torch.float32))
        ts = graph_time(lambda: split3(x))
        t32 = graph_time(lambda: torch.mm(x, w.t()))
        rel = lambda y: float((y.double() - ref).norm() / ref.norm())
        e16, eb = rel(torch.mm(x16, w16.t())), rel(torch.mm(xb, wb.t()))
        e3, e32 = rel(torch.mm(x3, w3.t(), out_dtype=torch.float32)), rel(torch.mm(x, w.t()))
        print(f"  {name:12s} {K:5d} {N:5d} | {t16:7.1f} | {tg:11.1f} {ts:7.1f} | {t32:7.1f} | {e16:.1e} / {eb:.1e} / {e3:.1e} / {e32:.1e}")
        c = COUNTS[name]
        tot["fp16"] += c * t16; tot["bf16x3"] += c * (tg + ts); tot["fp32"] += c * t32
    print(f"# GEMM time per tracked frame's trunk (1 encode + decoder, us): " + ", ".join(f"{k} {v / 1e3:.2f} ms" for k, v in tot.items()))


if __name__ == "__main__":
    main()
