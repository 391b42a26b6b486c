#!/usr/bin/env python
"""DEV TOOL: tools/lab/reduce_lab.hip -- is a rows-first transposing butterfly (v_permlane32/16_swap before the DPP stages)
cheaper than the shipped wave_reduce10?  `--build-only` here (hipcc cross-compiles), run on the GPU box: checks both variants
against a float64 sum and times them with every SIMD of the chip holding 4 waves."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "reduce_lab.so")
if "--build-only" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
                    os.path.join(HERE, "reduce_lab.hip"), "-o", SO], check=True)
    sys.exit(0)
import torch
lib = ctypes.CDLL(SO)
dev = torch.device("cuda:0")
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
g = torch.Generator().manual_seed(0)
x = torch.randn(10, 64, generator=g).to(dev)
o0, o1 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
s0, w0 = torch.zeros(64, dtype=torch.int32, device=dev), torch.zeros(64, dtype=torch.int32, device=dev)
assert lib.reduce_lab_check(P(x), P(o0), P(s0), P(w0), P(o1), st()) == 0
torch.cuda.synchronize()
want = x.double().sum(dim=1).cpu()
bad0 = [(l, int(s0[l]), float(o0[l]), float(want[int(s0[l])])) for l in range(64) if int(w0[l]) and abs(float(o0[l]) - float(want[int(s0[l])])) > 1e-4]
layout = {0: [0, 2, 1, 3], 1: [8, 8, 9, 9], 2: [4, 6, 5, 7], 3: [8, 8, 9, 9]}
bad1 = []
for l in range(64):
    k = layout[(l % 16) // 4][l // 16]
    if abs(float(o1[l]) - float(want[k])) > 1e-4:
        bad1.append((l, k, float(o1[l]), float(want[k])))
print("shipped reduce: owners", int(w0.sum()), "mismatches", bad0[:4])
print("rows-first reduce: mismatches", bad1[:4])
# round 4: adk::wave_reduce20 -- lane = 16 r + 4 b + l holds, for splat b >> 1, z0 = total of value (b & 1) * 5 + {0, 2, 1, 3}[r], z1 = total of (b & 1) * 5 + 4
x20 = torch.randn(20, 64, generator=g).to(dev)
z0, z1 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
assert lib.reduce20_lab_check(P(x20), P(z0), P(z1), st()) == 0
torch.cuda.synchronize()
want20 = x20.double().sum(dim=1).cpu()
bad2 = []
for l in range(64):
    r, b = l // 16, (l % 16) // 4
    k0 = 10 * (b >> 1) + (b & 1) * 5 + [0, 2, 1, 3][r]
    k1 = 10 * (b >> 1) + (b & 1) * 5 + 4
    if abs(float(z0[l]) - float(want20[k0])) > 1e-4 or abs(float(z1[l]) - float(want20[k1])) > 1e-4:
        bad2.append((l, k0, float(z0[l]), float(want20[k0]), k1, float(z1[l]), float(want20[k1])))
print("paired reduce20: mismatches", bad2[:4])
# round 5: adk::wave_reduce20_rows_first -- lane = 16 r + 4 b + l: splat r >> 1, z0 = total of (r & 1) * 5 + {0, 2, 1, 3}[b], z1 = total of (r & 1) * 5 + 4
z0, z1 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
assert lib.reduce20s_lab_check(P(x20), P(z0), P(z1), st()) == 0
torch.cuda.synchronize()
bad3 = []
for l in range(64):
    r, b = l // 16, (l % 16) // 4
    k0 = 10 * (r >> 1) + (r & 1) * 5 + [0, 2, 1, 3][b]
    k1 = 10 * (r >> 1) + (r & 1) * 5 + 4
    if abs(float(z0[l]) - float(want20[k0])) > 1e-4 or abs(float(z1[l]) - float(want20[k1])) > 1e-4:
        bad3.append((l, k0, float(z0[l]), float(want20[k0]), k1, float(z1[l]), float(want20[k1])))
print("paired reduce20, rows first: mismatches", bad3[:4])
blocks, iters = 256 * 4 * 4, 4000          # 4 waves on every SIMD
out = torch.zeros(blocks * 64, device=dev)
for variant, name in ((0, "shipped (in-row DPP stages first)"), (1, "rows first (permlane swaps, then DPP)"),
                      (2, "paired reduce20, per PAIR (2 x (10 fma + reduction))"),
                      (3, "paired reduce20 ROWS FIRST, per PAIR")):
    for _ in range(2):
        lib.reduce_lab_time(variant, blocks, iters, P(out), st())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.reduce_lab_time(variant, blocks, iters, P(out), st())
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    # cycles per reduction per wave slot: 4 waves share a SIMD, so SIMD cycles per reduction = ms * 2.4e6 / (iters * 4)
    print(f"{name:42s} {ms:8.3f} ms  = {ms * 2.4e6 / (iters * 4):6.1f} SIMD cycles per (10 fma + reduction)")
