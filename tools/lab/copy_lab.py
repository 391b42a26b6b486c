#!/usr/bin/env python
"""DEV TOOL: time the streaming-copy variants of tools/lab/copy_lab.hip on 1 GiB (--build-only in the CPU container first)."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "copy_lab.so")
if "--build-only" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", os.path.join(HERE, "copy_lab.hip"), "-o", SO], check=True)
    sys.exit(0)
import torch
lib = ctypes.CDLL(SO)
dev = torch.device("cuda:0")
NB = 1 << 30
src = torch.empty(NB // 4, device=dev).normal_()
dst = torch.empty(NB // 4, device=dev)
scr = torch.zeros(4, device=dev)
names = {0: "grid-stride 2048 blocks (adk_stream_copy)", 1: "one float4 per thread", 2: "4 per thread, loads first", 3: "8 per thread",
         4: "4 per thread, nontemporal", 5: "8 per thread, nontemporal", 6: "grid-stride x4 unrolled, 2048 blocks", 7: "grid-stride x4 unrolled, 8192 blocks",
         8: "READ only (4 per thread)", 9: "WRITE only (4 per thread)", 10: "2 per thread", 11: "16 per thread"}
P = ctypes.c_void_p
st = P(torch.cuda.current_stream().cuda_stream)
for v, name in names.items():
    run = lambda: lib.copy_lab(v, P(dst.data_ptr()), P(src.data_ptr()), ctypes.c_int64(NB), P(scr.data_ptr()), st)
    for _ in range(3): assert run() == 0
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[len(ts) // 2]
    traffic = NB * (1 if v in (8, 9) else 2)
    ok = "" if v in (8, 9) else f"  correct={bool(torch.equal(dst, src))}"
    print(f"v{v:2d} {name:45s} {t:.4f} ms  {traffic / t / 1e9:.2f} TB/s{ok}", flush=True)
