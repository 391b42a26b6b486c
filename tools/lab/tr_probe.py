#!/usr/bin/env python
"""DEV TOOL: run tools/lab/tr_probe.hip (build here with --build-only, run on the GPU box) and print, for a few per-lane address
patterns, which (source lane, element) each result element of ds_read_b64_tr_b16 came from."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "tr_probe.so")
if "--build-only" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", os.path.join(HERE, "tr_probe.hip"), "-o", SO], check=True)
    sys.exit(0)
import torch
lib = ctypes.CDLL(SO)
dev = torch.device("cuda:0")
pats = {"8*l": [8 * l for l in range(64)],
        "8*(l%16)+1024*(l//16)": [8 * (l % 16) + 1024 * (l // 16) for l in range(64)],
        "rows of 160 B: (l%16//4)*160 + (l%4)*8 + (l//16)*640": [((l % 16) // 4) * 160 + (l % 4) * 8 + (l // 16) * 640 for l in range(64)]}
for name, addrs in pats.items():
    a = torch.tensor(addrs, dtype=torch.int32, device=dev)
    out = torch.zeros(256, dtype=torch.int16, device=dev)
    rc = lib.tr_probe(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    o = out.cpu().view(64, 4).tolist()
    byaddr = {}
    for l, ad in enumerate(addrs):
        for e in range(4):
            byaddr[ad // 2 + e] = (l, e)
    print("pattern", name, "rc", rc)
    for l in range(64):
        print(f"  lane {l:2d}: idx {o[l]}  <- (lane,elem) {[byaddr.get(v & 0xffff) for v in o[l]]}")
