#!/usr/bin/env python
"""Which GPU precision mode of the MASt3R ViT-L trunk is "the reference's precision"?

The reference runs the model in fp32 with TF32 GEMMs allowed (run_system.py:73 `allow_tf32 = True`): 10-bit operand
mantissas, fp32 accumulation.  gfx950 has no TF32 MFMA; fp16 operands have the same 10-bit mantissa.  This tool measures,
on the ViT-L 512x384 pair with random-init weights shared between CPU and GPU, the error of the model
outputs against the fp32 CPU forward for:
   fp32            everything fp32 on the GPU
   tf32_emulated   fp32 GPU forward with the operands of every Linear rounded to 10 mantissa bits (what a TF32 GEMM sees)
   fp16_fp32stream fp16 GEMM operands, fp32 accumulate / residual stream / LayerNorm / softmax (to_inference_dtype(fp16, fp32_stream=True))
   fp16_fp32stream+heads  the same narrowing for the heads' convolutions and Linear layers (the reference's allow_tf32 covers them too)
   bf16_fp32stream same with bf16 operands (8-bit mantissa)
   bf16_full       whole trunk in bf16 (round 1's fast mode)
and the time of one asymmetric pair inference in each mode.  Error = max |x - ref| / max |ref| and rel_l2 per output."""
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd.mast3r_model import vit_large


def tf32_round(t):
    """round-to-nearest-even to 10 explicit mantissa bits (TF32 operand precision), fp32 container"""
    i = t.contiguous().view(torch.int32)
    r = i + 0x0FFF + ((i >> 13) & 1)
    return (r & ~0x1FFF).view(torch.float32)


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cpu_net = vit_large().eval()
    img1, img2 = torch.rand(1, 3, 384, 512) * 2 - 1, torch.rand(1, 3, 384, 512) * 2 - 1
    t0 = time.perf_counter()
    with torch.inference_mode():
        ref1, ref2 = cpu_net({"img": img1}, {"img": img2})
    cpu_s = time.perf_counter() - t0
    ref = {**{"1." + k: v for k, v in ref1.items()}, **{"2." + k: v for k, v in ref2.items()}}
    g1, g2 = img1.to(dev), img2.to(dev)
    out = {"cpu_fp32_pair_s": cpu_s, "cpu_threads": torch.get_num_threads(), "modes": {}}

    def run(net, name, patch_linear=False):
        orig = F.linear
        if patch_linear:
            F.linear = lambda x, w, b=None: orig(tf32_round(x.float()), tf32_round(w.float()), b)
        try:
            with torch.inference_mode():
                for _ in range(2):
                    r1, r2 = net({"img": g1}, {"img": g2})
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    r1, r2 = net({"img": g1}, {"img": g2})
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 5 * 1e3
        finally:
            F.linear = orig
        got = {**{"1." + k: v for k, v in r1.items()}, **{"2." + k: v for k, v in r2.items()}}
        errs = {}
        for k, v in ref.items():
            a, b = got[k].float().cpu().double(), v.double()
            errs[k] = {"max_rel": float((a - b).abs().max() / b.abs().max()), "rel_l2": float((a - b).norm() / b.norm())}
        out["modes"][name] = {"ms_per_pair_model_only": ms, "worst_max_rel": max(e["max_rel"] for e in errs.values()),
                              "worst_rel_l2": max(e["rel_l2"] for e in errs.values()), "errors": errs}
        print(f"{name:24s} {ms:8.2f} ms  worst max_rel {out['modes'][name]['worst_max_rel']:.2e}  worst rel_l2 {out['modes'][name]['worst_rel_l2']:.2e}", flush=True)

    base = copy.deepcopy(cpu_net).to(dev)
    run(base, "fp32")
    run(base, "tf32_emulated", patch_linear=True)
    run(copy.deepcopy(cpu_net).to(dev).to_inference_dtype(torch.float16, fp32_stream=True), "fp16_fp32stream")
    run(copy.deepcopy(cpu_net).to(dev).to_inference_dtype(torch.float16, fp32_stream=True, heads=True), "fp16_fp32stream+heads")
    torch.backends.cudnn.benchmark = True   # MIOpen find mode instead of immediate mode for the head convolutions
    run(copy.deepcopy(cpu_net).to(dev).to_inference_dtype(torch.float16, fp32_stream=True, heads=True), "same, MIOpen find")
    torch.backends.cudnn.benchmark = False
    run(copy.deepcopy(cpu_net).to(dev).to_inference_dtype(torch.bfloat16, fp32_stream=True), "bf16_fp32stream")
    run(copy.deepcopy(cpu_net).to(dev).to_inference_dtype(torch.bfloat16), "bf16_full")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
