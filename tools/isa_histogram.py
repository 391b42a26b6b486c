#!/usr/bin/env python
"""Static ISA histogram of one kernel of a HIP source file, by instruction class and by loop depth -- the evidence behind
"which instructions does raster_bwd_kernel spend its VALU time on" (profiles/r02_raster_bwd_isa.txt).

    python tools/isa_histogram.py artdeco_amd/csrc/raster_tiles.hip raster_bwd_kernel [extra hipcc flags]

Classes (cost in cycles per wave64 instruction, tools/dpp_bench.hip on MI355X): plain VALU ~2.6, v_pk_* f32 (packed) > 2 plain,
transcendental (v_exp/v_log/v_rcp/v_rsq/v_sqrt) quarter rate, DPP-modified VALU ~8.6, v_permlane*_swap ~7.2, LDS (ds_*),
SALU (s_*), VMEM (global_/buffer_/flat_), waitcnt/nop.  Depth = number of enclosing loops (LLVM's "Loop Header: Depth=" marks)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artdeco_amd import build as B  # noqa: E402


def classify(op, line):
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait/nop"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM" + ("-atomic" if "atomic" in op else "")
    if "permlane" in op:
        return "permlane-swap"
    if "dpp" in op or "row_" in line or "quad_perm" in line:
        return "VALU-DPP"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "VALU-trans"
    if op.startswith("v_pk_"):
        return "VALU-packed"
    if op.startswith("v_mfma"):
        return "MFMA"
    if op.startswith("v_cndmask"):
        return "VALU-select"
    if op.startswith("v_cmp"):
        return "VALU-compare"
    if op.startswith("v_"):
        return "VALU-plain"
    return "other"


def main():
    src, kernel = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [B._hipcc(), *B.COMMON_FLAGS, *B.EXTRA_FLAGS.get(os.path.basename(src), []), *extra, "--cuda-device-only", "-S", src, "-o", out]
        subprocess.run(cmd, check=True, capture_output=True)
        text = open(out).read()
    m = re.search(r"^(_Z\w*%s\w*):.*?s_endpgm" % re.escape(kernel), text, re.S | re.M)
    body = m.group(0).splitlines()
    meta = re.search(r"; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+)", text[m.end():], re.S)
    depth = 0
    hist = collections.defaultdict(collections.Counter)
    depth_of_block = {}
    for ln in body:
        lab = re.match(r"^(\.LBB\d+_\d+):", ln)
        if lab:
            d = re.search(r"Depth=(\d+)", ln)
            if d:
                depth = int(d.group(1))
            elif "Loop Header" not in ln and "in Loop" not in ln and "Parent Loop" not in ln:
                depth = 0
            continue
        if "; =>" in ln or ";   in Loop" in ln or ";     Child Loop" in ln or "Parent Loop" in ln:
            d = re.search(r"Depth=(\d+)", ln)
            if d and "Parent Loop" not in ln and "Child Loop" not in ln:
                depth = int(d.group(1))
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        hist[depth][classify(op, t)] += 1
    print(f"# {kernel} in {src}  flags: {' '.join(B.COMMON_FLAGS[1:5] + extra)}")
    if meta:
        print(f"# VGPRs {meta.group(1)}, scratch {meta.group(2)} B, occupancy {meta.group(3)} waves/SIMD")
    classes = sorted({c for d in hist.values() for c in d})
    print("%-8s" % "depth" + "".join("%15s" % c for c in classes) + "%10s" % "total")
    for d in sorted(hist):
        print("%-8d" % d + "".join("%15d" % hist[d][c] for c in classes) + "%10d" % sum(hist[d].values()))
    print("%-8s" % "all" + "".join("%15d" % sum(hist[d][c] for d in hist) for c in classes) + "%10d" % sum(sum(h.values()) for h in hist.values()))


if __name__ == "__main__":
    main()
