"""What hipcc makes of the two tile kernels' hot forms is part of their design (artdeco_amd/csrc/raster_tiles.hip): the backward's halves form
runs at 7 waves per SIMD only while it fits 72 VGPRs WITHOUT scratch, and its first-touch accumulators rest on the register allocator keeping
one register per sum through a branch (the comment at `float c0 .. c9`).  Neither is promised by the language, so both are pinned here: a
compiler or source change that moves them fails this test instead of silently costing a wave per SIMD.  CPU-only (hipcc cross-compiles)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    from artdeco_amd import build as B
    out = str(tmp_path_factory.mktemp("isa") / "raster_tiles.s")
    src = os.path.join(B.CSRC, "raster_tiles.hip")
    cmd = [B._hipcc(), *B.COMMON_FLAGS, *B.EXTRA_FLAGS.get("raster_tiles.hip", []), "--cuda-device-only", "-S", src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


def _kernel(text, mangled_prefix):
    m = re.search(r"^(%s\w*):.*?; Occupancy: (\d+)" % re.escape(mangled_prefix), text, re.S | re.M)
    assert m, mangled_prefix
    body = m.group(0)
    get = lambda k: int(re.search(r"; %s: (\d+)" % k, body).group(1))
    return {"vgprs": get("NumVgprs"), "scratch": get("ScratchSize"), "occupancy": int(m.group(2)),
            "v_mov": len(re.findall(r"^\s+v_mov_b32", body, re.M)), "lds": int(re.search(r"; LDSByteSize: (\d+)", body).group(1))}


def test_backward_halves_form_fits_seven_waves_without_scratch(asm):
    k = _kernel(asm, "_ZN3adk17raster_bwd_kernelILi2ELi1ELb1ELb0")      # raster_bwd_kernel<2, 1, true, false>: the form 1080p runs
    assert k["scratch"] == 0 and k["vgprs"] <= 72 and k["occupancy"] >= 7, k
    assert k["lds"] <= 3200, k            # parked totals share the staged records' 3 KB (6.4 KB capped a CU at 25 waves)
    assert k["v_mov"] <= 48, k            # 42 with hipcc 7.2; every defined spelling of the accumulators' "no value yet" gave >= 61


def test_backward_quadrant_and_tile_forms(asm):
    q = _kernel(asm, "_ZN3adk17raster_bwd_kernelILi1ELi1ELb1ELb0")      # quadrants: frames with few tiles (512x384)
    assert q["scratch"] == 0 and q["occupancy"] >= 8, q
    t = _kernel(asm, "_ZN3adk17raster_bwd_kernelILi2ELi2ELb0ELb0")      # whole tile: >= 20 000 tiles
    assert t["scratch"] == 0 and t["occupancy"] >= 5, t


def test_forward_forms_keep_eight_waves(asm):
    for name in ("_ZN3adk17raster_fwd_kernelILi2ELi1ELb0ELb1", "_ZN3adk17raster_fwd_kernelILi1ELi1ELb0ELb1", "_ZN3adk17raster_fwd_kernelILi2ELi2ELb0ELb0"):
        k = _kernel(asm, name)
        assert k["scratch"] == 0 and k["vgprs"] <= 64 and k["occupancy"] == 8, (name, k)
