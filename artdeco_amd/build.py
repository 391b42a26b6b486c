"""In-tree build of libartdeco_hip.so (gfx950 only) with hipcc.

`python -m artdeco_amd.build` or `artdeco_amd.build.build()`.  hipcc cross-compiles
for gfx950 without a GPU, so this also runs in the CPU-only container.  Objects
are rebuilt only when their source (or a shared header) is newer.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIBNAME = "libartdeco_hip.so"

ARCH = "gfx950"
# -fno-slp-vectorize: measured on MI355X (round 2, gpurun_out/r02_ab_*.json): hipcc's SLP vectoriser turns adjacent scalar
# fp32 adds/muls/fmas into v_pk_add/mul/fma_f32, and on gfx950 a packed-fp32 VALU op costs MORE than the two scalar ops
# it replaces.  Whole step 2.44 -> 2.21 ms with it off: raster bwd 0.73 -> 0.62, raster fwd 0.35 -> 0.29, projection
# bwd 0.345 -> 0.29, SSIM fwd 0.060 -> 0.052, LoD fwd 0.070 -> 0.060 ms; nothing got slower.  Results are unchanged
# where they are compared bit-exactly (packed ops are the same IEEE operations per element).
COMMON_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
                "-fno-slp-vectorize", "-I" + CSRC, "-I" + os.path.join(os.path.dirname(HERE), "include")]

# Per-file extra flags.  IEEE-unfused arithmetic wherever the result feeds an
# integer decision (radii, tile ids, sort keys) or is compared bit-exactly
# against the oracle; fast contraction only in the compositing kernels.
EXTRA_FLAGS = {
    "adam.hip": ["-ffp-contract=off"],
    "raster_project.hip": ["-ffp-contract=off"],
    "matching.hip": ["-ffp-contract=off"],
    "knn.hip": ["-ffp-contract=off"],
    "tracker.hip": ["-ffp-contract=off"],
    "densify.hip": ["-ffp-contract=off"],  # sample positions / resize weights follow torch's operation sequence (a fused u*k-1 moves a grid_sample position by 1e-7 of the map)
    "voxel.hip": ["-ffp-contract=off"],    # voxel indices are floor((p - min) / size): integer-deciding fp32 chain  # same arithmetic as the host-compiled test harness (tests/host/)
}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def sources() -> list[str]:
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime() -> float:
    m = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            if f.endswith((".hpp", ".h")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _compile_one(hipcc: str, src: str, hdr_m: float, force: bool, verbose: bool, objdir: str = OBJDIR,
                 extra: tuple = ()) -> str:
    obj = os.path.join(objdir, src[:-4] + ".o")
    sp = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(sp), hdr_m):
        return obj
    cmd = [hipcc, *COMMON_FLAGS, *EXTRA_FLAGS.get(src, []), *extra, "-c", sp, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def build(force: bool = False, verbose: bool = False, variant: str | None = None, extra_flags: tuple = ()) -> str:
    """variant/extra_flags: developer A/B builds (lib/libartdeco_hip.<variant>.so, selected with ARTDECO_HIP_LIB)."""
    objdir = OBJDIR if variant is None else OBJDIR + "_" + variant
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    hdr_m = _headers_mtime()
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs) or 1)) as ex:
        objs = list(ex.map(lambda s: _compile_one(hipcc, s, hdr_m, force, verbose, objdir, tuple(extra_flags)), srcs))
    out = lib_path() if variant is None else os.path.join(LIBDIR, f"libartdeco_hip.{variant}.so")
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(out) or os.path.getmtime(out) < newest:
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return out


if __name__ == "__main__":
    _variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    _extra = tuple(a.split("=", 1)[1] for a in sys.argv if a.startswith("--extra-flags="))
    print(build(force="--force" in sys.argv, verbose=True, variant=_variant, extra_flags=_extra))
