#!/usr/bin/env python
"""DEV TOOL: A/B differently-compiled builds of the SAME sources on the bench step, on ONE box in ONE gpurun call.

    python tools/lab/ab_step.py --build name1=-DFOO name2=-DBAR,-DBAZ     (CPU container: lib/libartdeco_hip.<name>.so)
    gpurun -- python tools/lab/ab_step.py [--check=name,..] [--only=name,..]                     (times the default build and every variant)

Each build runs `bench.py --no-extra-configs --no-cpu-baseline --no-frontend` in its own process (ARTDECO_HIP_LIB selects
the library), twice, interleaved (A B C A B C) so that clock drift of the box hits every build alike.  --check=... also
runs the bit-exact / oracle raster tests against each variant first."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
LIBDIR = os.path.join(ROOT, "artdeco_amd", "lib")


def build(specs):
    from artdeco_amd import build as B
    for spec in specs:
        name, _, flags = spec.partition("=")
        extra = tuple(f for f in flags.split(",") if f)
        print(name, extra, B.build(variant=name, extra_flags=extra), flush=True)


PASS = [a for i, a in enumerate(sys.argv) if a in ("--gaussians", "--width", "--height") or (i and sys.argv[i - 1] in ("--gaussians", "--width", "--height"))]


def main():
    if "--build" in sys.argv:
        return build(sys.argv[sys.argv.index("--build") + 1:])
    libs = {"default": os.path.join(LIBDIR, "libartdeco_hip.so")}
    for f in sorted(os.listdir(LIBDIR)):
        if f.startswith("libartdeco_hip.") and f.endswith(".so") and f != "libartdeco_hip.so":
            libs[f[len("libartdeco_hip."):-3]] = os.path.join(LIBDIR, f)
    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
    if only:
        libs = {k: v for k, v in libs.items() if k in only[0] or k == "default"}
    res = {k: [] for k in libs}
    check = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--check=")]
    if check:
        for name, so in libs.items():
            if name not in check[0]:
                continue
            env = dict(os.environ, ARTDECO_HIP_LIB=so)
            r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_raster.py")],
                               env=env, cwd=ROOT, capture_output=True, text=True)
            print(f"[check] {name:24s} rc={r.returncode} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
    for rep in range(2):
        for name, so in libs.items():
            env = dict(os.environ, ARTDECO_HIP_LIB=so)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "10", "--no-extra-configs",
                                "--no-cpu-baseline", "--no-frontend", *PASS], env=env, cwd=ROOT, capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode or not line:
                print(f"{name}: FAILED rc={r.returncode}\n{r.stderr[-600:]}", flush=True)
                continue
            d = json.loads(line[-1])
            st = d.get("stage_ms", {})
            res[name].append({"ms_per_step": d["ms_per_step"], "raster_fwd": st.get("raster_fwd"), "raster_bwd": st.get("raster_bwd"),
                              "bwd_in_region": d.get("raster_bwd_ms"), "stage_ms": st})
            print(f"{name:24s} step {d['ms_per_step']:.4f} ms  raster_fwd {st.get('raster_fwd')}  raster_bwd {st.get('raster_bwd')} "
                  f"(timed region {d.get('raster_bwd_ms'):.4f})  project_bwd {st.get('project_bwd')}  adam_multi {st.get('adam_multi')}  "
                  f"lod_fwd {st.get('lod_params_fwd')}  lod_bwd {st.get('lod_params_bwd')}  project_fwd {st.get('project_fwd')}  "
                  f"bin {st.get('bin_count')} {st.get('bin_scatter')} {st.get('bin_sort')}", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
