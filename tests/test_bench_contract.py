"""bench.py's accounting and the frame schedule of harness/stream.py (CPU; the numbers themselves are measured on the GPU box)."""
import importlib

import pytest
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_frame_schedule_is_the_stated_cadence():
    from harness import stream
    fl = [stream.frame_flags(i, kf_every=5, slam_every=15, test_hold=8) for i in range(120)]
    assert [i for i, f in enumerate(fl) if f["is_slam_keyframe"]] == list(range(0, 120, 15))
    assert [i for i, f in enumerate(fl) if f["is_test"]] == list(range(8, 120, 8))                    # frame 0 is never a test frame
    important = {i for i, f in enumerate(fl) if f["is_important"]}
    assert important == set(range(0, 120, 5)) | set(range(8, 120, 8))                                # mapper keyframes (incl. SLAM ones) + test frames
    assert all(f["is_important"] for f in fl if f["is_slam_keyframe"])                               # CameraTracker.py:143-148
    assert 0.30 <= len(important) / 120 <= 0.36


def test_roofline_stages_follow_the_survey_formulas():
    bench = importlib.import_module("bench")
    N, V, I, P, W, H = 1_000_000, 900_000, 3_700_000, 1920 * 1080, 1920, 1080
    st = {k: {"mean_ms": v} for k, v in dict(project_fwd=0.06, bin_count=0.03, bin_scatter=0.07, bin_sort=0.04, raster_fwd=0.25, raster_bwd=0.56,
                                             project_bwd=0.24, ssim_fwd=0.05, ssim_bwd=0.06, adam_multi=0.14, lod_params_fwd=0.06,
                                             lod_params_bwd=0.16, photometric_fwd=0.02, photometric_bwd=0.05, photometric_loss=0.01).items()}
    r = bench.roofline_stages(st, N, V, I, P, W, H)
    assert r["raster_bwd"]["alg_bytes"] == 44.0 * I + 28.0 * P + 40.0 * V
    assert r["raster_fwd"]["alg_bytes"] == 44.0 * I + 24.0 * P
    assert r["binning"]["alg_bytes"] == 36.0 * I + 4.0 * (120 * 68) and abs(r["binning"]["ms"] - 0.14) < 1e-9
    assert r["project_fwd"]["alg_bytes"] == 76.0 * N + 216.0 * V
    assert r["ssim_fwd"]["alg_bytes"] == 24.0 * P * 3 and r["ssim_bwd"]["alg_bytes"] == 28.0 * P * 3
    for k, v in r.items():
        if k != "whole_step":
            assert abs(v["frac"] - v["alg_bytes"] / (v["ms"] * 1e-3) / 1e9 / 8000.0) < 1e-3, k
    total_ms = sum(v["mean_ms"] for v in st.values())
    assert abs(r["whole_step"]["ms_sum_of_stages"] - total_ms) < 1e-3                                # every stage's time, also those without a formula
    # the one-call step's fused LoD + projection forward: both formulas minus the 32 B per Gaussian the projection no longer reads back
    fused = {k: v for k, v in st.items() if k not in ("project_fwd", "lod_params_fwd")}
    fused["lod_project_fwd"] = {"mean_ms": 0.11}
    r2 = bench.roofline_stages(fused, N, V, I, P, W, H)
    assert r2["lod_project_fwd"]["alg_bytes"] == 190.0 * N + 76.0 * N + 216.0 * V - 32.0 * N and "project_fwd" not in r2 and "lod_params_fwd" not in r2


def test_project_bwd_bytes_do_not_count_the_traffic_the_fused_colour_adam_removed():
    """SURVEY's SH-bwd term 420 V includes WRITING 192 B of colour gradients, which the fused colour Adam never does; the
    stage is priced at 148 N + (36 + 6 x 192) V (the Adam triple read and written once, the coefficients read once for both purposes)."""
    bench = importlib.import_module("bench")
    N, V = 1_000_000, 880_000
    r = bench.roofline_stages({"project_bwd": {"mean_ms": 0.25}}, N, V, 3_700_000, 1920 * 1080, 1920, 1080)
    assert r["project_bwd"]["alg_bytes"] == 148.0 * N + 1188.0 * V
    assert r["project_bwd"]["alg_bytes"] < 148.0 * N + 420.0 * V + 1152.0 * V


def test_roofline_names_the_backward_form_the_frame_size_uses():
    """raster_tiles.hip:split_parts (round 4): quadrants below 1 600 tiles, halves below 20 000, the whole tile above."""
    bench = importlib.import_module("bench")
    assert "<1, 1, true>" in bench.bwd_kernel_name(512, 384) and "<1, 1, true>" in bench.bwd_kernel_name(648, 486)
    assert "<2, 1, true>" in bench.bwd_kernel_name(1920, 1080) and "<2, 1, true>" in bench.bwd_kernel_name(2592, 1944)
    assert "<2, 2, false>" in bench.bwd_kernel_name(4096, 2160)
    src = open(os.path.join(ROOT, "artdeco_amd", "csrc", "raster_tiles.hip")).read()
    assert "n_tiles < 1600 ? 4 : (n_tiles < 20000 ? 2 : 1)" in src          # the kernel's rule and the label's rule are the same


def test_full_sequence_runs_the_baseline_sequences_from_frame_zero():
    """bench.full_sequence: BASELINE's 300-frame (configs[1], north-star geometry) and 1 000-frame (configs[2]) sequences, each from frame 0
    of a fresh scene (nothing fast-forwarded), once with run_system.py's per-keyframe SLAM loop and once batched; `value` itself is measured
    without gc.freeze()."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "1_000_000, 512, 384, 300)" in src and "args.gaussians, args.width, args.height, 1000)" in src
    assert "fast_forward" not in src.split("def _full_sequence")[1].split("def full_sequence")[0].replace("nothing fast-forwarded", "")
    main = src.split("def main():")[1].split("def bwd_kernel_name")[0]
    timed = main.split("t0 = time.perf_counter()")[0]
    calls = [ln.strip() for ln in timed.splitlines() if not ln.strip().startswith("#")]
    assert not any("freeze_gc()" in ln for ln in calls)     # the headline's timed region starts with the interpreter's default collector
    assert main.index("\n        fused.freeze_gc()") > main.index("elapsed = time.perf_counter() - t0") and "fused.unfreeze_gc()" in main


@pytest.mark.gpu
def test_warm_process_runs_every_frame_kind_and_leaves_the_rng_streams_alone():
    """harness/stream.warm_process (what bench.py / bench_system.py call before anything is timed): a SLAM keyframe, a densified frame
    and a test frame on a throw-away scene, with numpy's / torch's / the device's random streams left exactly where they were (the
    timed stream's keyframe choice and its uniform draws must not depend on whether the process was warmed)."""
    import numpy as np
    import torch
    import artdeco_amd
    artdeco_amd.install_dropins()
    from harness import stream
    kinds = [stream.frame_flags(i, kf_every=3, slam_every=4, test_hold=5) for i in range(9)]
    assert any(k["is_slam_keyframe"] for k in kinds[1:]) and any(k["is_test"] for k in kinds)
    assert any(k["is_important"] and not k["is_test"] for k in kinds[1:])
    dev = torch.device("cuda:0")
    np.random.seed(5)
    torch.manual_seed(5)
    a = (np.random.get_state()[1].copy(), torch.get_rng_state().clone(), torch.cuda.get_rng_state(dev).clone())
    stream.warm_process(dev)
    b = (np.random.get_state()[1], torch.get_rng_state(), torch.cuda.get_rng_state(dev))
    assert (a[0] == b[0]).all() and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
