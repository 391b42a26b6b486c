#!/usr/bin/env python
"""Golden vectors from the UPSTREAM gsplat (>= 1.5) and from on-the-fly-nvs' diff_gaussian_rasterization.adamUpdate -- the two natives of
ARTDECO's hot path whose sources are NOT in the reference tree ("parity unpinned").  This container has neither package
and no network, so the vectors cannot be produced here; this script is what produces them on ANY box that has the upstream wheels
(INTEGRATION.md section 6), and tests/test_gsplat_pinning.py then holds BOTH oracle/gsplat_oracle.py (+ oracle/adam_oracle.py) and the HIP
path to them: integers (radii, tiles per Gaussian, 64-bit sort keys, sorted ids, tile offsets) bit for bit, floats at 1e-4.

    pip install "gsplat>=1.5"            # and, for the Adam vectors, the on-the-fly-nvs fork of diff-gaussian-rasterization
    python tests/golden/make_golden_gsplat.py [--device cuda:0] [--out tests/golden]
    python -m pytest tests/test_gsplat_pinning.py

Call sites restated: gsplat.rendering.rasterization as Reconstruct/scene/scene_models/h3dgsv3.py:664-680 calls it (one camera,
render_mode "RGB+D", rasterize_mode "classic", absgrad False, packed False, sh_degree 3, eps2d 0.01); adamUpdate as
Reconstruct/scene/optimizers.py:116-128,144-156 calls it (0-dim, [N] and [N,M] learning rates).

What is written (one .npz per case; inputs are re-generated from the seed by the test and checked by SHA-256, not stored):
  gsplat_<case>.npz   source, gsplat_version, device, input_sha256,
                      radii [N,2] i32, means2d [N,2], depths [N], conics [N,3], tiles_per_gauss [N] i32, isect_ids [I] i64, flatten_ids [I] i32,
                      isect_offsets [th,tw] i32, render [H,W,4], alphas [H,W,1],
                      loss weights are re-generated from the seed; v_means, v_quats, v_scales, v_opacities, v_colors, v_viewmat,
                      and -- when gsplat.cuda._torch_impl runs on CPU -- torch_impl_* copies of the projection / tile-intersection outputs
  adam_<case>.npz     param / exp_avg / exp_avg_sq after one adamUpdate for each learning-rate shape

--self-check writes the same files from THIS repo's oracle (source = "oracle-selfcheck") into a scratch directory: it exists so that the
test's machinery is exercised in CI without upstream (tests/test_gsplat_pinning.py::test_pinning_machinery_*), never to be committed as a pin.
The script refuses to run against this repo's own drop-in `gsplat` module (artdeco_amd/dropin): that would pin the code to itself.
"""
from __future__ import annotations

import argparse
import hashlib
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

# (name, N, W, H, seed, sigma_px): SURVEY.md 8(d)'s seeded frustum-uniform clouds at sizes a CPU oracle finishes in seconds; "dense" has
# long tile lists and saturating pixels (the terminate-before-adding rule), "tiny" is the smoke size of __graft_entry__.smoke()
CASES = [("tiny", 1500, 96, 64, 0, 2.0), ("small", 4000, 160, 112, 11, 2.0), ("medium", 20000, 256, 192, 12, 2.0), ("dense", 6000, 128, 96, 13, 5.0)]
EPS2D = 0.01


def camera(seed):
    """A seeded rigid world->camera matrix close to the identity (the 8(d) clouds are frustum-uniform for the identity camera)."""
    g = torch.Generator().manual_seed(1000 + seed)
    w = 0.04 * torch.randn(3, generator=g)
    th = float(w.norm())
    k = w / th
    Kx = torch.tensor([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = torch.eye(3) + math.sin(th) * Kx + (1 - math.cos(th)) * (Kx @ Kx)
    vm = torch.eye(4)
    vm[:3, :3] = R
    vm[:3, 3] = 0.05 * torch.randn(3, generator=g)
    return vm


def scene(case):
    from oracle import gsplat_oracle as go   # the INPUT generator only (8(d)'s clouds); no oracle result enters a golden file
    name, N, W, H, seed, sigma = case
    sc = go.synthetic_scene(N, W, H, seed=seed, sigma_px=sigma)
    sc["viewmat"] = camera(seed)
    g = torch.Generator().manual_seed(2000 + seed)
    sc["w_render"] = torch.randn(H, W, 4, generator=g)
    sc["w_alpha"] = torch.randn(H, W, 1, generator=g)
    return sc


def input_sha(sc):
    h = hashlib.sha256()
    for k in ("means", "quats", "scales", "opacities", "colors", "viewmat", "K", "w_render", "w_alpha"):
        h.update(np.ascontiguousarray(sc[k].numpy()).tobytes())
    return h.hexdigest()


def upstream_gsplat():
    import gsplat
    where = os.path.abspath(getattr(gsplat, "__file__", "") or "")
    if os.sep + "artdeco_amd" + os.sep in where or where.startswith(os.path.join(ROOT, "artdeco_amd")):
        raise SystemExit(f"`import gsplat` resolved to this repo's drop-in ({where}): run with the upstream wheel first on sys.path "
                         "(do not call artdeco_amd.install_dropins() / do not put artdeco_amd/dropin on PYTHONPATH)")
    ver = getattr(gsplat, "__version__", "?")
    return gsplat, ver


def run_upstream(sc, device):
    """gsplat.rendering.rasterization exactly as h3dgsv3.py:664-680 calls it, + its autograd gradients."""
    gsplat, ver = upstream_gsplat()
    from gsplat.rendering import rasterization
    dev = torch.device(device)
    W, H = sc["width"], sc["height"]
    leaves = {k: sc[k].to(dev).clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    viewmats = sc["viewmat"].to(dev)[None].clone().requires_grad_(True)
    colors, alphas, meta = rasterization(means=leaves["means"], quats=leaves["quats"], scales=leaves["scales"], opacities=leaves["opacities"],
                                         colors=leaves["colors"], viewmats=viewmats, Ks=sc["K"].to(dev)[None], width=W, height=H,
                                         render_mode="RGB+D", rasterize_mode="classic", absgrad=False, packed=False, sh_degree=3, eps2d=EPS2D)
    loss = (colors[0] * sc["w_render"].to(dev)).sum() + (alphas[0] * sc["w_alpha"].to(dev)).sum()
    loss.backward()
    c = lambda t: t.detach().cpu().numpy()
    out = {"source": "gsplat.rendering.rasterization", "gsplat_version": str(ver), "device": str(torch.cuda.get_device_name(dev)) if dev.type == "cuda" else "cpu",
           "render": c(colors[0]), "alphas": c(alphas[0]), "radii": c(meta["radii"][0]).astype(np.int32).reshape(-1, 2) if meta["radii"].dim() == 3
           else np.repeat(c(meta["radii"][0]).astype(np.int32)[:, None], 2, 1),
           "means2d": c(meta["means2d"][0]), "depths": c(meta["depths"][0]), "conics": c(meta["conics"][0])}
    for k_meta, k_out, dt in (("tiles_per_gauss", "tiles_per_gauss", np.int32), ("isect_ids", "isect_ids", np.int64), ("flatten_ids", "flatten_ids", np.int32),
                              ("isect_offsets", "isect_offsets", np.int32)):
        if k_meta in meta:
            a = c(meta[k_meta]).astype(dt)
            out[k_out] = a[0] if (k_meta in ("tiles_per_gauss", "isect_offsets") and a.ndim >= 2 and a.shape[0] == 1) else a
    for k in leaves:
        out["v_" + k] = c(leaves[k].grad)
    out["v_viewmat"] = c(viewmats.grad[0])
    return out


def run_torch_impl(sc):
    """gsplat.cuda._torch_impl (upstream's own pure-torch reference) on CPU, as far as it runs without the compiled extension: projection
    and tile intersection.  Signatures are looked up, not assumed; a piece that does not run is reported and left out."""
    out, notes = {}, []
    try:
        upstream_gsplat()
        from gsplat.cuda import _torch_impl as ti
        W, H = sc["width"], sc["height"]
        covars, _ = ti._quat_scale_to_covar_preci(sc["quats"], sc["scales"], compute_covar=True, compute_preci=False)
        r = ti._fully_fused_projection(sc["means"], covars, sc["viewmat"][None], sc["K"][None], W, H, eps2d=EPS2D)
        radii, means2d, depths, conics = r[0], r[1], r[2], r[3]
        out.update(torch_impl_radii=radii[0].numpy().astype(np.int32), torch_impl_means2d=means2d[0].numpy(), torch_impl_depths=depths[0].numpy(),
                   torch_impl_conics=conics[0].numpy())
        tw, th = (W + 15) // 16, (H + 15) // 16
        try:
            tpg, ids, flat = ti._isect_tiles(means2d, radii, depths, 16, tw, th, sort=True)
            out.update(torch_impl_tiles_per_gauss=tpg[0].numpy().astype(np.int32), torch_impl_isect_ids=ids.numpy().astype(np.int64),
                       torch_impl_flatten_ids=flat.numpy().astype(np.int32))
            off = ti._isect_offset_encode(ids, 1, tw, th)
            out["torch_impl_isect_offsets"] = off[0].numpy().astype(np.int32)
        except Exception as e:  # noqa: BLE001
            notes.append(f"_isect_tiles/_isect_offset_encode: {e!r}")
    except SystemExit:
        raise
    except Exception as e:  # noqa: BLE001
        notes.append(f"_torch_impl: {e!r}")
    out["torch_impl_notes"] = "; ".join(notes)
    return out


def run_selfcheck(sc):
    """The same file layout from THIS repo's oracle: exercises the test machinery only (source says so)."""
    from oracle import gsplat_oracle as go
    W, H = sc["width"], sc["height"]
    r, a, meta = go.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["colors"], sc["viewmat"], sc["K"], W, H, eps2d=EPS2D)
    lv = {k: sc[k].double().clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors", "viewmat")}
    r64, a64, _ = go.rasterization(lv["means"], lv["quats"], lv["scales"], lv["opacities"], lv["colors"], lv["viewmat"], sc["K"], W, H, eps2d=EPS2D,
                                   grad_dtype=torch.float64)
    ((r64 * sc["w_render"].double()).sum() + (a64 * sc["w_alpha"].double()).sum()).backward()
    isx = meta["isects"]
    out = {"source": "oracle-selfcheck", "gsplat_version": "none", "device": "cpu", "render": r.numpy(), "alphas": a.numpy(),
           "radii": meta["radii"].numpy().astype(np.int32), "means2d": meta["means2d"].numpy(), "depths": meta["depths"].numpy(),
           "conics": meta["conics"].numpy(), "tiles_per_gauss": isx["tiles_per_gauss"], "isect_ids": isx["isect_ids"], "flatten_ids": isx["flatten_ids"],
           "isect_offsets": isx["offsets"].astype(np.int32)}
    for k in ("means", "quats", "scales", "opacities", "colors", "viewmat"):
        out["v_" + k] = lv[k].grad.float().numpy()
    return out


# ------------------------------------------------------------------------------------------------ adamUpdate
ADAM_CASES = [("adam_small", 3000, 21), ("adam_rows", 5000, 22)]
ADAM_HP = dict(b1=0.9, b2=0.999, eps=1e-15)


def adam_inputs(N, seed):
    g = torch.Generator().manual_seed(seed)
    t = {}
    for name, M in (("xyz", 3), ("f_rest", 45), ("opacity", 1), ("rotation", 4)):
        t[name] = dict(param=torch.randn(N, M, generator=g), grad=0.01 * torch.randn(N, M, generator=g), exp_avg=0.001 * torch.randn(N, M, generator=g),
                       exp_avg_sq=1e-6 * torch.rand(N, M, generator=g), M=M)
    vis = torch.rand(N, generator=g) < 0.3
    lrs = {"scalar": torch.tensor(1.6e-4), "rows": 1e-4 * (1 + torch.rand(N, generator=g)), "elems": None}
    return t, vis, lrs


def run_adam_upstream(N, seed, device):
    import diff_gaussian_rasterization as dgr
    where = os.path.abspath(getattr(dgr, "__file__", "") or "")
    if where.startswith(os.path.join(ROOT, "artdeco_amd")):
        raise SystemExit(f"`import diff_gaussian_rasterization` resolved to this repo's drop-in ({where})")
    dev = torch.device(device)
    t, vis, lrs = adam_inputs(N, seed)
    out = {"source": "diff_gaussian_rasterization.adamUpdate", "device": str(torch.cuda.get_device_name(dev))}
    for name, d in t.items():
        M = d["M"]
        for lr_kind in ("scalar", "rows", "elems"):
            lr = lrs[lr_kind] if lr_kind != "elems" else lrs["rows"][:, None].expand(N, M).contiguous()
            if lr_kind == "rows" and M == 1:
                continue
            p, m, v = (d[k].to(dev).clone() for k in ("param", "exp_avg", "exp_avg_sq"))
            with torch.no_grad():
                dgr.adamUpdate(p, d["grad"].to(dev), m, v, vis.to(dev), lr.to(dev), ADAM_HP["b1"], ADAM_HP["b2"], ADAM_HP["eps"], N, M)
            torch.cuda.synchronize()
            for k, x in (("param", p), ("exp_avg", m), ("exp_avg_sq", v)):
                out[f"{name}.{lr_kind}.{k}"] = x.cpu().numpy()
    return out


def run_adam_selfcheck(N, seed):
    from oracle import adam_oracle
    t, vis, lrs = adam_inputs(N, seed)
    out = {"source": "oracle-selfcheck", "device": "cpu"}
    for name, d in t.items():
        M = d["M"]
        for lr_kind in ("scalar", "rows", "elems"):
            if lr_kind == "rows" and M == 1:
                continue
            lr = lrs[lr_kind].numpy() if lr_kind != "elems" else lrs["rows"][:, None].expand(N, M).contiguous().numpy()
            po, mo, vo = adam_oracle.adam_update_oracle(d["param"].numpy(), d["grad"].numpy(), d["exp_avg"].numpy(), d["exp_avg_sq"].numpy(), vis.numpy(),
                                                        lr if lr_kind != "scalar" else np.float32(lr), ADAM_HP["b1"], ADAM_HP["b2"], ADAM_HP["eps"], N, M)
            for k, x in (("param", po), ("exp_avg", mo), ("exp_avg_sq", vo)):
                out[f"{name}.{lr_kind}.{k}"] = x
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--self-check", action="store_true", help="write the files from this repo's ORACLE (test-machinery check; never commit these as pins)")
    ap.add_argument("--cases", default=None, help="comma-separated subset of case names")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    want = set(a.cases.split(",")) if a.cases else None
    if a.self_check and os.path.abspath(a.out) == os.path.join(ROOT, "tests", "golden"):
        raise SystemExit("--self-check writes ORACLE outputs: give it a scratch --out, not tests/golden")
    for case in CASES:
        if want and case[0] not in want:
            continue
        sc = scene(case)
        rec = run_selfcheck(sc) if a.self_check else {**run_upstream(sc, a.device), **run_torch_impl(sc)}
        rec["input_sha256"] = input_sha(sc)
        rec["case"] = np.array([case[1], case[2], case[3], case[4]], np.int64)
        path = os.path.join(a.out, f"gsplat_{case[0]}.npz")
        np.savez_compressed(path, **rec)
        print(f"wrote {path}: source={rec['source']} I={len(rec.get('isect_ids', []))} sha={rec['input_sha256'][:12]}", flush=True)
    for name, N, seed in ADAM_CASES:
        if want and name not in want:
            continue
        try:
            rec = run_adam_selfcheck(N, seed) if a.self_check else run_adam_upstream(N, seed, a.device)
        except ImportError as e:
            print(f"skipped {name}: {e!r} (the on-the-fly-nvs fork of diff-gaussian-rasterization is not installed)", flush=True)
            continue
        rec["case"] = np.array([N, seed], np.int64)
        path = os.path.join(a.out, f"{name}.npz")
        np.savez_compressed(path, **rec)
        print(f"wrote {path}: source={rec['source']}", flush=True)


if __name__ == "__main__":
    main()
