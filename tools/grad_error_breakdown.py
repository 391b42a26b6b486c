#!/usr/bin/env python
"""Where does the fp32 gradient error of the rasteriser come from?  (GPU diagnostic, test infrastructure: uses oracle/.)

For a window of a synthetic scene: fp64 autograd over the oracle vs the HIP path, (a) at the raster level -- gradients
with respect to the projected quantities (means2d, conics, opacity, colour/depth), i.e. raster_bwd_kernel alone -- and
(b) at the input level (means, quats, scales, opacities, SH, viewmat), i.e. raster_bwd + project_bwd.
    python tools/grad_error_breakdown.py [N W H tx0 ty0 tx1 ty1]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
torch.set_num_threads(1)
import artdeco_amd  # noqa: E402

artdeco_amd.install_dropins()
from artdeco_amd import _lib, rasterizer  # noqa: E402
from oracle import gsplat_oracle as go  # noqa: E402
from test_raster import _tilted_viewmat  # noqa: E402


def err(name, gh, gref):
    gh, gref = gh.double(), gref.double()
    d = gh - gref
    print(f"  {name:12s} rel_l2 {float(d.norm() / gref.norm()):.3e}  rel_max {float(d.abs().max() / gref.abs().max()):.3e}"
          f"  |g|max {float(gref.abs().max()):.3e}")


def main():
    a = [int(x) for x in sys.argv[1:]]
    N, W, H = (a + [200_000, 512, 384])[:3] if len(a) >= 3 else (200_000, 512, 384)
    window = tuple(a[3:7]) if len(a) >= 7 else (10, 8, 18, 14)
    dev = torch.device("cuda:0")
    lib = _lib.load()
    sc = dict(go.synthetic_scene(N, W, H, 0), viewmat=_tilted_viewmat(2))
    o = go.rasterization_window(sc, window)
    tx0, ty0, tx1, ty1 = window
    ys, xs = slice(ty0 * 16, min(ty1 * 16, H)), slice(tx0 * 16, min(tx1 * 16, W))
    keep = torch.zeros(H, W, 1, dtype=torch.bool)
    keep[ys, xs] = ~o["extras"]["knife"][ys, xs, None]
    g = torch.Generator().manual_seed(5)
    v_r = torch.randn(H, W, 4, generator=g) * keep
    v_a = torch.randn(H, W, 1, generator=g) * keep
    # oracle with the raster-level intermediates kept: redo the window pass by hand
    ids = o["ids"]
    L = {k: sc[k][ids].double().clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    L["viewmat"] = sc["viewmat"].double().clone().requires_grad_(True)
    p = go.project(L["means"], L["quats"], L["scales"], L["opacities"], L["viewmat"], sc["K"].double(), W, H, 0.01)
    vis = o["p32"]["valid"][ids]
    dirs = torch.where(vis[:, None], L["means"] - go.camera_position(L["viewmat"])[None], torch.ones(len(ids), 3, dtype=torch.float64))
    feat = torch.cat([go.sh_to_rgb(3, dirs, L["colors"]), p["depths"][:, None]], -1)
    m2, cn, op_r = p["means2d"], p["conics"], L["opacities"] * 1.0
    for t in (m2, cn, feat, op_r):
        t.retain_grad()
    import numpy as np
    sub = dict(o["isects"]); sub["flatten_ids"] = np.searchsorted(ids.numpy(), o["isects"]["flatten_ids"]).astype(np.int32)
    ro, ao, _ = go.rasterize_to_pixels(m2, cn, feat, op_r, W, H, sub, tile_window=window)
    ((ro * v_r.double()).sum() + (ao * v_a.double()).sum()).backward()

    t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}
    leaves = {k: t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors", "viewmat")}
    out = rasterizer.render_camera(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                                   leaves["viewmat"], t["K"], W, H, sh_degree=3, eps2d=0.01)
    rc, ra, radii, rec, tpg, flat, offs, isect_ids, last_ids, main_ids, final_T = out
    # (a) raster level: call the backward kernel directly
    v_rec = torch.zeros(N, 12, device=dev)
    vr, va = v_r.to(dev).contiguous(), v_a.to(dev).contiguous()
    rc_ = lib.adk_raster_bwd(W, H, rec.data_ptr(), flat.data_ptr(), offs.data_ptr(), flat.numel(), None, final_T.data_ptr(),
                             last_ids.data_ptr(), vr.data_ptr(), va.data_ptr(), v_rec.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc_, "adk_raster_bwd")
    torch.cuda.synchronize()
    vrec = v_rec.cpu()[ids].double()
    recs = rec.detach().cpu()[ids].double()
    a_, b_, c_ = recs[:, 4], recs[:, 5], recs[:, 6]
    vm2 = torch.stack([a_ * vrec[:, 0] + b_ * vrec[:, 1], b_ * vrec[:, 0] + c_ * vrec[:, 1]], -1)
    print(f"N={N} {W}x{H} window={window} Gaussians in window={len(ids)} knife={1 - float(keep[ys, xs].double().mean()):.3f}")
    print("forward:")
    print(f"  render max err {float(((rc.detach().cpu().double() - ro.detach()).abs() * keep).max()):.3e} (scale {float(ro.abs().max()):.2f})")
    print("(a) raster_bwd alone (gradients w.r.t. projected quantities):")
    err("means2d", vm2, m2.grad)
    err("conics", vrec[:, 4:7], cn.grad)
    err("opacity", vrec[:, 2], op_r.grad)
    err("rgb", vrec[:, 8:11], feat.grad[:, :3])
    err("depth", vrec[:, 11], feat.grad[:, 3])
    print("(b) raster_bwd + project_bwd:")
    ((rc * vr).sum() + (ra * va).sum()).backward()
    for k in ("means", "quats", "scales", "opacities", "colors"):
        err(k, leaves[k].grad.cpu()[ids], L[k].grad)
    err("viewmat", leaves["viewmat"].grad.cpu()[:3], L["viewmat"].grad[:3])
    # (c) project_bwd alone: feed the ORACLE's raster-level gradients (rounded to fp32) through adk_project_bwd
    print("(c) the same with fp32-ROUNDED INPUTS in the fp64 oracle (how much of (b) is input rounding of means2d/conics):")
    m2r = rec.detach().cpu()[ids][:, :2].double().requires_grad_(True)
    cnr = rec.detach().cpu()[ids][:, 4:7].double().requires_grad_(True)
    ftr = rec.detach().cpu()[ids][:, 8:12].double().requires_grad_(True)
    opr = sc["opacities"][ids].double().requires_grad_(True)
    ro2, ao2, _ = go.rasterize_to_pixels(m2r, cnr, ftr, opr, W, H, sub, tile_window=window)
    ((ro2 * v_r.double()).sum() + (ao2 * v_a.double()).sum()).backward()
    err("means2d", vm2, m2r.grad)
    err("conics", vrec[:, 4:7], cnr.grad)
    err("opacity", vrec[:, 2], opr.grad)
    err("rgb", vrec[:, 8:11], ftr.grad[:, :3])


if __name__ == "__main__":
    main()
