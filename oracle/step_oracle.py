"""fp64 oracle of ONE WHOLE optimisation step of ARTDECO's mapper -- TEST INFRASTRUCTURE, never on the product path.

Restates, on the CPU, everything `SceneModel.optimization_step` differentiates
(/root/reference/Reconstruct/scene/scene_models/h3dgsv3.py:401-469):

  Keyframe.get_Rt                       scene/keyframe.py:150-154 + Reconstruct/utils.py:223-229 (6D rotation -> matrix)
  render(): LoD cull / fade             h3dgsv3.py:626-639   (selection by `dist < 2 d_max`, alpha_ratio between d_max and 2 d_max)
            gather + mlp_cov            h3dgsv3.py:641-662   (Linear 32->32, ReLU, Linear 32->7; scale * sigmoid, normalize(rot * out))
            gsplat rasterization        h3dgsv3.py:664-680   (oracle/gsplat_oracle.py: projection, SH deg 3, binning, compositing; [UPSTREAM])
            bg composite, 1/depth, masks h3dgsv3.py:682-692
  render_from_id(): exposure + clamp    h3dgsv3.py:611-614
  loss                                  h3dgsv3.py:428-449   (outlier mask on unimportant frames -- incl. its `error_map[1]` used twice --,
                                                              radial-decay L1, 1 - fused_ssim, inverse-depth L1, scaling regulariser)

Every integer / boolean DECISION (LoD selection and fade masks, cull, radii, tile lists, depth order, and per pixel: which splats pass
alpha >= 1/255 and where the pixel terminates) comes from an fp32 pass written with the reference's own torch operations / App. A's
expressions; everything differentiable is then evaluated in float64 ON those decisions and differentiated
by torch autograd -- i.e. independently of every hand-derived backward in artdeco_amd/csrc.  The compositing is the per-tile function of
oracle/gsplat_oracle.py (`composite_tile`), run tile by tile in a pool of worker processes: pass 1 renders the frame, the image-space chain
gives dL/d(render, alpha), pass 2 re-runs each tile under autograd with that cotangent, and one last backward carries the per-splat
gradients through projection / SH / mlp_cov / LoD / pose to the 15 leaves.  Cost: ~1e-7 core-seconds per (intersection x pixel), so a
1 M-Gaussian 1080p frame is minutes on one core and seconds on a node's worth.

Parity: the gsplat part is UNPINNED (gsplat's source is not under /root/reference; see gsplat_oracle.py).  The host-side part is pinned:
tests/test_reference_scene_model.py runs the reference's REAL `SceneModel.optimization_step` on CPU (natives = the fp32 oracles) from the
same state and compares loss, masks and every gradient with this file's (fp32 vs fp64: 1e-4).
"""
from __future__ import annotations

import atexit
import json
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch
import torch.nn.functional as F

from oracle import gsplat_oracle as go
from oracle import ssim_oracle

TILE = go.TILE
GAUSS_KEYS = ("xyz", "scaling", "rotation", "opacity", "f_dc", "f_rest", "local_feat", "global_feat")
MLP_KEYS = ("mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias")
KF_KEYS = ("kf.rW2C", "kf.tW2C", "kf.exposure")
#: relative half-width of the band around a rasteriser decision inside which an fp32 evaluation may fall on the other side
#: (fp32 alpha / transmittance carry ~1e-6 relative error after a few hundred blended splats)
KNIFE_EPS = 2e-5
#: half-width of the band around 0 of a hidden unit's pre-activation in mlp_cov (two fp32 evaluations of a 32-term sum of O(1) products)
KNIFE_EPS_RELU = 2e-5
#: the same for the termination test T (1 - alpha) <= 1e-4: T is a product of hundreds of factors, two fp32 evaluations drift apart further
KNIFE_EPS_T = 1e-3
#: pixels per call of composite_tile: [n, 64] float64 temporaries stay in a core's L2 (whole 16x16 tiles made the tile passes DRAM-bound
#: with a node's worth of workers: 96 workers only 2x faster than 8)
P_BLOCK = 64
#: intra-op threads of the serial phases (projection graph, SSIM, final backward): the test process itself runs torch single-threaded
SERIAL_THREADS = 16
#: half-width of the band around an image-space decision of the loss (sign of an L1 term, clamp at 0 / 1, the 0.2 outlier threshold)
IMAGE_KNIFE_TOL = 2e-5


# ----------------------------------------------------------------------------- snapshots of a scene (mirror or the real class)
def snapshot(scene, keyframe_id):
    """(state, kf, cfg) as CPU tensors / python numbers from a `MapperScene` (harness/mapper.py) or the reference's real `SceneModel`
    (same attribute names: gaussian_params[k]["val"], mlp_cov, keyframes, tanfovx ...)."""
    P = scene.gaussian_params
    cpu = lambda t: t.detach().cpu().clone()
    state = {k: cpu(P[k]["val"]) for k in GAUSS_KEYS + ("cls_id", "d_max")}
    for n, p in scene.mlp_cov.named_parameters():
        state["mlp." + n] = cpu(p)
    kf = scene.keyframes[keyframe_id]
    lvl = kf.pyr_lvl
    args = getattr(scene, "args", None)
    cfg = dict(width=scene.width // 2 ** lvl, height=scene.height // 2 ** lvl, tanfovx=float(scene.tanfovx), tanfovy=float(scene.tanfovy),
               sh_degree=int(scene.active_sh_degree), lambda_dssim=float(scene.lambda_dssim), rad_decay=float(scene.rad_decay),
               scaling_reg_factor=float(scene.scaling_reg_factor),
               eps2d=float(getattr(scene, "eps2d", None) if hasattr(scene, "eps2d") else args.low_pass_filter_eps))
    kfd = dict(rW2C=cpu(kf.rW2C), tW2C=cpu(kf.tW2C), exposure=cpu(kf.exposure), image=cpu(kf.image_pyr[lvl]),
               mono_idepth=cpu(kf.get_mono_idepth(lvl)), depth_loss_weight=float(kf.depth_loss_weight))
    return state, kfd, cfg


# ----------------------------------------------------------------------------- host-side pieces, restated
def sixD2mtx(r):
    """Reconstruct/utils.py:223-229."""
    b1 = r[..., 0]
    b1 = b1 / torch.norm(b1, dim=-1, keepdim=True)
    b2 = r[..., 1] - torch.sum(b1 * r[..., 1], dim=-1, keepdim=True) * b1
    b2 = b2 / torch.norm(b2, dim=-1, keepdim=True)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack([b1, b2, b3], dim=-1)


def get_Rt(rW2C, tW2C):
    """scene/keyframe.py:150-154."""
    Rt = torch.eye(4, dtype=rW2C.dtype)
    Rt[:3, :3] = sixD2mtx(rW2C)
    Rt[:3, 3] = tW2C
    return Rt


def radial_decay_kernel(H, W, sigma):
    """Reconstruct/utils.py:818-827 (built in fp32 like the reference; the fp64 chain reads the fp32 values)."""
    y = torch.linspace(-1, 1, steps=H)
    x = torch.linspace(-1, 1, steps=W)
    yy, xx = torch.meshgrid(y, x, indexing="ij")
    return torch.exp(-(xx ** 2 + yy ** 2) / (2 * sigma ** 2))


def _mlp_cov(x, st, relu_mask=None):
    """mlp_cov = Linear(32, 32) -> ReLU -> Linear(32, 7) (h3dgsv3.py:118-123).  relu_mask given: the ReLU's on / off DECISIONS of another
    (fp32) evaluation are applied instead of this one's sign test; returns (output, pre-activation, mask)."""
    z = F.linear(x, st["mlp.0.weight"], st["mlp.0.bias"])
    mask = (z > 0) if relu_mask is None else relu_mask
    h = torch.where(mask, z, torch.zeros_like(z))
    return F.linear(h, st["mlp.2.weight"], st["mlp.2.bias"]), z, mask


def _lod_and_params(st, Rt, decisions=None):
    """h3dgsv3.py:626-662 for one dtype.  decisions=None: make them (fp32 pass) and return them; otherwise evaluate ON them."""
    xyz = st["xyz"]
    cam_centre = Rt.detach().inverse()[:3, 3]
    ob_dist = (xyz - cam_centre).norm(dim=1, keepdim=True)
    d_max = st["d_max"].to(xyz.dtype)
    if decisions is None:
        selection_mask = (ob_dist < 2 * d_max).squeeze(-1)
        alpha_mask = torch.logical_and(ob_dist > d_max, ob_dist < 2 * d_max).squeeze(-1)
        decisions = dict(selection_mask=selection_mask, alpha_mask=alpha_mask)
    sel, am = decisions["selection_mask"], decisions["alpha_mask"]
    alpha_ratio = torch.where(am[:, None], (2 * d_max - ob_dist) / d_max, torch.ones_like(ob_dist))
    opacity = (torch.sigmoid(st["opacity"]) * alpha_ratio)[sel]
    scaling = torch.exp(st["scaling"])[sel]
    rotation = st["rotation"][sel]
    feats = torch.cat([st["f_dc"][sel], st["f_rest"][sel]], dim=1)
    ids = st["cls_id"][sel].squeeze(-1).long()
    scale_rot, z, relu_mask = _mlp_cov(torch.cat([st["global_feat"][ids], st["local_feat"][sel]], dim=1), st, decisions.get("relu_mask"))
    if "relu_mask" not in decisions:       # the pass that MAKES the decisions: the ReLU's are among them (a hidden unit within an ulp of 0)
        decisions["relu_mask"], decisions["relu_pre"] = relu_mask, z.detach()
    scaling = scaling * torch.sigmoid(scale_rot[:, :3])
    rotation = F.normalize(rotation * scale_rot[:, 3:])
    return dict(xyz=xyz[sel], opacity=opacity.squeeze(-1), scaling=scaling, rotation=rotation, feats=feats), decisions


# ----------------------------------------------------------------------------- tile passes
# The per-tile work runs in a pool of SPAWNED worker processes (fresh interpreters: a process that has run a backward pass with a GPU
# present owns autograd device threads, and torch refuses autograd in its forked children; the test process is such a process).  The
# frame's arrays travel as .npy files in a temporary directory that the workers map read-only; the pool outlives a call (importing torch
# costs a worker ~2 s) and is torn down at interpreter exit.
_SH: dict = {}
_POOL = None
_POOL_WORKERS = 0
_WORKER_PREFIX = None
_PASS1 = ("means2d", "conics", "feat", "opac", "flat", "offsets", "means2d32", "conics32", "opac32")
_PASS2 = ("g_col", "g_T", "knife_px")
_SCAN = ("flat", "offsets", "means2d32", "conics32", "opac32")


def _publish(prefix, names):
    for k in names:
        v = _SH[k]
        np.save(f"{prefix}.{k}.npy", v.numpy() if torch.is_tensor(v) else np.asarray(v))
    with open(f"{prefix}.scalars.json", "w") as f:
        json.dump({"n_isects": int(_SH["n_isects"]), "tile_w": int(_SH["tile_w"]), "knife_eps": float(_SH["knife_eps"]),
                   "knife_eps_T": float(_SH["knife_eps_T"])}, f)


def _attach(prefix, names):
    """Worker side: map the arrays of the frame `prefix` (cached: one frame at a time)."""
    global _WORKER_PREFIX
    if _WORKER_PREFIX != prefix:
        _SH.clear()
        _SH.update(json.load(open(f"{prefix}.scalars.json")))
        _SH["grid"] = torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij")
        _WORKER_PREFIX = prefix
    for k in names:
        if k not in _SH:
            a = np.load(f"{prefix}.{k}.npy", mmap_mode="c")   # copy-on-write: torch wants a writable array, nothing writes
            _SH[k] = a if k == "offsets" else torch.from_numpy(np.asarray(a))


def _tile_pixels(tid, dt):
    ty, tx = divmod(tid, _SH["tile_w"])
    ys, xs = _SH["grid"]
    return (tx * TILE + xs).reshape(-1).to(dt) + 0.5, (ty * TILE + ys).reshape(-1).to(dt) + 0.5, tx, ty


def _span(tid):
    off = _SH["offsets"]
    return int(off[tid]), (int(off[tid + 1]) if tid + 1 < off.shape[0] else _SH["n_isects"])


def _decide(g):
    """The tile's fp32 projection outputs (what an fp32 rasteriser takes its per-pixel decisions on), or None for the all-fp64 form."""
    if _SH.get("means2d32") is None:
        return None
    return _SH["means2d32"][g], _SH["conics32"][g], _SH["opac32"][g]


def _fwd_chunk(task):
    prefix, tids = task
    if prefix is not None:
        _attach(prefix, _PASS1)
    m2, cn, ft, op, flat = _SH["means2d"], _SH["conics"], _SH["feat"], _SH["opac"], _SH["flat"]
    eps, eps_T = _SH["knife_eps"], _SH["knife_eps_T"]
    out = []
    with torch.no_grad():
        for tid in tids:
            s, e = _span(tid)
            if e <= s:
                continue
            g = flat[s:e]
            px, py, tx, ty = _tile_pixels(tid, m2.dtype)
            a, b, c, d, dec = m2[g], cn[g], ft[g], op[g], _decide(g)
            cols, Ts, kns = [], [], []
            for p0 in range(0, px.shape[0], P_BLOCK):
                sl = slice(p0, p0 + P_BLOCK)
                col, T, last, ex = go.composite_tile(a, b, c, d, px[sl], py[sl], first_index=s, want_extras=True, knife_eps=eps,
                                                     decide=dec, knife_eps_T=eps_T)
                cols.append(col); Ts.append(T); kns.append(ex[0].to(torch.uint8) + 2 * ex[5].to(torch.uint8))   # bit 0: any edge, bit 1: skip edge
            out.append((tid, torch.cat(cols).numpy(), torch.cat(Ts).numpy(), torch.cat(kns).numpy()))
    return out


def _bwd_chunk(task):
    prefix, tids = task
    if prefix is not None:
        _attach(prefix, _PASS1 + _PASS2)
    m2, cn, ft, op, flat = _SH["means2d"], _SH["conics"], _SH["feat"], _SH["opac"], _SH["flat"]
    g_col, g_T, knife = _SH["g_col"], _SH["g_T"], _SH["knife_px"]
    eps, eps_T = _SH["knife_eps"], _SH["knife_eps_T"]
    ids_all, grads_all, touched = [], [], []
    for tid in tids:
        s, e = _span(tid)
        if e <= s:
            continue
        g = flat[s:e]
        px, py, tx, ty = _tile_pixels(tid, m2.dtype)
        sl2 = (slice(ty * TILE, (ty + 1) * TILE), slice(tx * TILE, (tx + 1) * TILE))
        leaves = [t[g].clone().requires_grad_(True) for t in (m2, cn, ft, op)]
        gc, gT, kn = g_col[sl2].reshape(-1, ft.shape[1]), g_T[sl2].reshape(-1), knife[sl2].reshape(-1)
        dec = _decide(g)
        hit = torch.zeros(g.shape[0], dtype=torch.bool)
        for p0 in range(0, px.shape[0], P_BLOCK):
            sl = slice(p0, p0 + P_BLOCK)
            want = bool(kn[sl].any())
            col, T, last, ex = go.composite_tile(*leaves, px[sl], py[sl], first_index=s, want_extras=want, knife_eps=eps, decide=dec,
                                                 knife_eps_T=eps_T)
            ((col * gc[sl]).sum() + (T * gT[sl]).sum()).backward()       # accumulates into the leaves' .grad
            if want:
                hit |= ex[4][:, kn[sl]].any(1)
        ids_all.append(g.numpy())
        grads_all.append(np.concatenate([l.grad.reshape(g.shape[0], -1).numpy() for l in leaves], axis=1))   # [n, 2+3+4+1]
        if bool(hit.any()):
            touched.append(g[hit].numpy())
    if not ids_all:
        return None
    ids = np.concatenate(ids_all)
    gr = np.concatenate(grads_all)
    uniq, inv = np.unique(ids, return_inverse=True)
    acc = np.zeros((uniq.shape[0], gr.shape[1]))
    np.add.at(acc, inv, gr)
    return uniq, acc, (np.unique(np.concatenate(touched)) if touched else np.zeros(0, np.int64))


def _scan_chunk(task):
    """fp32 decisions only: the Gaussians (rows of the selected set) that sit ON a decision at some live pixel, and how many such
    (splat, pixel) pairs / pixels there are."""
    prefix, tids = task
    if prefix is not None:
        _attach(prefix, _SCAN)
    flat = _SH["flat"]
    eps, eps_T = _SH["knife_eps"], _SH["knife_eps_T"]
    rows, n_pairs, n_pix = [], 0, 0
    with torch.no_grad():
        for tid in tids:
            s, e = _span(tid)
            if e <= s:
                continue
            g = flat[s:e]
            px, py, tx, ty = _tile_pixels(tid, torch.float32)
            a, b, d = _decide(g)
            for p0 in range(0, px.shape[0], P_BLOCK):
                sl = slice(p0, p0 + P_BLOCK)
                dec = go._tile_decisions(a, b, d, px[sl], py[sl])
                on_edge = go.knife_pairs(dec, dec[0], eps, eps_T)
                if bool(on_edge.any()):
                    rows.append(g[on_edge.any(1)].numpy())
                    n_pairs += int(on_edge.sum())
                    n_pix += int(on_edge.any(0).sum())
    return (np.unique(np.concatenate(rows)) if rows else np.zeros(0, np.int64)), n_pairs, n_pix


def _chunks(workers):
    """Tile ids in chunks of roughly equal work (list length), several chunks per worker."""
    off = _SH["offsets"]
    lens = np.diff(np.append(off, _SH["n_isects"])).astype(np.int64)
    order = np.argsort(-lens, kind="stable")
    order = order[lens[order] > 0]
    n_chunks = max(1, min(len(order), workers * 8))
    bins = [[] for _ in range(n_chunks)]
    for i, tid in enumerate(order):           # round-robin over a descending sort: near-equal sums
        bins[i % n_chunks].append(int(tid))
    return [b for b in bins if b]


def _one_thread_worker():
    torch.set_num_threads(1)


def _pool(workers):
    global _POOL, _POOL_WORKERS
    if _POOL is None or _POOL_WORKERS != workers:
        shutdown_pool()
        # a spawned child re-imports the parent's __main__ (by path or by module name) before it unpickles its task: a test runner's or a
        # script's main module must not run again in every worker, and the workers need nothing from it (their functions live in this
        # module).  Hide it while the pool starts its processes.
        main = sys.modules.get("__main__")
        saved = {k: getattr(main, k) for k in ("__file__", "__spec__") if main is not None and hasattr(main, k)}
        try:
            if main is not None:
                if hasattr(main, "__file__"):
                    del main.__file__
                main.__spec__ = None
            _POOL = mp.get_context("spawn").Pool(workers, initializer=_one_thread_worker)
        finally:
            for k, v in saved.items():
                setattr(main, k, v)
        _POOL_WORKERS = workers
        atexit.register(shutdown_pool)
    return _POOL


def shutdown_pool():
    global _POOL
    if _POOL is not None:
        _POOL.terminate()
        _POOL.join()
        _POOL = None


def _run(fn, chunks, workers, prefix):
    """Inline (one intra-op thread: the per-tile tensors are far too small for OpenMP, see gsplat_oracle._one_thread) or in the pool."""
    if prefix is None:
        with go._one_thread():
            return [fn((None, c)) for c in chunks]
    return _pool(workers).map(fn, [(prefix, c) for c in chunks], chunksize=1)


def default_workers():
    return max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 96))


# ----------------------------------------------------------------------------- the step
class _threads:
    """torch intra-op threads for the serial phases (the caller's setting is restored)."""

    def __init__(self, n):
        self.n = max(1, min(n, default_workers()))

    def __enter__(self):
        self.old = torch.get_num_threads()
        torch.set_num_threads(self.n)

    def __exit__(self, *a):
        torch.set_num_threads(self.old)


def _fp32_decisions(state, kf, cfg):
    """Phase A: every decision of the step's forward with the reference's own fp32 operations (harness/mapper.py:render is the same chain)."""
    W, H = cfg["width"], cfg["height"]
    f32 = torch.float32
    st32 = {k: (v.to(f32) if v.is_floating_point() else v) for k, v in state.items()}
    with torch.no_grad():
        Rt32 = get_Rt(kf["rW2C"].to(f32), kf["tW2C"].to(f32))
        par32, dec = _lod_and_params(st32, Rt32)
        fl_x, fl_y = W / (2 * cfg["tanfovx"]), H / (2 * cfg["tanfovy"])
        K32 = torch.tensor([[fl_x, 0, W / 2.0], [0, fl_y, H / 2.0], [0, 0, 1]], dtype=f32)
        p32 = go.project(par32["xyz"], par32["rotation"], par32["scaling"], par32["opacity"], Rt32, K32, W, H, cfg["eps2d"])
        isects = go.isect_tiles(p32["means2d"], p32["radii"], p32["depths"], W, H)
    return st32, par32, dec, K32, p32, isects


def knife_scan(state, kf, cfg, *, workers=None, knife_eps=KNIFE_EPS, knife_eps_T=KNIFE_EPS_T, relu_eps=KNIFE_EPS_RELU):
    """Which Gaussians (bool [N]) sit ON a per-pixel decision of the rasteriser for this view -- fp32 evaluation, (splat, pixel) pairs within
    `knife_eps` of alpha = 1/255 / 0.999 or within `knife_eps_T` of the termination threshold -- and the number of such pairs and pixels;
    and (bool [N], unit index [N], sign [N]) of the visible Gaussians with a hidden unit of mlp_cov within `relu_eps` of its ReLU's switch."""
    workers = default_workers() if workers is None else workers
    with _threads(SERIAL_THREADS):
        st32, par32, dec, K32, p32, isects = _fp32_decisions(state, kf, cfg)
    _SH.clear()
    _SH.update(means2d32=p32["means2d"], conics32=p32["conics"], opac32=par32["opacity"],
               flat=torch.from_numpy(isects["flatten_ids"].astype(np.int64)), offsets=isects["offsets"].reshape(-1), n_isects=isects["n_isects"],
               tile_w=isects["tile_w"], grid=torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij"), knife_eps=knife_eps,
               knife_eps_T=knife_eps_T)
    chunks = _chunks(workers)
    tmpdir = prefix = None
    if workers > 1 and len(chunks) > 1:
        tmpdir = tempfile.mkdtemp(prefix="adk_step_oracle_")
        prefix = os.path.join(tmpdir, "frame")
        _publish(prefix, _SCAN)
    sel_rows = torch.zeros(par32["opacity"].shape[0], dtype=torch.bool)
    pairs = pixels = 0
    for rows, n_pairs, n_pix in _run(_scan_chunk, chunks, workers, prefix):
        sel_rows[torch.from_numpy(rows)] = True
        pairs += n_pairs
        pixels += n_pix
    _SH.clear()
    if tmpdir:
        shutil.rmtree(tmpdir, ignore_errors=True)
    sel_idx = torch.nonzero(dec["selection_mask"]).squeeze(-1)
    on_edge = torch.zeros(state["xyz"].shape[0], dtype=torch.bool)
    on_edge[sel_idx[sel_rows]] = True
    # hidden units of mlp_cov within relu_eps of their ReLU's switch (visible Gaussians only: the others' mlp output reaches nothing)
    vis = p32["radii"].max(dim=1).values > 0
    zmin, unit = dec["relu_pre"].abs().min(dim=1)
    relu_sel = (zmin <= relu_eps) & vis
    relu_edge = torch.zeros(state["xyz"].shape[0], dtype=torch.bool)
    relu_edge[sel_idx[relu_sel]] = True
    relu_unit = torch.full((state["xyz"].shape[0],), -1, dtype=torch.long)
    relu_unit[sel_idx[relu_sel]] = unit[relu_sel]
    relu_sign = torch.zeros(state["xyz"].shape[0])
    relu_sign[sel_idx[relu_sel]] = torch.sign(dec["relu_pre"][relu_sel, unit[relu_sel]])
    return on_edge, pairs, pixels, (relu_edge, relu_unit, relu_sign)


def settle_scene(state, kf, cfg, *, workers=None, knife_eps=1e-4, knife_eps_T=0.0, nudge=2e-3, relu_band=1e-4, relu_nudge=2e-3, max_rounds=6,
                 log=None):
    """A TEST's way of keeping its own scene off the rasteriser's knife edges (as `adjust_targets` does for the loss's): the opacity of every
    Gaussian that sits on a SKIP decision at some pixel of this view -- alpha within `knife_eps` of 1/255 (or of 0.999), far wider than two
    fp32 evaluations differ -- is raised by a relative `nudge`, and the view is scanned again, until no (splat, pixel) pair is left on such
    an edge (each round leaves ~1e-5 of the moved pairs on a new one).  The TERMINATION edge (T (1 - alpha) against 1e-4) is left alone by
    default (knife_eps_T = 0): a transmittance is the product of hundreds of factors, ~1 % of the terminating pixels pass within 1e-3 of the
    threshold whatever one splat's opacity is, and a splat that enters or leaves there carries a weight <= 1e-4 -- nothing a gradient
    comparison at 1e-4 sees; pixel-level comparisons mask those pixels (`raster_knife` of optimisation_step, which keeps KNIFE_EPS_T).
    The ReLU of mlp_cov is a decision of the same kind -- a hidden unit whose pre-activation two fp32 evaluations place on different sides
    of 0 switches that Gaussian's feature gradients on or off (seen at 1 M / 1080p: local_feat / global_feat / mlp.0.* at 1e-4 with every
    other leaf at 1e-5) -- and is handled the same way: a visible Gaussian with a unit within `relu_band` of 0 gets its local_feat moved
    along that unit's weights so that the pre-activation lands `relu_nudge` away.
    Returns (state with the adjusted `opacity` logits and `local_feat`, rounds, edges left)."""
    state = dict(state)
    state["opacity"] = state["opacity"].clone()
    state["local_feat"] = state["local_feat"].clone()
    W1l = state["mlp.0.weight"][:, state["global_feat"].shape[1]:].float()       # the columns of Linear 1 that read local_feat
    pairs, last = -1, None
    for r in range(max_rounds + 1):
        on_edge, pairs, pixels, (relu_edge, relu_unit, relu_sign) = knife_scan(state, kf, cfg, workers=workers, knife_eps=knife_eps,
                                                                               knife_eps_T=knife_eps_T, relu_eps=relu_band)
        n_relu = int(relu_edge.sum())
        if log is not None:
            log.append((r, int(on_edge.sum()), pairs, pixels, n_relu))
        stuck = last is not None and pairs >= last      # the few that opacity cannot move: a pixel centre ON a splat's centre (sigma = 0)
        if (pairs == 0 or stuck) and n_relu == 0 or r == max_rounds:
            return state, r, pairs + n_relu
        last = pairs
        if pairs and not stuck:
            x = state["opacity"][on_edge]
            # d ln sigmoid(x) / dx = 1 - sigmoid(x): the step that multiplies the opacity by (1 + nudge)
            state["opacity"][on_edge] = x + (nudge / (1.0 - torch.sigmoid(x)).clamp_min(1e-3)).clamp_max(0.5)
        if n_relu:
            # move the unit's pre-activation away from 0 by relu_nudge along the unit's own weights on local_feat (the smallest change that does)
            w = W1l[relu_unit[relu_edge]]
            s = torch.where(relu_sign[relu_edge] == 0, torch.ones(n_relu), relu_sign[relu_edge])
            state["local_feat"][relu_edge] += (s * relu_nudge / w.pow(2).sum(1).clamp_min(1e-12))[:, None] * w
    return state, max_rounds, pairs


def optimisation_step(state, kf, cfg, bg, is_important, *, workers=None, adjust_targets=None, knife_eps=KNIFE_EPS, knife_eps_T=KNIFE_EPS_T,
                      image_knife_tol=IMAGE_KNIFE_TOL, want_grads=True, dtype=torch.float64, knife_rows_from="image", timings=None):
    """One `optimization_step` up to (not including) the optimisers.  state / kf / cfg: `snapshot()`.  bg: the step's random background
    [3].  adjust_targets(image [3,H,W] fp64, invdepth [1,H,W] fp64, gt, mono) -> (gt, mono): lets a TEST move its own targets off the
    loss's knife edges after it has seen the oracle's render (the targets used are returned).

    dtype=torch.float32 evaluates the SAME chain in single precision (other operation order than any kernel's): the measure of what fp32
    arithmetic itself leaves of a gradient on this workload.  knife_rows_from: "image" | "raster" | "both" (which knife pixels' contributors
    `knife_rows` marks).  timings: a dict that receives the seconds of each phase.

    Returns dict: loss (float), image [3,H,W] (exposed, clamped), invdepth [1,H,W], visibility bool [N], global_visibility bool [Nvox],
    grads {15 leaves: GAUSS_KEYS, MLP_KEYS, KF_KEYS} fp64, raster_knife (any rasteriser edge, the termination edge at KNIFE_EPS_T included:
    what a pixel-level comparison masks) / skip_knife (the skip / clamp edges only) / image_knife bool [H,W], knife_rows bool [N] (Gaussians
    that contribute to a pixel of the set `knife_rows_from` names: their gradient moves if that pixel is decided the other way), gt / mono
    (the targets used), n_isects, selected."""
    W, H = cfg["width"], cfg["height"]
    workers = default_workers() if workers is None else workers
    f32, f64 = torch.float32, dtype
    _t = [time.time()]

    def lap(name):
        if timings is not None:
            now = time.time()
            timings[name] = timings.get(name, 0.0) + now - _t[0]
            _t[0] = now
    serial = _threads(SERIAL_THREADS)
    serial.__enter__()
    st32, par32, dec, K32, p32, isects = _fp32_decisions(state, kf, cfg)
    N = st32["xyz"].shape[0]
    with torch.no_grad():
        sel = dec["selection_mask"]
        visibility = torch.zeros(N, dtype=torch.bool)
        visibility[sel] = p32["radii"].max(dim=1).values > 0
        global_visibility = torch.zeros(st32["global_feat"].shape[0], dtype=torch.bool)
        global_visibility[st32["cls_id"][visibility].squeeze(-1)] = True
    lap("A fp32 decisions")
    # ---- B. the differentiable chain in fp64 on those decisions
    leaves = {k: state[k].to(f64).clone().requires_grad_(want_grads) for k in GAUSS_KEYS + MLP_KEYS}
    kfl = {k: kf[k].to(f64).clone().requires_grad_(want_grads) for k in ("rW2C", "tW2C", "exposure")}
    st64 = dict(leaves, cls_id=state["cls_id"], d_max=state["d_max"])
    with torch.set_grad_enabled(want_grads):
        Rt = get_Rt(kfl["rW2C"], kfl["tW2C"])
        par, _ = _lod_and_params(st64, Rt, dec)
        p = go.project(par["xyz"], par["rotation"], par["scaling"], par["opacity"], Rt, K32.to(f64), W, H, cfg["eps2d"],
                       force_valid=p32["valid"])
        dirs = par["xyz"] - go.camera_position(Rt)[None, :]
        dirs = torch.where(p32["valid"][:, None], dirs, torch.ones_like(dirs))
        rgb = go.sh_to_rgb(cfg["sh_degree"], dirs, par["feats"])
        feat = torch.cat([rgb, p["depths"][:, None]], -1)
    ras_in = (p["means2d"], p["conics"], feat, par["opacity"])

    lap("B projection graph")
    # ---- C. pass 1 over the tiles: the frame
    tile_w, tile_h = isects["tile_w"], isects["tile_h"]
    Hp, Wp = tile_h * TILE, tile_w * TILE
    _SH.clear()
    _SH.update(means2d32=p32["means2d"], conics32=p32["conics"], opac32=par32["opacity"])     # the per-pixel decisions are taken on these
    _SH.update(means2d=ras_in[0].detach(), conics=ras_in[1].detach(), feat=ras_in[2].detach(), opac=ras_in[3].detach(),
               flat=torch.from_numpy(isects["flatten_ids"].astype(np.int64)), offsets=isects["offsets"].reshape(-1), n_isects=isects["n_isects"],
               tile_w=tile_w, grid=torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij"), knife_eps=knife_eps,
               knife_eps_T=knife_eps_T)
    chunks = _chunks(workers)
    tmpdir = prefix = None
    if workers > 1 and len(chunks) > 1:
        tmpdir = tempfile.mkdtemp(prefix="adk_step_oracle_")
        prefix = os.path.join(tmpdir, "frame")
        _publish(prefix, _PASS1)
    render = torch.zeros(Hp, Wp, 4, dtype=f64)
    T_img = torch.ones(Hp, Wp, dtype=f64)
    knife_bits = torch.zeros(Hp, Wp, dtype=torch.uint8)
    for res in _run(_fwd_chunk, chunks, workers, prefix):
        for tid, col, T, kn in res:
            ty, tx = divmod(tid, tile_w)
            sl = (slice(ty * TILE, (ty + 1) * TILE), slice(tx * TILE, (tx + 1) * TILE))
            render[sl] = torch.from_numpy(col).reshape(TILE, TILE, 4)
            T_img[sl] = torch.from_numpy(T).reshape(TILE, TILE)
            knife_bits[sl] = torch.from_numpy(kn).reshape(TILE, TILE)
    render, T_img, knife_bits = render[:H, :W].contiguous(), T_img[:H, :W].contiguous(), knife_bits[:H, :W]
    raster_knife, skip_knife = (knife_bits & 1).bool(), (knife_bits & 2).bool()

    lap("C tile pass 1")
    # ---- D. the image-space chain (h3dgsv3.py:682-686, 611-614, 428-449)
    render_l = render.clone().requires_grad_(want_grads)
    T_l = T_img.clone().requires_grad_(want_grads)
    bg64 = bg.detach().cpu().to(f64)
    rdk = radial_decay_kernel(H, W, cfg["rad_decay"]).to(f64)

    def image_chain(gt, mono):
        color = render_l[..., :3].permute(2, 0, 1) + T_l[None] * bg64[:, None, None]       # + (1 - alpha) bg
        invdepth = (1.0 / render_l[..., 3])[None]
        raw = (kfl["exposure"][:3, :3] @ color.reshape(3, -1)) + kfl["exposure"][:3, 3, None]
        image = raw.clamp(0, 1).view(3, H, W)
        out = dict(raw=raw.view(3, H, W), image=image, invdepth=invdepth)
        img, g, inv, mn = image, gt, invdepth, mono
        if not is_important:
            error_map = rdk * (img - g).abs()
            out["outlier_error"] = error_map
            alpha_mask = ~((error_map[0] > 0.2) | (error_map[1] > 0.2) | (error_map[1] > 0.2))
            img, g, inv, mn = img * alpha_mask, g * alpha_mask, inv * alpha_mask, mn * alpha_mask
        l1 = (rdk * (img - g).abs()).mean()
        ssim_loss = 1 - ssim_oracle.fused_ssim_oracle(img[None], g[None])
        depth_loss = (rdk * (inv - mn).abs()).mean()
        out["loss_image"] = cfg["lambda_dssim"] * ssim_loss + (1 - cfg["lambda_dssim"]) * l1 + kf["depth_loss_weight"] * depth_loss
        return out

    gt, mono = kf["image"].to(f64), kf["mono_idepth"].to(f64)
    if adjust_targets is not None:
        with torch.no_grad():
            first = image_chain(gt, mono)
            gt, mono = adjust_targets(first["image"], first["invdepth"], gt, mono)
            gt, mono = gt.float().to(f64), mono.float().to(f64)      # what an fp32 keyframe can hold: both sides train on THESE values
    with torch.set_grad_enabled(want_grads):
        img = image_chain(gt, mono)
        reg = par["scaling"].prod(dim=1).mean()
        loss = img["loss_image"] + cfg["scaling_reg_factor"] * reg
    with torch.no_grad():
        raw = img["raw"]
        tol = image_knife_tol
        image_knife = (((img["image"] - gt).abs() < tol) | (raw.abs() < tol) | ((raw - 1).abs() < tol)).any(0)
        image_knife |= ((img["invdepth"] - mono).abs() < tol)[0]
        if not is_important:
            image_knife |= ((img["outlier_error"][:2] - 0.2).abs() < tol).any(0)
    out = dict(loss=float(loss.detach()), image=img["image"].detach(), invdepth=img["invdepth"].detach(), visibility=visibility,
               global_visibility=global_visibility, raster_knife=raster_knife, skip_knife=skip_knife, image_knife=image_knife, gt=gt, mono=mono,
               n_isects=isects["n_isects"], selected=sel, isects=isects, radii=p32["radii"])
    if not want_grads:
        _SH.clear()
        serial.__exit__()
        if tmpdir:
            shutil.rmtree(tmpdir, ignore_errors=True)
        return out
    g_render, g_T, g_exposure = torch.autograd.grad(img["loss_image"], [render_l, T_l, kfl["exposure"]])

    lap("D image chain")
    # ---- E. pass 2 over the tiles: per-splat gradients of the compositing under that cotangent
    def pad(t):
        full = torch.zeros((Hp, Wp) + tuple(t.shape[2:]), dtype=t.dtype)
        full[:H, :W] = t
        return full
    knife_px = {"image": image_knife, "raster": skip_knife, "both": image_knife | skip_knife}[knife_rows_from]
    _SH.update(g_col=pad(g_render), g_T=pad(g_T), knife_px=pad(knife_px))
    n_sel = ras_in[0].shape[0]
    acc = torch.zeros(n_sel, 10, dtype=f64)
    knife_sel = torch.zeros(n_sel, dtype=torch.bool)
    if prefix is not None:
        _publish(prefix, _PASS2)
    for res in _run(_bwd_chunk, chunks, workers, prefix):
        if res is None:
            continue
        uniq, a, touched = res
        acc.index_add_(0, torch.from_numpy(uniq), torch.from_numpy(a).to(f64))
        knife_sel[torch.from_numpy(touched)] = True
    _SH.clear()
    if tmpdir:
        shutil.rmtree(tmpdir, ignore_errors=True)

    lap("E tile pass 2")
    # ---- F. through projection / SH / mlp_cov / LoD / pose to the leaves
    torch.autograd.backward([ras_in[0], ras_in[1], ras_in[2], ras_in[3], reg],
                            [acc[:, 0:2], acc[:, 2:5], acc[:, 5:9], acc[:, 9], torch.tensor(cfg["scaling_reg_factor"], dtype=f64)])
    grads = {k: (leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])) for k in GAUSS_KEYS + MLP_KEYS}
    grads["kf.rW2C"], grads["kf.tW2C"], grads["kf.exposure"] = kfl["rW2C"].grad, kfl["tW2C"].grad, g_exposure
    knife_rows = torch.zeros(N, dtype=torch.bool)
    sel_idx = torch.nonzero(sel).squeeze(-1)
    knife_rows[sel_idx[knife_sel]] = True
    # a hidden unit of mlp_cov on its ReLU's switch: that Gaussian's feature gradients are one decision away from another value
    relu_sel = (dec["relu_pre"].abs().min(dim=1).values <= KNIFE_EPS_RELU) & (p32["radii"].max(dim=1).values > 0)
    knife_rows[sel_idx[relu_sel]] = True
    out.update(grads=grads, knife_rows=knife_rows, relu_knife_rows=int(relu_sel.sum()))
    lap("F backward to the leaves")
    serial.__exit__()
    return out


def move_targets_off_the_knife_edges(margin=1e-3, outlier=False, rdk=None):
    """An `adjust_targets` for tests: a target value within `margin` of the oracle's rendered value (sign of the L1 term undecidable
    at fp32) moves `2 margin` away from it, staying inside [0, 1]; same for the inverse-depth target; with outlier=True (unimportant
    frames) a target whose weighted error is within `margin` of the 0.2 threshold moves so that the pixel is clearly inside."""
    def adjust(image, invdepth, gt, mono):
        gt, mono = gt.clone(), mono.clone()
        near = (image - gt).abs() < margin
        up = image + 2 * margin
        gt[near] = torch.where(up <= 1.0, up, image - 2 * margin)[near]
        nd = (invdepth - mono).abs() < margin
        mono[nd] = (invdepth + 2 * margin)[nd]
        if outlier:
            err = rdk * (image - gt).abs()
            edge = (err - 0.2).abs() < margin
            # pull the target towards the rendered value by 3 margin / rdk: the error drops clearly below the threshold
            gt[edge] = (gt + torch.sign(image - gt) * 3 * margin / rdk)[edge]
        return gt, mono
    return adjust
