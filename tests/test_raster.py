"""Rasteriser parity: HIP (through the gsplat drop-in and the C ABI) vs oracle/gsplat_oracle.py.

Bit-exact: radii, tiles-per-Gaussian, sorted intersection keys (tile | depth bits), flatten ids,
tile offsets.  fp32 tolerance (north star: 1e-4 relative): rendered RGB+D / alpha and every
per-attribute gradient, with the gradient reference obtained by fp64 autograd over the oracle.
The rasteriser has four discontinuities per (splat, pixel): alpha vs 1/255 (skip), alpha vs 0.999 (clamp),
sigma vs 0, T(1-alpha) vs 1e-4 (terminate).  An fp32 and an fp64 evaluation may decide a pixel sitting on one of them
differently, so the gradient tests IDENTIFY those pixels (oracle `extras["knife"]`, relative margin 5e-4: MEASURED 0.7 - 3.0 % of the
pixels over every scene and window used here, and each test asserts its own measured fraction + 2 points, not a blanket allowance) and
take them out of the loss on both sides; on everything else the north-star criterion is asserted as is:
rel_l2 <= 1e-4 per attribute and max |error| <= 1e-4 max |gradient|, no outlier allowance.
"""
import numpy as np
import pytest
import torch

from oracle import gsplat_oracle as go


def _scene(N, W, H, seed, **kw):
    return go.synthetic_scene(N, W, H, seed=seed, **kw)


def _close_frac(a, b, rel=1e-4, abs_floor=None):
    """fraction of entries with |a-b| <= rel*max|b| (+floor)."""
    scale = float(b.abs().max()) if b.numel() else 1.0
    tol = rel * scale + (abs_floor or 0.0)
    return float(((a - b).abs() <= tol).double().mean()) if b.numel() else 1.0


# ----------------------------------------------------------------------------- CPU: oracle self-checks
def test_oracle_vectorised_matches_literal_loop():
    sc = _scene(300, 48, 40, seed=1, sigma_px=3.0)
    r, a, meta = go.rasterization(**sc, eps2d=0.01)
    rl, al, ll = go.rasterize_to_pixels_loop(meta["means2d"], meta["conics"], meta["colors"], sc["opacities"], 48, 40, meta["isects"])
    assert np.abs(rl - r.numpy()).max() < 1e-5
    assert np.abs(al - a.numpy()).max() < 1e-6
    assert (ll == meta["last_ids"].numpy()).all()


def test_oracle_sort_keys_are_tile_then_depth_then_id():
    sc = _scene(2000, 96, 64, seed=2)
    _, _, meta = go.rasterization(**sc, eps2d=0.01)
    ids, flat = meta["isects"]["isect_ids"], meta["isects"]["flatten_ids"]
    assert (np.diff(ids) >= 0).all()
    same = np.diff(ids) == 0
    assert (np.diff(flat)[same] > 0).all()  # ties broken by ascending Gaussian id (stable sort)
    off = meta["isects"]["offsets"].reshape(-1)
    tiles = ids >> 32
    for t in (0, 5, len(off) - 1):
        assert off[t] == np.searchsorted(tiles, t, side="left")


def test_oracle_fp64_gradients_match_finite_differences():
    sc = _scene(40, 32, 32, seed=3, sigma_px=4.0, dtype=torch.float64)
    g = torch.Generator().manual_seed(0)
    wgt = torch.randn(32, 32, 4, generator=g, dtype=torch.float64)

    def loss(means):
        s = dict(sc, means=means)
        r, a, _ = go.rasterization(**s, eps2d=0.01, grad_dtype=torch.float64)
        return (r * wgt).sum() + a.sum()

    m = sc["means"].clone().requires_grad_(True)
    loss(m).backward()
    idx = [(3, 0), (7, 2), (20, 1)]
    for i, j in idx:
        e = torch.zeros_like(m); e[i, j] = 1e-6
        fd = (loss(m.detach() + e) - loss(m.detach() - e)) / 2e-6
        assert abs(float(fd) - float(m.grad[i, j])) <= 1e-4 * max(1.0, abs(float(fd)))


# ----------------------------------------------------------------------------- GPU parity
def _run_hip(sc, dev, sh_degree=3, eps2d=0.01, render_mode="RGB+D", backgrounds=None, requires_grad=False, viewmat=None):
    from gsplat.rendering import rasterization
    t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}
    if viewmat is not None:
        t["viewmat"] = viewmat.to(dev)
    leaves = {}
    for k in ("means", "quats", "scales", "opacities", "colors", "viewmat"):
        leaves[k] = t[k].clone().requires_grad_(requires_grad)
    r, a, meta = rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                               leaves["viewmat"][None], t["K"][None], sc["width"], sc["height"], render_mode=render_mode,
                               rasterize_mode="classic", absgrad=False, packed=False, sh_degree=sh_degree, eps2d=eps2d,
                               backgrounds=backgrounds.to(dev)[None] if backgrounds is not None else None,
                               return_isect_ids=True)
    return r, a, meta, leaves


def _tilted_viewmat(seed=0):
    g = torch.Generator().manual_seed(seed)
    ax = torch.randn(3, generator=g); ax = ax / ax.norm()
    ang = 0.15
    Kx = torch.tensor([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = torch.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * (Kx @ Kx)
    V = torch.eye(4); V[:3, :3] = R; V[:3, 3] = torch.tensor([0.1, -0.05, 0.3])
    return V


@pytest.mark.gpu
@pytest.mark.parametrize("N,W,H,seed,tilt", [(5000, 160, 112, 0, False), (20000, 256, 192, 1, True),
                                             (3000, 75, 53, 2, True), (200000, 512, 384, 0, False)])
def test_binning_is_bit_exact(N, W, H, seed, tilt, dev):
    sc = _scene(N, W, H, seed)
    V = _tilted_viewmat(seed) if tilt else sc["viewmat"]
    sc = dict(sc, viewmat=V)
    _, _, ometa = go.rasterization(**sc, eps2d=0.01)
    r, a, meta, _ = _run_hip(sc, dev)
    oi = ometa["isects"]
    assert torch.equal(meta["radii"][0].cpu(), ometa["radii"])
    assert np.array_equal(meta["tiles_per_gauss"][0].cpu().numpy(), oi["tiles_per_gauss"])
    assert np.array_equal(meta["isect_ids"].cpu().numpy(), oi["isect_ids"])          # tile | depth-bits keys
    assert np.array_equal(meta["flatten_ids"].cpu().numpy(), oi["flatten_ids"])      # stable order
    assert np.array_equal(meta["isect_offsets"][0].cpu().numpy(), oi["offsets"])
    # projected quantities of the visible Gaussians are the same IEEE chain => bit-exact too
    vis = ometa["p32"]["valid"]
    assert torch.equal(meta["means2d"][0].cpu()[vis], ometa["p32"]["means2d"][vis])
    assert torch.equal(meta["depths"][0].cpu()[vis], ometa["p32"]["depths"][vis])
    assert torch.equal(meta["conics"][0].cpu()[vis], ometa["p32"]["conics"][vis])


@pytest.mark.gpu
@pytest.mark.parametrize("N,W,H,seed,sh_degree,mode", [(5000, 160, 112, 0, 3, "RGB+D"), (20000, 256, 192, 1, 3, "RGB+D"),
                                                       (3000, 75, 53, 2, 2, "RGB"), (3000, 75, 53, 3, 0, "D"),
                                                       (200000, 512, 384, 0, 3, "RGB+D")])
def test_forward_render_matches_oracle(N, W, H, seed, sh_degree, mode, dev):
    sc = dict(_scene(N, W, H, seed), viewmat=_tilted_viewmat(seed))
    ex = {}
    ro, ao, ometa = go.rasterization(**sc, eps2d=0.01, sh_degree=sh_degree, render_mode=mode, extras=ex)
    r, a, meta, _ = _run_hip(sc, dev, sh_degree=sh_degree, render_mode=mode)
    r, a = r[0].cpu(), a[0].cpu()
    assert r.shape == ro.shape and a.shape == ao.shape
    keep = (~ex["knife"])[..., None]   # every pixel that is not ON a skip / clamp / terminate threshold: 1e-4, no allowance
    assert float(keep.float().mean()) > 0.96   # measured knife fraction of these five scenes: 0.8 - 1.9 % (round 6) + 2 points
    assert float(((r - ro).abs() * keep).max()) <= 1e-4 * float(ro.abs().max())
    assert float(((a - ao).abs() * keep).max()) <= 1e-4
    # knife-edge pixels: a flipped decision moves the pixel by at most one splat's contribution
    assert float((r - ro).abs().max()) <= 2e-2 * float(ro.abs().max())


@pytest.mark.gpu
def test_forward_backgrounds_and_expected_depth(dev):
    sc = dict(_scene(1500, 96, 80, 5, sigma_px=1.5), viewmat=_tilted_viewmat(5))
    bg = torch.tensor([0.2, 0.5, 0.7])
    ro, ao, _ = go.rasterization(**sc, eps2d=0.01, render_mode="RGB", backgrounds=bg)
    r, a, _, _ = _run_hip(sc, dev, render_mode="RGB", backgrounds=bg)
    assert _close_frac(r[0].cpu(), ro, 1e-4) >= 0.999
    rd, ad, _ = go.rasterization(**sc, eps2d=0.01, render_mode="RGB+D")
    re, ae, _, _ = _run_hip(sc, dev, render_mode="RGB+ED")
    exp_depth = rd[..., 3:4] / ad.clamp(min=1e-10)
    assert _close_frac(re[0, ..., 3:4].cpu(), exp_depth, 1e-4) >= 0.999


def _assert_grad(name, gh, gref, rel=1e-4):
    """north-star criterion, no outlier allowance: rel_l2 and max error relative to the largest entry."""
    gh, gref = gh.double(), gref.double()
    nrm = float(gref.norm())
    rel_l2 = float((gh - gref).norm()) / max(nrm, 1e-300)
    rel_max = float((gh - gref).abs().max()) / max(float(gref.abs().max()), 1e-300)
    assert rel_l2 <= rel and rel_max <= rel, (name, "rel_l2", rel_l2, "rel_max", rel_max)
    return rel_l2, rel_max


@pytest.mark.gpu
@pytest.mark.parametrize("N,W,H,seed", [(1500, 96, 80, 0), (6000, 160, 128, 1), (40000, 256, 192, 2)])
def test_backward_matches_fp64_autograd_oracle(N, W, H, seed, dev):
    sc = dict(_scene(N, W, H, seed), viewmat=_tilted_viewmat(seed))
    g = torch.Generator().manual_seed(100 + seed)
    v_r = torch.randn(H, W, 4, generator=g)
    v_a = torch.randn(H, W, 1, generator=g)
    # fp64 autograd over the oracle, on the fp32 tile lists
    leaves = {k: sc[k].double().clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors", "viewmat")}
    ex = {}
    ro, ao, _ = go.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"],
                                 leaves["viewmat"], sc["K"], W, H, eps2d=0.01, grad_dtype=torch.float64, extras=ex)
    keep = (~ex["knife"])[..., None]
    assert float(keep.double().mean()) > 0.96    # measured knife fraction: 0.7 / 0.8 / 1.9 % (round 6) + 2 points
    v_r, v_a = v_r * keep, v_a * keep          # knife-edge pixels leave the loss on both sides
    ((ro * v_r.double()).sum() + (ao * v_a.double()).sum()).backward()
    # HIP
    r, a, meta, hl = _run_hip(sc, dev, requires_grad=True)
    ((r[0] * v_r.to(dev)).sum() + (a[0] * v_a.to(dev)).sum()).backward()
    scale = float(ro.detach().abs().max())
    assert float(((r[0].cpu().double() - ro.detach()).abs() * keep).max()) <= 1e-4 * scale
    assert float(((a[0].cpu().double() - ao.detach()).abs() * keep).max()) <= 1e-4
    for k in ("means", "quats", "scales", "opacities", "colors"):
        _assert_grad(k, hl[k].grad.cpu(), leaves[k].grad)
    _assert_grad("viewmat", hl["viewmat"].grad.cpu()[:3], leaves["viewmat"].grad[:3])


@pytest.mark.gpu
def test_camera_gradient_accumulator_is_left_zeroed(dev):
    """adk_project_bwd hands its 16-double cam_grad scratch back zeroed (include/artdeco_hip.h), which is what lets the
    binding keep one per stream: two backward passes in a row give the same view-matrix gradient (up to the order of the
    atomics), not the sum of both, and the cached accumulators read zero afterwards."""
    from artdeco_amd import rasterizer
    sc = dict(_scene(3000, 128, 96, 4), viewmat=_tilted_viewmat(3))
    grads = []
    for _ in range(3):
        r, a, meta, hl = _run_hip(sc, dev, requires_grad=True)
        (r[0].square().sum() + a[0].sum()).backward()
        grads.append(hl["viewmat"].grad.double().cpu())
    assert float(grads[0][:3].abs().max()) > 0
    for g in grads[1:]:
        assert float((g - grads[0]).abs().max()) <= 1e-4 * float(grads[0].abs().max())
    torch.cuda.synchronize()
    assert rasterizer._CAM_GRAD and all(float(b.abs().max()) == 0.0 for b in rasterizer._CAM_GRAD.values())


@pytest.mark.gpu
@pytest.mark.parametrize("N,W,H", [(1_000_000, 1920, 1080), (4_000_000, 2592, 1944)])
def test_binning_is_bit_exact_at_baseline_sizes(N, W, H, dev):
    """BASELINE configs[2] and configs[3] at FULL size: radii, tiles per Gaussian, the sorted 64-bit (tile | depth-bits)
    keys, the sorted Gaussian ids and the tile offsets, bit for bit against the vectorised oracle (projection + isect only;
    the oracle's compositing is not needed for the integer outputs)."""
    sc = dict(_scene(N, W, H, 0), viewmat=_tilted_viewmat(1))
    p = go.project(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["viewmat"], sc["K"], W, H, 0.01)
    oi = go.isect_tiles(p["means2d"], p["radii"], p["depths"], W, H)
    r, a, meta, _ = _run_hip(sc, dev)
    assert torch.equal(meta["radii"][0].cpu(), p["radii"])
    assert np.array_equal(meta["tiles_per_gauss"][0].cpu().numpy(), oi["tiles_per_gauss"])
    assert meta["isect_ids"].numel() == oi["n_isects"] and oi["n_isects"] > 2 * N
    assert np.array_equal(meta["isect_ids"].cpu().numpy(), oi["isect_ids"])
    assert np.array_equal(meta["flatten_ids"].cpu().numpy(), oi["flatten_ids"])
    assert np.array_equal(meta["isect_offsets"][0].cpu().numpy(), oi["offsets"])
    vis = p["valid"]
    assert torch.equal(meta["means2d"][0].cpu()[vis], p["means2d"][vis])
    assert torch.equal(meta["depths"][0].cpu()[vis], p["depths"][vis])
    assert torch.equal(meta["conics"][0].cpu()[vis], p["conics"][vis])


_NORTHSTAR_ORACLE = {}


def _northstar_oracle(N, W, H):
    """projection + isect_tiles of the oracle for one north-star-size scene, computed once per session (three routes share it)."""
    key = (N, W, H)
    if key not in _NORTHSTAR_ORACLE:
        sc = dict(_scene(N, W, H, 0), viewmat=_tilted_viewmat(1))
        p = go.project(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["viewmat"], sc["K"], W, H, 0.01)
        _NORTHSTAR_ORACLE[key] = (sc, p, go.isect_tiles(p["means2d"], p["radii"], p["depths"], W, H))
    return _NORTHSTAR_ORACLE[key]


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["bucket", "merge", "global"])
@pytest.mark.parametrize("N,W,H", [(1_000_000, 512, 384), (1_000_000, 648, 486)])
def test_binning_is_bit_exact_at_northstar_sizes_on_every_route(N, W, H, route, dev, monkeypatch):
    """The configuration the >= 30 frames/s claim is made on (BASELINE north_star: 1 M Gaussians on 512x384; run.sh trains at
    648x486, `--downsampling 2.0`): every tile list has 1 025 .. 8 192 entries there, so EVERY list goes through the sample
    sort (bin_tile_sort_bucket_kernel, default), the chunk-merge sort (bin_tile_sort_merge_kernel, ADK_BIN_BUCKET_SORT=0) or the
    global two-level radix sort (ADK_BIN_LOCAL=0).  Each route against the ORACLE (go.isect_tiles), not against another HIP route:
    radii, tiles per Gaussian, sorted 64-bit (tile | depth bits) keys, sorted ids and offsets bit for bit.
    Reference: gsplat isect_tiles + radix sort + isect_offset_encode reached from h3dgsv3.py:664-680."""
    sc, p, oi = _northstar_oracle(N, W, H)
    for k in ("ADK_BIN_BUCKET_SORT", "ADK_BIN_LOCAL"):
        monkeypatch.delenv(k, raising=False)
    if route == "merge":
        monkeypatch.setenv("ADK_BIN_BUCKET_SORT", "0")
    elif route == "global":
        monkeypatch.setenv("ADK_BIN_LOCAL", "0")
    counts = np.diff(np.append(oi["offsets"].reshape(-1), oi["n_isects"]))
    assert counts.max() <= 8192 and (counts > 1024).mean() > 0.7, (counts.max(), (counts > 1024).mean())  # the long-list sorts' range (the tilted view leaves a border of short lists)
    r, a, meta, _ = _run_hip(sc, dev)
    assert torch.equal(meta["radii"][0].cpu(), p["radii"])
    assert np.array_equal(meta["tiles_per_gauss"][0].cpu().numpy(), oi["tiles_per_gauss"])
    assert meta["isect_ids"].numel() == oi["n_isects"]
    assert np.array_equal(meta["isect_ids"].cpu().numpy(), oi["isect_ids"])
    assert np.array_equal(meta["flatten_ids"].cpu().numpy(), oi["flatten_ids"])
    assert np.array_equal(meta["isect_offsets"][0].cpu().numpy(), oi["offsets"])
    vis = p["valid"]
    assert torch.equal(meta["means2d"][0].cpu()[vis], p["means2d"][vis])
    assert torch.equal(meta["depths"][0].cpu()[vis], p["depths"][vis])
    assert torch.equal(meta["conics"][0].cpu()[vis], p["conics"][vis])


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["local", "global"])
@pytest.mark.parametrize("N,W,H,n_big,grow", [(30_000, 512, 384, 40, 40.0), (200_000, 1920, 1080, 12, 400.0), (6_000, 512, 384, 6_000, 12.0)])
def test_binning_is_bit_exact_with_screen_filling_gaussians(N, W, H, n_big, grow, route, dev, monkeypatch):
    """Round 6 (finding 62): over a long sequence the optimiser grows a handful of Gaussians until they cover most of the frame.  Their tile
    rectangles (hundreds to thousands of tiles; ALL 8 160 at 1080p) are counted / scattered / emitted by a whole workgroup or wave instead of
    by the one thread that owns the Gaussian -- same lists: radii, tiles per Gaussian, sorted keys, sorted ids and offsets against the ORACLE,
    bit for bit, on the tile-local and on the global route.  Third case: a close-up, EVERY Gaussian grown x12 (the typical rectangle is above
    the threshold, every lane of a wave hands its Gaussian to the wave)."""
    sc = dict(_scene(N, W, H, 21), viewmat=_tilted_viewmat(4))
    idx = torch.randperm(N, generator=torch.Generator().manual_seed(1))[:n_big]
    sc["scales"] = sc["scales"].clone()
    sc["scales"][idx] *= grow
    sc["opacities"] = sc["opacities"].clone()
    sc["opacities"][idx] = 0.9
    for k in ("ADK_BIN_BUCKET_SORT", "ADK_BIN_LOCAL", "ADK_BIN_LONG"):
        monkeypatch.delenv(k, raising=False)
    if route == "global":
        monkeypatch.setenv("ADK_BIN_LOCAL", "0")
    p = go.project(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["viewmat"], sc["K"], W, H, 0.01)
    oi = go.isect_tiles(p["means2d"], p["radii"], p["depths"], W, H)
    n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
    big = oi["tiles_per_gauss"][idx.numpy()]
    if n_big < N:
        assert (big > 64).sum() >= n_big // 2 and big.max() >= 0.5 * n_tiles, (big.max(), n_tiles)     # some rectangles cover most of the frame
    else:
        assert np.median(big[big > 0]) > 64, np.median(big[big > 0])                                   # the close-up: the typical one is large
    r, a, meta, _ = _run_hip(sc, dev)
    assert torch.equal(meta["radii"][0].cpu(), p["radii"])
    assert np.array_equal(meta["tiles_per_gauss"][0].cpu().numpy(), oi["tiles_per_gauss"])
    assert meta["isect_ids"].numel() == oi["n_isects"]
    assert np.array_equal(meta["isect_ids"].cpu().numpy(), oi["isect_ids"])
    assert np.array_equal(meta["flatten_ids"].cpu().numpy(), oi["flatten_ids"])
    assert np.array_equal(meta["isect_offsets"][0].cpu().numpy(), oi["offsets"])
    assert bool(torch.isfinite(r).all())


@pytest.mark.gpu
def test_binning_time_does_not_blow_up_with_screen_filling_gaussians(dev):
    """The PERFORMANCE side of finding 62, with a margin no box-to-box variation can eat: sixteen screen-filling Gaussians in a million took
    `bin_count` + `bin_scatter` from 0.10 to 1.01 ms (10x) while one thread walked each rectangle; wave-cooperative they cost +16 %
    (profiles/r06_bin_big_lab.txt).  Asserted: less than 2.5x."""
    from artdeco_amd import rasterizer
    from gsplat.rendering import rasterization
    N, W, H = 1_000_000, 1920, 1080
    times = {}
    for name, n_big in (("base", 0), ("big", 16)):
        sc = _scene(N, W, H, 0)
        if n_big:
            idx = torch.randperm(N, generator=torch.Generator().manual_seed(1))[:n_big]
            sc["scales"] = sc["scales"].clone()
            sc["scales"][idx] *= 400.0
        t = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}
        run = lambda: rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], t["viewmat"][None], t["K"][None], W, H,
                                    render_mode="RGB+D", rasterize_mode="classic", absgrad=False, packed=False, sh_degree=3, eps2d=0.01)
        for _ in range(3):
            run()
        tm = rasterizer.StageTimer(only=("bin_count", "bin_scatter"))
        rasterizer.set_stage_timer(tm)
        try:
            for _ in range(6):
                run()
        finally:
            rasterizer.set_stage_timer(None)
        sm = tm.summary_ms()
        times[name] = sm["bin_count"]["mean_ms"] + sm["bin_scatter"]["mean_ms"]
    assert times["big"] < 2.5 * times["base"], times


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["dense-northstar", "pile", "equal-depths"])
def test_binning_is_bit_exact_beyond_8192_entries_per_tile(case, dev, monkeypatch):
    """Tile lists ABOVE 8 192 entries stay on the tile-local route (round 5: bin_tile_sort_long_kernel, recursive partition on the key range
    between two buffers); gsplat's isect_tiles + radix sort have no list-length limit (h3dgsv3.py:664-680).  Against the ORACLE, bit for bit:
      dense-northstar  1 M Gaussians on 512x384 with sigma ~ 4 px (I / N ~ 10): EVERY interior tile list has 10-20 k entries;
      pile             200 k Gaussians of which 120 k sit in front of one pixel: one list of > 120 k entries (several partition levels) next
                       to medium lists (1 025 .. 8 192) in the same frame;
      equal-depths     the pile with every z rounded to 1/8: thousands of EQUAL depth keys per list, which only the ids order."""
    for k in ("ADK_BIN_BUCKET_SORT", "ADK_BIN_LOCAL", "ADK_BIN_LONG"):
        monkeypatch.delenv(k, raising=False)
    if case == "dense-northstar":
        N, W, H = 1_000_000, 512, 384
        sc = dict(_scene(N, W, H, 0, sigma_px=4.0), viewmat=_tilted_viewmat(1))
    else:
        N, W, H = 200_000, 256, 192
        sc = dict(_scene(N, W, H, 3))
        g = torch.Generator().manual_seed(9)
        z = 2.0 + 3.0 * torch.rand(120_000, generator=g)
        sc["means"][:120_000] = torch.stack([0.01 * torch.randn(120_000, generator=g), 0.01 * torch.randn(120_000, generator=g), z], -1)
        if case == "equal-depths":
            sc["means"][:, 2] = torch.round(sc["means"][:, 2] * 8) / 8
    p = go.project(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["viewmat"], sc["K"], W, H, 0.01)
    oi = go.isect_tiles(p["means2d"], p["radii"], p["depths"], W, H)
    counts = np.diff(np.append(oi["offsets"].reshape(-1), oi["n_isects"]))
    if case == "dense-northstar":
        assert (counts > 8192).mean() > 0.6 and counts.max() < 40_000, ((counts > 8192).mean(), counts.max())
    else:
        assert counts.max() > 100_000 and ((counts > 1024) & (counts <= 8192)).any(), counts.max()
        if case == "equal-depths":
            assert np.unique(p["depths"][p["valid"]].numpy()).size < 64
    r, a, meta, _ = _run_hip(sc, dev)
    assert torch.equal(meta["radii"][0].cpu(), p["radii"])
    assert np.array_equal(meta["tiles_per_gauss"][0].cpu().numpy(), oi["tiles_per_gauss"])
    assert meta["isect_ids"].numel() == oi["n_isects"]
    assert np.array_equal(meta["isect_offsets"][0].cpu().numpy(), oi["offsets"])
    assert np.array_equal(meta["flatten_ids"].cpu().numpy(), oi["flatten_ids"])
    assert np.array_equal(meta["isect_ids"].cpu().numpy(), oi["isect_ids"])
    # and the old way out of such a frame (ADK_BIN_LONG=0: global radix route) still produces the same bytes
    monkeypatch.setenv("ADK_BIN_LONG", "0")
    r2, a2, meta2, _ = _run_hip(sc, dev)
    assert torch.equal(meta2["flatten_ids"], meta["flatten_ids"]) and torch.equal(meta2["isect_offsets"], meta["isect_offsets"])
    assert torch.equal(r2, r) and torch.equal(a2, a)


def _binning_outputs(meta):
    return {k: meta[k].cpu().numpy() for k in ("isect_ids", "flatten_ids", "isect_offsets", "tiles_per_gauss")}


@pytest.mark.gpu
def test_binning_routes_agree_and_survive_a_wrong_capacity_guess(dev, monkeypatch):
    """The two binning routes (tile-local counting sort + per-tile LDS sort; global two-level radix sort) produce the same
    bytes, and the tile-local route's pre-launched scatter is correct when its capacity guess is far too small (first frame
    of a scene / a map that grew by more than 25 %: the scatter is repeated with the exact count) or far too large."""
    from artdeco_amd import rasterizer
    N, W, H = 1_000_000, 1920, 1080
    sc = dict(_scene(N, W, H, 0), viewmat=_tilted_viewmat(1))
    monkeypatch.setenv("ADK_BIN_LOCAL", "0")
    ref = _binning_outputs(_run_hip(sc, dev)[2])
    assert ref["flatten_ids"].size > (1 << 21)
    monkeypatch.setenv("ADK_BIN_LOCAL", "1")
    key = (torch.device(dev).index or 0, W, H, 16)   # the 16x16 lists these outputs are compared on
    for hint in (None, 10, 3 * ref["flatten_ids"].size):
        rasterizer._CAPACITY_HINT.pop(key, None)
        if hint is not None:
            rasterizer._CAPACITY_HINT[key] = hint
        got = _binning_outputs(_run_hip(sc, dev)[2])
        for k in ref:
            assert np.array_equal(got[k], ref[k]), (hint, k)
        assert rasterizer._CAPACITY_HINT[key] == ref["flatten_ids"].size


@pytest.mark.gpu
@pytest.mark.parametrize("N,W,H", [(1_000_000, 512, 384), (150_000, 200, 150), (18_000, 96, 64), (8_000, 96, 64)])
def test_binning_routes_agree_on_long_tile_lists(N, W, H, dev, monkeypatch):
    """Tile lists of 1025 .. 8192 entries (1 M Gaussians on the north-star 512x384 frame: every tile) go through the
    register-sorted-chunks + LDS-exchange merge sort of the tile-local route: the same bytes as the global radix route, for
    list lengths that are not powers of two and for 2, 4 and 8 chunks per tile."""
    sc = dict(_scene(N, W, H, 7), viewmat=_tilted_viewmat(2))
    monkeypatch.setenv("ADK_BIN_LOCAL", "0")
    ref = _binning_outputs(_run_hip(sc, dev)[2])
    monkeypatch.setenv("ADK_BIN_LOCAL", "1")
    got = _binning_outputs(_run_hip(sc, dev)[2])
    counts = np.diff(np.append(ref["isect_offsets"].reshape(-1), ref["flatten_ids"].size))
    assert counts.max() > 1024 and counts.max() <= 8192, counts.max()   # the merge sort's range (otherwise this test tests nothing)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k


@pytest.mark.gpu
def test_binning_tile_list_longer_than_the_lds_sort(dev):
    """More than 8192 splats on ONE tile (the limit of the in-LDS / in-register sorts): until round 4 the forward fell back to the global
    route for that frame, since round 5 the tile goes through bin_tile_sort_long_kernel; lists and render match the oracle either way."""
    N, W, H = 9000, 64, 48
    g = torch.Generator().manual_seed(3)
    sc = _scene(N, W, H, 4)
    means = sc["means"].clone()
    means[:, 0] = 0.02 * torch.randn(N, generator=g)      # everything lands around the principal point
    means[:, 1] = 0.02 * torch.randn(N, generator=g)
    sc = dict(sc, means=means, opacities=torch.full((N,), 0.02))
    ro, ao, ometa = go.rasterization(**sc, eps2d=0.01)
    counts = np.diff(np.append(ometa["isects"]["offsets"].reshape(-1), ometa["isects"]["n_isects"]))
    assert counts.max() > 8192
    r, a, meta, _ = _run_hip(sc, dev)
    assert np.array_equal(meta["flatten_ids"].cpu().numpy(), ometa["isects"]["flatten_ids"])
    assert np.array_equal(meta["isect_offsets"][0].cpu().numpy(), ometa["isects"]["offsets"])
    assert float((r[0].cpu() - ro).abs().max()) <= 2e-3 * float(ro.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("N,W,H,window,tilt", [(200_000, 512, 384, (10, 8, 18, 14), 2), (1_000_000, 1920, 1080, (50, 30, 56, 34), 2),
                                               (1_000_000, 1920, 1080, (0, 0, 4, 3), 2),
                                               (4_000_000, 2592, 1944, (80, 119, 84, 122), None),   # ragged bottom row of tiles
                                               (4_000_000, 2592, 1944, (40, 50, 43, 53), 2)])
def test_render_and_backward_at_baseline_sizes(N, W, H, window, tilt, dev):
    """Forward render and every per-attribute gradient at the BASELINE sizes (configs[1] 200 k / 512x384, configs[2]
    1 M / 1080p, configs[3] 4 M / 2592x1944; interior, corner and ragged-edge windows) against fp64 autograd over the oracle.
    The HIP path renders and back-propagates the WHOLE frame; the loss is supported on a window of tiles so that the fp64
    reference (restricted to the Gaussians on those tiles' lists; everything else has an exactly zero gradient) stays
    tractable."""
    sc = _scene(N, W, H, 0)
    if tilt is not None:
        sc = dict(sc, viewmat=_tilted_viewmat(tilt))
    o = go.rasterization_window(sc, window)
    assert len(o["ids"]) > 500
    tx0, ty0, tx1, ty1 = window
    ys, xs = slice(ty0 * 16, min(ty1 * 16, H)), slice(tx0 * 16, min(tx1 * 16, W))
    g = torch.Generator().manual_seed(5)
    keep = torch.zeros(H, W, 1, dtype=torch.bool)
    keep[ys, xs] = ~o["extras"]["knife"][ys, xs, None]
    assert float(keep[ys, xs].double().mean()) > 0.95   # measured knife fraction of the five windows: 1.7 - 2.7 % (round 6) + 2 points
    v_r = torch.randn(H, W, 4, generator=g) * keep
    v_a = torch.randn(H, W, 1, generator=g) * keep
    ((o["render"] * v_r.double()).sum() + (o["alphas"] * v_a.double()).sum()).backward()
    r, a, meta, hl = _run_hip(sc, dev, requires_grad=True)
    ((r[0] * v_r.to(dev)).sum() + (a[0] * v_a.to(dev)).sum()).backward()
    # forward: every pixel of the window that is not on a knife edge
    ro = o["render"].detach()
    assert float(((r[0].cpu().double() - ro).abs() * keep).max()) <= 1e-4 * float(ro.abs().max())
    assert float(((a[0].cpu().double() - o["alphas"].detach()).abs() * keep).max()) <= 1e-4
    ids = o["ids"]
    rest = torch.ones(N, dtype=torch.bool); rest[ids] = False
    for k in ("means", "quats", "scales", "opacities", "colors"):
        gh = hl[k].grad.cpu()
        _assert_grad(k, gh[ids], o["leaves"][k].grad)
        assert float(gh[rest].abs().max()) == 0.0, k   # Gaussians off the window's lists: exactly zero
    _assert_grad("viewmat", hl["viewmat"].grad.cpu()[:3], o["leaves"]["viewmat"].grad[:3])


_WINDOW_ORACLE = {}


def _window_oracle(N, W, H, window, tilt):
    """fp64 autograd over the oracle on a window of tiles, once per session: (scene, oracle dict with .grad filled, v_r, v_a, keep)."""
    key = (N, W, H, window, tilt)
    if key not in _WINDOW_ORACLE:
        sc = _scene(N, W, H, 0)
        if tilt is not None:
            sc = dict(sc, viewmat=_tilted_viewmat(tilt))
        o = go.rasterization_window(sc, window)
        tx0, ty0, tx1, ty1 = window
        ys, xs = slice(ty0 * 16, min(ty1 * 16, H)), slice(tx0 * 16, min(tx1 * 16, W))
        g = torch.Generator().manual_seed(5)
        keep = torch.zeros(H, W, 1, dtype=torch.bool)
        keep[ys, xs] = ~o["extras"]["knife"][ys, xs, None]
        v_r = torch.randn(H, W, 4, generator=g) * keep
        v_a = torch.randn(H, W, 1, generator=g) * keep
        ((o["render"] * v_r.double()).sum() + (o["alphas"] * v_a.double()).sum()).backward()
        _WINDOW_ORACLE[key] = (sc, o, v_r, v_a, keep, float(keep[ys, xs].double().mean()))
    return _WINDOW_ORACLE[key]


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["0", "2", "1"])   # one wave per 16x16 list tile | per 16x8 half | per 8x8 quadrant
@pytest.mark.parametrize("N,W,H,window,tilt", [(1_000_000, 512, 384, (10, 8, 13, 10), 2),      # interior
                                               (1_000_000, 512, 384, (30, 22, 32, 24), None),   # bottom-right corner (untilted: the tilted view leaves it empty)
                                               (1_000_000, 648, 486, (18, 12, 21, 14), 2),      # interior (run.sh geometry)
                                               (1_000_000, 648, 486, (39, 29, 41, 31), None)])  # ragged right column AND bottom row
def test_render_and_backward_at_northstar_sizes_in_every_wave_form(N, W, H, window, tilt, form, dev, monkeypatch):
    """The north-star configuration (1 M Gaussians on 512x384) and run.sh's training geometry (648x486): every tile list has
    thousands of entries and the frame is rasterised by the four-waves-per-tile kernels (raster_fwd/bwd_kernel<1,1,SUB>).  Each of the
    three forms (ADK_RASTER_SPLIT_FWD/BWD = 0 tile, 2 halves, 1 quadrants) against fp64 AUTOGRAD OVER THE ORACLE -- not against
    each other: forward on every non-knife pixel of the window and every per-attribute gradient at the north-star 1e-4.
    Reference: gsplat rasterize_to_pixels fwd/bwd reached from h3dgsv3.py:664-680; run.sh:15 (--downsampling 2.0)."""
    sc, o, v_r, v_a, keep, keep_frac = _window_oracle(N, W, H, window, tilt)
    assert len(o["ids"]) > 500 and keep_frac > 0.95    # measured knife fraction of the four windows: 2.1 - 3.0 % (round 6) + 2 points
    monkeypatch.setenv("ADK_RASTER_SPLIT_FWD", form)
    monkeypatch.setenv("ADK_RASTER_SPLIT_BWD", form)
    r, a, meta, hl = _run_hip(sc, dev, requires_grad=True)
    ((r[0] * v_r.to(dev)).sum() + (a[0] * v_a.to(dev)).sum()).backward()
    ro = o["render"].detach()
    assert float(((r[0].cpu().double() - ro).abs() * keep).max()) <= 1e-4 * float(ro.abs().max())
    assert float(((a[0].cpu().double() - o["alphas"].detach()).abs() * keep).max()) <= 1e-4
    ids = o["ids"]
    rest = torch.ones(N, dtype=torch.bool); rest[ids] = False
    for k in ("means", "quats", "scales", "opacities", "colors"):
        gh = hl[k].grad.cpu()
        _assert_grad(k, gh[ids], o["leaves"][k].grad)
        assert float(gh[rest].abs().max()) == 0.0, k
    _assert_grad("viewmat", hl["viewmat"].grad.cpu()[:3], o["leaves"]["viewmat"].grad[:3])


@pytest.mark.gpu
def test_edge_cases_empty_and_all_culled(dev):
    from gsplat.rendering import rasterization
    K = torch.tensor([[50.0, 0, 32], [0, 50.0, 24], [0, 0, 1]], device=dev)[None]
    V = torch.eye(4, device=dev)[None]
    # N = 0
    r, a, meta = rasterization(torch.zeros(0, 3, device=dev), torch.zeros(0, 4, device=dev), torch.zeros(0, 3, device=dev),
                               torch.zeros(0, device=dev), torch.zeros(0, 16, 3, device=dev), V, K, 64, 48,
                               render_mode="RGB+D", packed=False, sh_degree=3, eps2d=0.01)
    assert r.shape == (1, 48, 64, 4) and float(r.abs().max()) == 0 and float(a.abs().max()) == 0
    # everything behind the camera / transparent -> nothing rendered, zero gradients, radii 0
    means = torch.tensor([[0, 0, -2.0], [0, 0, 3.0]], device=dev, requires_grad=True)
    quats = torch.tensor([[1.0, 0, 0, 0]] * 2, device=dev)
    scales = torch.full((2, 3), 0.05, device=dev)
    opac = torch.tensor([0.9, 0.001], device=dev)
    r, a, meta = rasterization(means, quats, scales, opac, torch.rand(2, 16, 3, device=dev), V, K, 64, 48,
                               render_mode="RGB+D", packed=False, sh_degree=3, eps2d=0.01)
    assert int(meta["radii"].abs().sum()) == 0 and float(a.max()) == 0
    r.sum().backward()
    assert float(means.grad.abs().max()) == 0


@pytest.mark.gpu
def test_full_size_properties_1M_1080p(dev):
    """BASELINE config 3 (1M Gaussians, 1920x1080): size-independent properties.
    sortedness of (tile, depth) keys, offsets consistent with keys, alpha in [0,1], colours
    finite, gradient of sum(alpha) w.r.t. opacity non-negative (more opaque never lowers coverage)."""
    sc = _scene(1_000_000, 1920, 1080, 0)
    r, a, meta, hl = _run_hip(sc, dev, requires_grad=True)
    ids = meta["isect_ids"]
    assert bool((ids[1:] >= ids[:-1]).all())
    tiles = (ids >> 32).to(torch.int32)
    off = meta["isect_offsets"][0].reshape(-1)
    probe = torch.randint(0, off.numel(), (2000,), device=dev)
    starts = off[probe].long()
    ok = (starts == ids.numel()) | (tiles[starts.clamp(max=ids.numel() - 1)] >= probe.int())
    assert bool(ok.all())
    assert int(meta["tiles_per_gauss"].sum()) == ids.numel()
    assert float(a.min()) >= 0 and float(a.max()) <= 1 and bool(torch.isfinite(r).all())
    a.sum().backward()
    assert float(hl["opacities"].grad.min()) >= -1e-3


@pytest.mark.gpu
def test_gaussian_rasterizer_adapter(dev):
    """diff_gaussian_rasterization.GaussianRasterizer (webviewer/scene_models.py:559-605) over the same
    kernels: colour/alpha agree with the oracle at eps2d=0.3, invdepth is the composited 1/z, mainGaussID
    is the arg-max contributor, radii = max(rx, ry); transposed view-matrix convention."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    W, H, N = 128, 96, 4000
    sc = dict(_scene(N, W, H, 4), viewmat=_tilted_viewmat(4))
    fx = float(sc["K"][0, 0])
    bg = torch.tensor([0.1, 0.2, 0.3])
    st = GaussianRasterizationSettings(H, W, W / (2 * fx), H / (2 * fx), bg.to(dev), 1.0, torch.eye(4, device=dev), 3,
                                       torch.zeros(3, device=dev), False, False)
    t = {k: v.to(dev) for k, v in sc.items() if torch.is_tensor(v)}
    means = t["means"].clone().requires_grad_(True)
    color, invdepth, main_id, radii = GaussianRasterizer(st)(
        means, torch.zeros_like(t["means"]), t["opacities"][:, None], t["colors"][:, :1], t["colors"][:, 1:],
        t["scales"], t["quats"], t["viewmat"].transpose(0, 1))
    assert color.shape == (3, H, W) and invdepth.shape == (1, H, W) and main_id.shape == (1, H, W) and radii.shape == (N,)
    assert main_id.dtype == torch.int32
    ro, ao, ometa = go.rasterization(**sc, eps2d=0.3, render_mode="RGB", backgrounds=bg)
    assert _close_frac(color.detach().cpu().permute(1, 2, 0), ro, 1e-4) >= 0.999
    assert torch.equal(radii.cpu(), ometa["radii"].max(dim=1).values)
    # invdepth: composite 1/z with the oracle's compositing stage
    p = ometa["p32"]
    inv = torch.where(p["valid"], 1.0 / p["depths"].clamp(min=1e-9), torch.zeros_like(p["depths"]))
    rid, _, _ = go.rasterize_to_pixels(p["means2d"], p["conics"], inv[:, None], sc["opacities"], W, H, ometa["isects"])
    assert _close_frac(invdepth.detach().cpu()[0], rid[..., 0], 1e-4) >= 0.999
    # mainGaussID == argmax over the pixel's list of alpha*T (the oracle's compositing weights), -1 where nothing was
    # composited; pixels whose two largest weights agree to 1e-5 relative (an fp32 tie) are the only ones left out
    ex = {}
    go.rasterize_to_pixels(p["means2d"], p["conics"], inv[:, None], sc["opacities"], W, H, ometa["isects"], extras=ex)
    mid = main_id[0].cpu()
    assert bool(((mid == -1) == (ao[..., 0] == 0)).all())
    clear = ((ex["main_w"] - ex["second_w"]) > 1e-5 * ex["main_w"]) | (ex["main_w"] == 0)
    assert float(clear.float().mean()) > 0.999
    assert bool((mid[clear] == ex["main_ids"][clear]).all())
    assert bool((mid[ao[..., 0] == 0] == -1).all()) and int(mid.max()) < N
    (color.sum() + invdepth.sum()).backward()
    assert bool(torch.isfinite(means.grad).all()) and float(means.grad.abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("N,W,H,seed", [(20_000, 250, 187, 0), (300_000, 1000, 530, 1), (1_000_000, 1920, 1080, 2), (1_000_000, 512, 384, 3)])
def test_wide_internal_tiles_composite_the_same_pixels(N, W, H, seed, dev, monkeypatch):
    """Round 3: one wave can rasterise a 32x16 INTERNAL tile (ADK_TILE_SHAPE=32x16) instead of gsplat's 16x16.  Every pixel must
    composite the same splats in the same order: the forward outputs are BIT-IDENTICAL to the 16x16 form (ragged right / bottom
    edges, odd tile counts and the long-list sort path included), the gradients differ only by the order of their sums, the
    lists handed out as gsplat's are the same 16x16 lists, and the wide form lists fewer (splat, tile) pairs."""
    from artdeco_amd import rasterizer
    sc = dict(_scene(N, W, H, seed), viewmat=_tilted_viewmat(seed))
    g = torch.Generator().manual_seed(seed)
    wgt = torch.randn(H, W, 4, generator=g).to(dev)
    res = {}
    for shape in ("16x16", "32x16"):
        monkeypatch.setenv("ADK_TILE_SHAPE", shape)
        r, a, meta, leaves = _run_hip(sc, dev, requires_grad=True)
        res[shape] = dict(r=r.detach().clone(), a=a.detach().clone(), I=rasterizer.LAST_STATS["I"], tile_px=rasterizer.LAST_STATS["tile_px"],
                          flat=meta["flatten_ids"].clone(), off=meta["isect_offsets"].clone(), ids=meta["isect_ids"].clone())
        ((r[0] * wgt).sum() + a.sum()).backward()
        res[shape]["grads"] = {k: v.grad.detach().clone() for k, v in leaves.items()}
    a16, a32 = res["16x16"], res["32x16"]
    # (1 M / 512x384: the 32x16 lists exceed 8 192 entries.  Until round 4 that frame went to the global route and its 16x16 lists; since
    # round 5 the long-list sort keeps it on the tile-local route, wide tiles included)
    assert a16["tile_px"] == (16, 16) and a32["tile_px"] == (32, 16)
    assert torch.equal(a16["r"], a32["r"]) and torch.equal(a16["a"], a32["a"])
    assert torch.equal(a16["flat"], a32["flat"]) and torch.equal(a16["off"], a32["off"]) and torch.equal(a16["ids"], a32["ids"])
    assert a32["I"] < 0.9 * a16["I"]
    for k, g16 in a16["grads"].items():
        g32 = a32["grads"][k]
        rel = float((g32 - g16).norm() / g16.norm().clamp_min(1e-30))
        assert rel <= 2e-5, (k, rel)


@pytest.mark.gpu
@pytest.mark.parametrize("N,W,H,seed", [(20_000, 250, 187, 0), (300_000, 1000, 530, 1), (1_000_000, 1920, 1080, 2), (1_000_000, 512, 384, 3),
                                         (5_000, 41, 23, 4)])
def test_waves_per_tile_forms_composite_the_same_pixels(N, W, H, seed, dev, monkeypatch):
    """Round 3: a 16x16 list tile is served by one wave, by two (its 16x8 halves) or by four (its 8x8 quadrants), chosen from the frame's
    tile count (raster_tiles.hip, split_parts).  Every form composites the same splats in the same order on every pixel: forward
    outputs and the last / main contributor ids are BIT-IDENTICAL (ragged edges where a half or a quadrant lies outside the image
    included), the gradients differ only by the order of their sums."""
    sc = dict(_scene(N, W, H, seed), viewmat=_tilted_viewmat(seed))
    g = torch.Generator().manual_seed(seed)
    wgt = torch.randn(H, W, 4, generator=g).to(dev)
    res = {}
    for form in ("0", "2", "1", None):   # tile, halves, quadrants, the default choice
        for k in ("ADK_RASTER_SPLIT_FWD", "ADK_RASTER_SPLIT_BWD"):
            monkeypatch.delenv(k, raising=False) if form is None else monkeypatch.setenv(k, form)
        r, a, meta, leaves = _run_hip(sc, dev, requires_grad=True)
        ((r[0] * wgt).sum() + a.sum()).backward()
        res[form] = dict(r=r.detach().clone(), a=a.detach().clone(), grads={k: v.grad.detach().clone() for k, v in leaves.items()})
    ref = res["0"]
    assert float(ref["a"].max()) > 0.5
    for form in ("2", "1", None):
        assert torch.equal(ref["r"], res[form]["r"]) and torch.equal(ref["a"], res[form]["a"]), form
        for k, g0 in ref["grads"].items():
            rel = float((res[form]["grads"][k] - g0).norm() / g0.norm().clamp_min(1e-30))
            assert rel <= 2e-5, (form, k, rel)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", [(96, 64), (250, 187)])
def test_waves_per_tile_forms_report_the_same_main_gaussian(W, H, dev, monkeypatch):
    """The GaussianRasterizer adapter's mainGaussID (MAIN_ID kernels) in the three forms."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    N = 8_000
    sc = dict(_scene(N, W, H, 5), viewmat=_tilted_viewmat(5))
    fx = float(sc["K"][0, 0])
    st = GaussianRasterizationSettings(H, W, W / (2 * fx), H / (2 * fx), torch.tensor([0.1, 0.2, 0.3], device=dev), 1.0, torch.eye(4, device=dev), 3,
                                       torch.zeros(3, device=dev), False, False)
    t = {k: v.to(dev) for k, v in sc.items() if torch.is_tensor(v)}
    outs = []
    for form in ("0", "2", "1"):
        monkeypatch.setenv("ADK_RASTER_SPLIT_FWD", form)
        out = GaussianRasterizer(st)(t["means"], torch.zeros_like(t["means"]), t["opacities"][:, None], t["colors"][:, :1], t["colors"][:, 1:],
                                     t["scales"], t["quats"], t["viewmat"].transpose(0, 1))
        outs.append([o.detach().clone() for o in out])
    assert int((outs[0][2] >= 0).sum()) > 0.5 * W * H
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            assert torch.equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["uniform", "shell"])
def test_long_tile_lists_bucket_sort_equals_merge_sort_and_the_global_route(case, dev, monkeypatch):
    """Tile lists of 1 025 .. 6 400 entries (1 M Gaussians on a 512x384 frame: every tile) are split into 8 key-ordered buckets by
    sampled splitters and one wave sorts each bucket in registers (bin_tile_sort_bucket_kernel); longer lists, and tiles whose bucket
    overflows all the same, are left to the chunk-merge sort.  Every way -- buckets (default), buckets only below 3 000 keys (so that
    one frame takes BOTH paths), merge only (ADK_BIN_BUCKET_SORT=0), the global radix route (ADK_BIN_LOCAL=0) -- must produce the same
    bytes.  "shell": 85 % of the cloud inside a 1 %-thick depth shell plus a sparse far tail (what broke equal-width depth buckets)."""
    N, W, H = 1_000_000, 512, 384
    sc = dict(_scene(N, W, H, 7), viewmat=_tilted_viewmat(2))
    if case == "shell":
        g = torch.Generator().manual_seed(11)
        means = sc["means"].clone()
        V = torch.as_tensor(sc["viewmat"], dtype=torch.float32)
        cam = means @ V[:3, :3].T + V[:3, 3]
        shell = torch.rand(N, generator=g) < 0.85
        z_new = 3.0 * (1.0 + 0.01 * torch.rand(N, generator=g))
        cam[shell] = cam[shell] * (z_new[shell] / cam[shell, 2].clamp_min(1e-3))[:, None]      # same pixel, depth moved into the shell
        far = (~shell) & (torch.rand(N, generator=g) < 0.3)
        cam[far] = cam[far] * 6.0                                                               # a sparse far tail that stretches the range
        sc["means"] = ((cam - V[:3, 3]) @ V[:3, :3]).contiguous()
    outs = {}
    for name, env in (("buckets", {}), ("both", {"ADK_BIN_BUCKET_SORT": "3000"}), ("merge", {"ADK_BIN_BUCKET_SORT": "0"}), ("global", {"ADK_BIN_LOCAL": "0"})):
        for k in ("ADK_BIN_BUCKET_SORT", "ADK_BIN_LOCAL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        outs[name] = _binning_outputs(_run_hip(sc, dev)[2])
    counts = np.diff(np.append(outs["global"]["isect_offsets"].reshape(-1), outs["global"]["flatten_ids"].size))
    assert counts.max() <= 8192 and (counts > 1024).mean() > 0.5, (counts.max(), (counts > 1024).mean())
    assert ((counts > 1024) & (counts <= 3000)).any() and (counts > 3000).any()    # "both" really takes both paths
    for name in ("buckets", "both", "merge"):
        for k, ref in outs["global"].items():
            assert np.array_equal(outs[name][k], ref), (case, name, k)
