"""The DEFAULT product path of one optimisation step -- `fused.patch_scene_model` + `adk_mapper_step` (one native call: LoD cull / fade,
mlp_cov, projection + SH, binning, compositing, exposure, loss, and all of their backwards) -- against `oracle/step_oracle.py`: an fp64
autograd restatement of the whole of `SceneModel.optimization_step` (h3dgsv3.py:401-469, :595-700), at the BASELINE sizes.

North-star criterion on every one of the 15 gradient leaves: rel_l2 <= 1e-4 over ALL rows and max error <= 1e-4 of the largest entry.

Knife edges, and what is done about each (measured first, DESIGN round-5 findings):
 * the loss's own (sign(image - target) of the L1 terms, the 0.2 outlier threshold; finding 30): the test moves ITS OWN TARGETS off them --
   after the oracle's forward pass a target within 1e-3 of the rendered value is moved 2e-3 away (`adjust_targets`), both sides train on those;
 * the rasteriser's (alpha >= 1/255, T (1 - alpha) <= 1e-4): the oracle takes these per-pixel decisions from an fp32 evaluation of the fp32
   projection outputs -- the values an fp32 rasteriser decides on, bit-identical to the HIP projection's -- and differentiates in fp64 ON them
   (with fp64 decisions, an independent fp32 torch evaluation of the same step already differs from fp64 by 2e-4 / 2e-3 at 1 M / 512x384:
   exactly what the HIP path showed).  Two fp32 evaluations still disagree where a value sits within ~1e-6 of a threshold, and ONE such splat
   moves the summed leaves by 1e-4: so the test also moves ITS OWN SCENE off those edges first (`settle_scene`: the opacity of every Gaussian
   with an alpha within 1e-4 of 1/255 at some pixel -- ~1 % of them -- is raised by 2e-3, rescanned until none is left).  The handful that
   opacity cannot move (a pixel centre on a splat centre) and the clamp edges of the loss are identified by the ORACLE, and the Gaussians
   blended on those pixels (< 1 %, fraction asserted) leave the MAX-error criterion of the per-Gaussian leaves only.

CPU part (`-m "not gpu"`): the oracle against the harness mirror run on the fp32 CPU oracles (the chain the reference's real class is
pinned to in tests/test_reference_scene_model.py).
"""
import time

import pytest
import torch


def _perturb(sc, seed, lod_dmax):
    """Non-trivial features / mlp_cov so that the MLP path and its gradients carry signal (as tests/test_fused_glue.py:_scene)."""
    N = sc.xyz.shape[0]
    dev = sc.device
    g = torch.Generator().manual_seed(seed + 5)
    with torch.no_grad():
        sc.gaussian_params["local_feat"]["val"].copy_(0.5 * torch.randn(N, 16, generator=g))
        sc.gaussian_params["global_feat"]["val"].copy_(0.5 * torch.randn(sc.global_feat.shape[0], 16, generator=g))
        for p in sc.mlp_cov.parameters():
            p.add_(0.2 * torch.randn(p.shape, generator=g).to(dev))
        if lod_dmax:
            sc.gaussian_params["d_max"]["val"].copy_((1.5 + 2.0 * torch.rand(N, 1, generator=g)).to(dev))
        for kf in sc.keyframes:       # an exposure that is not the identity, a pose that is not axis-aligned
            kf.exposure.add_(0.03 * torch.randn(3, 4, generator=g).to(dev))
            kf.rW2C.add_(0.01 * torch.randn(3, 2, generator=g).to(dev))
    return sc


def _rel(x, y):
    x, y = x.double().cpu(), y.double().cpu()
    return float((x - y).norm() / (y.norm() + 1e-300)), float((x - y).abs().max() / (y.abs().max() + 1e-300))


def _compare(got, o, tol=1e-4, label="", max_knife_rows=0.01, tol_max=None, tol_pose=None):
    """rel_l2 over ALL rows and the max error (relative to the largest entry) of every leaf; for the per-Gaussian leaves the max error is
    taken over the rows the oracle does not place under a knife pixel (`knife_rows`: at most `max_knife_rows` of them).  tol_max / tol_pose:
    other bounds for the max error / the keyframe's pose (default: tol)."""
    from oracle import step_oracle as SO
    N = o["knife_rows"].shape[0]
    keep = ~o["knife_rows"]
    frac = float(o["knife_rows"].float().mean())
    worst = {}
    for k in SO.GAUSS_KEYS + SO.MLP_KEYS + SO.KF_KEYS:
        assert k in got, k
        x, y = got[k].double().cpu(), o["grads"][k]
        assert tuple(x.shape) == tuple(y.shape), (k, x.shape, y.shape)
        assert float(y.abs().max()) > 0, k
        rl2, rmax_all = _rel(x, y)
        rmax = rmax_all
        if k in SO.GAUSS_KEYS and k != "global_feat" and x.shape[0] == N:
            rmax = float((x - y)[keep].abs().max() / (y.abs().max() + 1e-300))
        worst[k] = (rl2, rmax, rmax_all)
    print(f"[step-oracle {label}] knife rows {frac:.5f};  rel_l2 / max (non-knife rows) / max (all rows):  "
          + "  ".join(f"{k} {a:.1e}/{b:.1e}/{c:.1e}" for k, (a, b, c) in worst.items()))
    assert frac <= max_knife_rows, (label, frac)
    for k, (rl2, rmax, _) in worst.items():
        t_l2 = tol_pose if (tol_pose is not None and k in ("kf.rW2C", "kf.tW2C")) else tol
        t_mx = tol_pose if (tol_pose is not None and k in ("kf.rW2C", "kf.tW2C")) else (tol if tol_max is None else tol_max)
        assert rl2 <= t_l2 and rmax <= t_mx, (label, k, rl2, rmax)
    return worst


# --------------------------------------------------------------------------------------------------------------- CPU: oracle vs the fp32 mirror
@pytest.mark.parametrize("important", [True, False])
def test_step_oracle_matches_the_fp32_mirror_on_cpu(important):
    from harness import psnr_proxy as PP
    from oracle import step_oracle as SO
    m = PP.cpu_mapper()
    sc = _perturb(m.build_synthetic_mapper(3000, 160, 112, "cpu", seed=3, n_keyframes=2), 3, True)
    sc.scaling_reg_factor = 0.05
    kid = 1
    kf = sc.keyframes[kid]
    state, kfd, cfg = SO.snapshot(sc, kid)
    torch.manual_seed(5)
    bg = torch.rand(3)
    o = SO.optimisation_step(state, kfd, cfg, bg, important, workers=1)
    got = {}

    def spy(*a, **k):
        got.update({kk: sc.gaussian_params[kk]["val"].grad.clone() for kk in SO.GAUSS_KEYS})
        got.update({"mlp." + n: p.grad.clone() for n, p in sc.mlp_cov.named_parameters()})
        got.update({"kf." + n: getattr(kf, n).grad.clone() for n in ("rW2C", "tW2C", "exposure")})
        got["vis"], got["gvis"] = a[0].clone(), a[2].clone()
    sc.optimizer.step = spy
    kf.optimizer.step = lambda: None
    torch.manual_seed(5)
    loss = sc.optimization_step(kid, is_important=important)
    assert abs(float(loss) - o["loss"]) <= 1e-6 * abs(o["loss"])
    assert torch.equal(got["vis"], o["visibility"]) and torch.equal(got["gvis"], o["global_visibility"])
    assert 0 < int(o["selected"].sum()) < 3000 and 0 < int(o["visibility"].sum())
    _compare(got, o, tol=2e-5, label=f"cpu mirror important={important}", max_knife_rows=0.05)
    assert float((kf.latest_invdepth.double() - o["invdepth"]).abs().max()) <= 1e-5 * float(o["invdepth"].abs().max())


def test_adjust_targets_moves_only_the_knife_pixels():
    from oracle import step_oracle as SO
    g = torch.Generator().manual_seed(0)
    img = torch.rand(3, 8, 8, generator=g).double()
    gt = torch.where(img < 0.5, img + 0.3, img - 0.3)
    gt[0, 0, 0] = img[0, 0, 0] + 1e-5
    gt[1, 2, 3] = img[1, 2, 3] - 2e-4
    inv = torch.rand(1, 8, 8, generator=g).double() + 0.2
    mono = inv + 0.1
    mono[0, 4, 4] = inv[0, 4, 4] - 1e-6
    g2, m2 = SO.move_targets_off_the_knife_edges(1e-3)(img, inv, gt, mono)
    moved = (g2 != gt)
    assert int(moved.sum()) == 2 and bool(moved[0, 0, 0]) and bool(moved[1, 2, 3])
    assert float((g2 - img).abs().min()) >= 1e-3 and int((m2 != mono).sum()) == 1 and float((m2 - inv).abs().min()) >= 1e-3
    assert float(g2.min()) >= 0 and float(g2.max()) <= 1


def test_settle_scene_moves_the_scene_off_the_skip_edges():
    """settle_scene: after it, no (splat, pixel) pair of the view has an alpha within 1e-4 of 1/255 (but for the handful a pixel centre on a
    splat centre leaves), only opacities moved, each by ~2e-3 relative, and the scan it is built on agrees with the oracle's own knife mask."""
    from harness import mapper
    from oracle import step_oracle as SO
    sc = _perturb(mapper.build_synthetic_mapper(6000, 160, 112, "cpu", seed=5, n_keyframes=2), 5, False)
    state, kfd, cfg = SO.snapshot(sc, 1)
    on_edge, pairs, pixels, (relu_edge, relu_unit, relu_sign) = SO.knife_scan(state, kfd, cfg, workers=1, knife_eps=1e-4, knife_eps_T=0.0, relu_eps=1e-3)
    assert pairs >= pixels > 0 and int(on_edge.sum()) > 0
    assert int(relu_edge.sum()) > 0 and bool((relu_unit[relu_edge] >= 0).all()) and bool((relu_unit[~relu_edge] == -1).all())
    log = []
    settled, rounds, left = SO.settle_scene(state, kfd, cfg, workers=1, log=log, relu_band=1e-3, relu_nudge=1e-2)
    assert left <= 2 and rounds <= 5 and log[0][2] == pairs and log[0][4] == int(relu_edge.sum()) and log[-1][4] == 0
    moved = settled["opacity"] != state["opacity"]
    assert bool(moved[on_edge].all()) and all(torch.equal(settled[k], state[k]) for k in state if k not in ("opacity", "local_feat"))
    rel = (torch.sigmoid(settled["opacity"][moved]) / torch.sigmoid(state["opacity"][moved]) - 1.0)
    assert 1.5e-3 <= float(rel.min()) and float(rel.max()) <= 1.3e-2          # one to a few nudges of 2e-3
    moved_f = (settled["local_feat"] != state["local_feat"]).any(1)
    assert bool(moved_f[relu_edge].all()) and int(moved_f.sum()) <= 3 * int(relu_edge.sum())
    # no hidden unit of a visible Gaussian is left within the band of its ReLU's switch, no alpha within 1e-4 of a skip threshold
    _, _, _, (relu_after, _, _) = SO.knife_scan(settled, kfd, cfg, workers=1, knife_eps=1e-4, knife_eps_T=0.0, relu_eps=1e-3)
    assert int(relu_after.sum()) == 0
    o = SO.optimisation_step(settled, kfd, cfg, torch.tensor([0.2, 0.5, 0.8]), True, workers=1, want_grads=False, knife_eps=1e-4)
    assert int(o["skip_knife"].sum()) <= 2 and int(o["raster_knife"].sum()) >= int(o["skip_knife"].sum())


# --------------------------------------------------------------------------------------------------------------- GPU: the default path
def _default_path_step(sc, kid, important, seed):
    """One `optimization_step` of the patched scene in the DEFAULT environment, the optimisers spied on: every gradient the step left
    in `.grad`; the SH colours (whose Adam step runs inside the projection backward and whose gradient is never written) through their
    first moment from a zero moment: exp_avg = (1 - b1) g on the visible rows."""
    from artdeco_amd import native_step
    from oracle import step_oracle as SO
    kf = sc.keyframes[kid]
    for k in ("f_dc", "f_rest"):
        sc.optimizer.params[k]["exp_avg"].zero_()
    got = {}
    orig = sc.optimizer.step

    def spy(*args, **kw):
        for k in ("xyz", "scaling", "rotation", "opacity", "local_feat", "global_feat"):
            got[k] = sc.gaussian_params[k]["val"].grad.clone()
        got.update({"mlp." + n: p.grad.clone() for n, p in sc.mlp_cov.named_parameters()})
        got.update({"kf." + n: getattr(kf, n).grad.clone() for n in ("rW2C", "tW2C", "exposure")})
        got["vis"], got["gvis"] = args[0].clone(), args[2].clone()
        return orig(*args, **kw)
    sc.optimizer.step = spy
    before = dict(native_step.STATS)
    torch.manual_seed(seed)
    got["loss"] = float(sc.optimization_step(kid, is_important=important))
    sc.optimizer.step = orig
    b1 = sc.optimizer.betas[0]
    for k in ("f_dc", "f_rest"):
        got[k] = sc.optimizer.params[k]["exp_avg"] / (1.0 - b1)
    got["native_calls"] = native_step.STATS["native"] - before["native"]
    got["invdepth"] = kf.latest_invdepth.clone()
    return got


def _hold_default_path_to_the_oracle(dev, N, W, H, important, lod, seed, monkeypatch, max_mask_flips=4, settle=True, **tols):
    from artdeco_amd import fused
    from harness import mapper
    from oracle import step_oracle as SO
    for k in ("ARTDECO_AMD_NATIVE_STEP", "ARTDECO_AMD_LOD_ADAM", "ARTDECO_AMD_HAND_CHAIN"):
        monkeypatch.delenv(k, raising=False)           # the default environment
    sc = mapper.build_synthetic_mapper(N, W, H, dev, seed=seed, n_keyframes=2, targets="random", lod=lod)
    _perturb(sc, seed, lod_dmax=False)
    assert fused.patch_scene_model(sc)
    kid = 1
    kf = sc.keyframes[kid]
    state, kfd, cfg = SO.snapshot(sc, kid)
    t0 = time.time()
    settle_log = []
    if settle:
        state, rounds, left = SO.settle_scene(state, kfd, cfg, log=settle_log)     # the scene off the rasteriser's skip edges: opacities of ~1 % of the Gaussians move by 2e-3
        with torch.no_grad():
            sc.gaussian_params["opacity"]["val"].copy_(state["opacity"].to(dev))
            sc.gaussian_params["local_feat"]["val"].copy_(state["local_feat"].to(dev))
        print(f"[step-oracle {N}/{W}x{H}] settle_scene {time.time() - t0:.1f} s: (round, Gaussians on a skip edge, pairs, pixels, Gaussians on a ReLU edge) {settle_log}")
        # what opacity cannot move: a pixel centre within ~3e-3 px of a splat's centre (|sigma| <= 1e-6, the band of the `sigma < 0` skip): ~2.5e-5 of
        # the visible Gaussians have one; the Gaussians blended on those pixels leave the max-error criterion (knife_rows), nothing else
        assert left <= 16 + 5e-5 * N, (left, settle_log)
    else:
        print(f"[step-oracle {N}/{W}x{H}] scene AS BUILT (no settle_scene): threshold-adjacent (splat, pixel) pairs stay in")
    torch.manual_seed(seed)
    bg = torch.rand(3, device=dev).cpu()              # what the step draws after the same seeding (h3dgsv3.py:421)
    rdk = SO.radial_decay_kernel(H, W, cfg["rad_decay"]).double()
    t0 = time.time()
    tm = {}
    o = SO.optimisation_step(state, kfd, cfg, bg, important, knife_rows_from="both", timings=tm,
                             adjust_targets=SO.move_targets_off_the_knife_edges(1e-3, outlier=not important, rdk=rdk))
    t_oracle = time.time() - t0
    lvl = kf.pyr_lvl
    kf.image_pyr[lvl] = o["gt"].float().to(dev).contiguous()
    kf.idepth_pyr[lvl] = o["mono"].float().to(dev).contiguous()
    got = _default_path_step(sc, kid, important, seed)
    assert got["native_calls"] == 1, "the step did not go through adk_mapper_step"
    label = f"{N}/{W}x{H} important={important} lod={lod}"
    n_img_knife = int(o["image_knife"].sum())
    print(f"[step-oracle {label}] oracle phases {({k: round(v, 1) for k, v in tm.items()})}")
    print(f"[step-oracle {label}] oracle {t_oracle:.1f} s, I = {o['n_isects']}, selected {int(o['selected'].sum())}, visible {int(o['visibility'].sum())}, "
          f"raster-knife pixels {float(o['raster_knife'].float().mean()):.4f}, image-knife pixels left {n_img_knife}, loss {got['loss']:.8f} vs {o['loss']:.8f}")
    # loss, masks, inverse depth
    assert abs(got["loss"] - o["loss"]) <= 1e-5 * abs(o["loss"]), (got["loss"], o["loss"])
    dv = int((got["vis"].cpu() != o["visibility"]).sum())
    dg = int((got["gvis"].cpu() != o["global_visibility"]).sum())
    print(f"[step-oracle {label}] visibility mask differs on {dv} of {N} Gaussians, voxel mask on {dg}")
    assert dv <= max_mask_flips and dg <= max_mask_flips     # a cull decision an ulp from its threshold (opacity 1/255, radius box on the border)
    keep = ~o["raster_knife"]
    inv_g, inv_o = got["invdepth"][0].double().cpu(), o["invdepth"][0]
    fin = torch.isfinite(inv_o) & torch.isfinite(inv_g)
    assert bool((torch.isfinite(inv_o) == torch.isfinite(inv_g))[keep].all())
    err = ((inv_g - inv_o).abs() * (keep & fin))
    assert float(torch.nan_to_num(err).max()) <= 1e-4 * float(inv_o[fin].abs().max())
    assert float(keep.float().mean()) > 0.95     # the termination edge at 1e-3: ~1 % of the pixels that terminate at all
    assert n_img_knife <= 64         # what is left are clamp edges (exposed render within 2e-5 of 0 or 1), which no target can move
    # The pose's 12 numbers are sums over every visible Gaussian of signed terms that nearly cancel on these centred synthetic views: the
    # oracle's OWN chain evaluated in torch fp32 on the same decisions (no kernel of this package) differs from its fp64 evaluation on them by
    # 2.4e-5 / 3.7e-5 at 1 M / 512x384, 1.05e-4 / 1.29e-4 at 648x486, 3.7e-4 / 4.7e-4 at 1920x1080 and 2e-3 / 3e-3 at 4 M / 2592x1944, with
    # every other leaf at 1e-5 (profiles/r05_fp32_floor_*.txt) -- and the HIP path measures THE SAME figures to two digits (5e-5, 1.2e-4 /
    # 1.5e-4, 3.7e-4 / 4.7e-4, 1.6e-3 / 2.4e-3): fp32 arithmetic, not a kernel.  The pose is therefore held to that floor with a margin
    # (tol_pose per size) and everything else to the north-star's 1e-4.
    _compare(got, o, label=label, **{"tol": 1e-4, "tol_pose": 3e-4, **tols})


@pytest.mark.gpu
def test_default_step_matches_the_fp64_oracle_at_1M_1080p(dev, monkeypatch):
    # max error: the fp32 floor of THIS frame is 8.47e-5 on xyz (profiles/r05_fp32_floor_1M_1080p.txt; the HIP path: 8.5e-5) -- too close to 1e-4 to
    # assert 1e-4 on a quantity that is a single worst element; rel_l2 stays at 1e-4 (measured <= 3.3e-5).
    # Pose (round 6): the camera sums are now carried in fp64 from each Gaussian's fp32 term to the accumulator (project_bwd: 64-bit shuffles +
    # fp64 atomics) -- and the deviation did NOT move: 3.7e-4 / 4.7e-4 before and after (profiles/r06_step_oracle_pose_fp64_accumulator.log).
    # It is not the reduction: each of the 925 k terms carries the rasteriser backward's own ~5e-6 fp32 error, and their sum cancels to ~12
    # terms' worth on this centred view -- the same figure the oracle's own chain gives in torch fp32.  Held to the measured floor + 25 %,
    # not to the 1e-3 of round 5.
    _hold_default_path_to_the_oracle(dev, 1_000_000, 1920, 1080, True, False, 11, monkeypatch, tol_pose=6e-4, tol_max=2e-4)


@pytest.mark.gpu
def test_default_step_matches_the_fp64_oracle_at_the_northstar_size(dev, monkeypatch):
    _hold_default_path_to_the_oracle(dev, 1_000_000, 512, 384, True, False, 12, monkeypatch)


@pytest.mark.gpu
def test_default_step_on_the_unsettled_scene_at_the_northstar_size(dev, monkeypatch):
    """The same step on the scene AS BUILT (round 6): no opacity is nudged off the alpha = 1/255 edge and no feature off a ReLU
    switch, so ~670 (splat, pixel) pairs within 1e-4 of a skip threshold -- and ~3 600 Gaussians with a hidden unit within 1e-3 of its switch --
    stay in.  The oracle decides every pixel on the SAME fp32 values the kernels decide on, so all but the few pairs within ~1e-6 of an edge
    agree; the Gaussians blended on a pixel that holds such a pair are the ORACLE's `knife_rows` (a few percent here: each near-threshold pixel
    has ~50 contributors) and leave the max-error criterion of the per-Gaussian leaves only.  Everything else is held as in the settled test:
    rel_l2 over ALL rows of every leaf, the summed leaves (mlp_cov, exposure, pose) included, and the max error on the non-knife rows."""
    _hold_default_path_to_the_oracle(dev, 1_000_000, 512, 384, True, False, 12, monkeypatch, settle=False, max_knife_rows=0.03)   # measured 1.05 % + 2 points
    # measured (profiles/r06_step_oracle_unsettled.log): rel_l2 <= 2.2e-5 on every leaf, max error on the non-knife rows <= 7.4e-5, over ALL
    # rows 1.2e-4 (xyz: one row under a flipped pair) -- the settled scene's figures (8e-6 / 1.7e-5) were not an artefact of settling


@pytest.mark.gpu
def test_default_step_matches_the_fp64_oracle_at_run_sh_geometry(dev, monkeypatch):
    _hold_default_path_to_the_oracle(dev, 1_000_000, 648, 486, False, False, 13, monkeypatch)


@pytest.mark.gpu
def test_default_step_matches_the_fp64_oracle_at_4M_with_lod(dev, monkeypatch):
    """configs[3].  At 2592x1944 a pixel coordinate has 2.4e-4 px of fp32 resolution, and the oracle's own chain evaluated in fp32 ON THE SAME
    DECISIONS (torch CPU, no kernel of this package involved) already differs from its fp64 evaluation by 1-3e-4 rel_l2 / up to 9e-3 max on the
    per-Gaussian leaves and 2-3e-3 on the pose, whose 12 numbers are sums of 3.4 M signed terms (profiles/r05_fp32_floor_4M.txt): the bounds
    here are that floor, not the 1e-4 an fp32 rasteriser cannot reach against fp64 at this size."""
    _hold_default_path_to_the_oracle(dev, 4_000_000, 2592, 1944, True, True, 14, monkeypatch, max_mask_flips=16, tol=3e-4, tol_max=2e-3,
                                     tol_pose=5e-3, max_knife_rows=0.02)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4, 5, 6, 7])
@pytest.mark.parametrize("reg", [0.0, 0.05])
def test_default_step_matches_the_fp64_oracle_small_scenes_with_lod_and_regulariser(reg, seed, dev, monkeypatch):
    """The five seeds and the regulariser of tests/test_fused_glue.py's former ladder (2e-4 / 2e-3 / 1e-2 against an fp32 GPU mirror), at 1e-4
    against the fp64 oracle: d_max in [1.5, 3.5] culls and fades part of the cloud, the scaling regulariser averages over the selected rows
    (h3dgsv3.py:443-449; since round 6 inside adk_mapper_step as three small launches -- run.sh's --scaling_reg_factor is 0)."""
    from artdeco_amd import fused
    from harness import mapper
    from oracle import step_oracle as SO
    for k in ("ARTDECO_AMD_NATIVE_STEP", "ARTDECO_AMD_LOD_ADAM", "ARTDECO_AMD_HAND_CHAIN"):
        monkeypatch.delenv(k, raising=False)
    N, W, H = 8000, 160, 112
    sc = _perturb(mapper.build_synthetic_mapper(N, W, H, dev, seed=seed, n_keyframes=2), seed, True)
    sc.scaling_reg_factor = reg
    assert fused.patch_scene_model(sc)
    for i in range(3):
        kid, important = i % 2, i != 1
        kf = sc.keyframes[kid]
        state, kfd, cfg = SO.snapshot(sc, kid)
        state, _, left = SO.settle_scene(state, kfd, cfg, workers=1)
        assert left <= 4
        with torch.no_grad():
            sc.gaussian_params["opacity"]["val"].copy_(state["opacity"].to(dev))
            sc.gaussian_params["local_feat"]["val"].copy_(state["local_feat"].to(dev))
        torch.manual_seed(100 + i)
        bg = torch.rand(3, device=dev).cpu()
        rdk = SO.radial_decay_kernel(H, W, cfg["rad_decay"]).double()
        o = SO.optimisation_step(state, kfd, cfg, bg, important, workers=1, knife_rows_from="both",
                                 adjust_targets=SO.move_targets_off_the_knife_edges(1e-3, outlier=not important, rdk=rdk))
        kf.image_pyr[kf.pyr_lvl] = o["gt"].float().to(dev).contiguous()
        kf.idepth_pyr[kf.pyr_lvl] = o["mono"].float().to(dev).contiguous()
        got = _default_path_step(sc, kid, important, 100 + i)
        assert got["native_calls"] == 1     # round 6: the regulariser rides inside adk_mapper_step too (it used to send the step to the per-stage chain)
        assert abs(got["loss"] - o["loss"]) <= 1e-5 * abs(o["loss"])
        assert torch.equal(got["vis"].cpu(), o["visibility"]) and torch.equal(got["gvis"].cpu(), o["global_visibility"])
        assert 0 < int(o["selected"].sum()) < N
        _compare(got, o, tol=1e-4, label=f"seed {seed} reg {reg} step {i}", max_knife_rows=0.05)   # one pixel of this frame blends ~1 % of the 8 000 Gaussians
