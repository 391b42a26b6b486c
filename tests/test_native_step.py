"""adk_mapper_step (one native call per optimisation step, artdeco_amd/native_step.py) against the per-stage chain it replaces
(fused._train_on_keyframe_by_hand): same kernels in the same order, so the forward is identical to the bit and the gradients agree up
to the order of the rasteriser's atomics; plus the ways out of the native path (capacity retry, global-route frames, test keyframes)."""
import ctypes

import pytest
import torch


def test_argument_block_layout_matches_the_library():
    """The binding builds its struct from the header's field list; the library reports its own sizeof."""
    from artdeco_amd import _lib, native_step
    lib = _lib.load()
    assert int(lib.adk_mapper_step_args_bytes()) == ctypes.sizeof(native_step.StepArgs)
    kinds = [k for k, _ in native_step._FIELDS]
    names = [n for _, n in native_step._FIELDS]
    assert len(set(names)) == len(names)
    # pointers, then int64, then int32 / float: no padding anywhere, so field offsets are the running sum of the sizes
    order = "".join(kinds)
    assert order == "P" * order.count("P") + "L" * order.count("L") + "I" * order.count("I") + "F" * order.count("F")
    off = 0
    for kind, name in native_step._FIELDS:
        assert getattr(native_step.StepArgs, name).offset == off, name
        off += 8 if kind in "PL" else 4
    assert off == ctypes.sizeof(native_step.StepArgs) and off % 8 == 0
    assert len(native_step.STAGES) == 14 and native_step.STAGES[11] == "raster_bwd"


def test_stage_timings_fold_to_nothing_without_a_step():
    from artdeco_amd import native_step
    assert native_step.drain_timings() == {}


def _scene(dev, N=8000, seed=9, **kw):
    from test_fused_glue import _scene as make
    return make(dev, N=N, seed=seed, **kw)


_KEYS = ("xyz", "scaling", "rotation", "opacity", "local_feat", "global_feat")


def _one_step(sc, monkeypatch, native, important, kid=1, seed=5):
    """One optimization_step with the Gaussians' optimiser spied on: the gradients every leaf carries when the optimisers run."""
    from artdeco_amd import native_step
    monkeypatch.setenv("ARTDECO_AMD_NATIVE_STEP", "1" if native else "0")
    kf = sc.keyframes[kid]
    got = {}
    orig = sc.optimizer.step

    def spy(*args, **kw):
        g = {k: sc.gaussian_params[k]["val"].grad.clone() for k in _KEYS if sc.gaussian_params[k]["val"].grad is not None}
        g.update({"mlp." + n: p.grad.clone() for n, p in sc.mlp_cov.named_parameters()})
        g.update({"kf." + n: getattr(kf, n).grad.clone() for n in ("rW2C", "tW2C", "exposure") if getattr(kf, n).grad is not None})
        got["grads"] = g
        got["vis"], got["gvis"] = args[0].clone(), args[2].clone()
        return orig(*args, **kw)
    sc.optimizer.step = spy
    before = dict(native_step.STATS)
    torch.manual_seed(seed)
    got["loss"] = sc.optimization_step(kid, is_important=important)
    sc.optimizer.step = orig
    got["native_calls"] = native_step.STATS["native"] - before["native"]
    got["invdepth"] = kf.latest_invdepth.clone()
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("important", [True, False])
def test_native_step_matches_the_per_stage_chain(important, dev, monkeypatch):
    from artdeco_amd import fused
    a, b = _scene(dev), _scene(dev)
    assert fused.patch_scene_model(a) and fused.patch_scene_model(b)
    ga = _one_step(a, monkeypatch, True, important)
    gb = _one_step(b, monkeypatch, False, important)
    assert ga["native_calls"] == 1 and gb["native_calls"] == 0
    # the forward has no atomics: loss, inverse depth and both masks to the bit
    assert torch.equal(ga["loss"], gb["loss"]) and not ga["loss"].requires_grad
    assert torch.equal(ga["invdepth"], gb["invdepth"])
    assert torch.equal(ga["vis"], gb["vis"]) and torch.equal(ga["gvis"], gb["gvis"])
    assert set(ga["grads"]) == set(gb["grads"]) and len(ga["grads"]) == 6 + 4 + 3
    for k, x in ga["grads"].items():
        y = gb["grads"][k]
        assert x.shape == y.shape and float(y.abs().max()) > 0, k
        rel = float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30))
        assert rel <= 1e-5, (k, rel)
    # the colours took their Adam step inside the projection backward in both
    for k in ("f_dc", "f_rest"):
        assert a.gaussian_params[k]["val"].grad is None
        x, y = a.optimizer.params[k]["exp_avg"], b.optimizer.params[k]["exp_avg"]
        assert float(y.abs().max()) > 0
        assert float((x.double() - y.double()).norm() / y.double().norm()) <= 1e-5


@pytest.mark.gpu
def test_every_reuse_of_the_plan_matches_the_per_stage_chain(dev, monkeypatch):
    """Eight consecutive steps over both keyframes and both kinds of frame.  Before each one the per-stage scene is set to the native
    scene's state, so every step starts from identical parameters: the plan's buffers are reused by every step, and a gradient of step i
    that leaked into step i + 1 (a buffer the backward accumulates into and nobody cleared) would show at 1e-5."""
    from test_fused_glue import _sync_state
    from artdeco_amd import fused, native_step
    a, b = _scene(dev, N=6000, seed=4), _scene(dev, N=6000, seed=4)
    assert fused.patch_scene_model(a) and fused.patch_scene_model(b)
    n0 = native_step.STATS["native"]
    for i in range(8):
        _sync_state(a, b)
        ga = _one_step(a, monkeypatch, True, i % 3 != 0, kid=i % 2, seed=20 + i)
        gb = _one_step(b, monkeypatch, False, i % 3 != 0, kid=i % 2, seed=20 + i)
        assert torch.equal(ga["loss"], gb["loss"]) and torch.equal(ga["invdepth"], gb["invdepth"]), i
        assert torch.equal(ga["vis"], gb["vis"]) and torch.equal(ga["gvis"], gb["gvis"]), i
        assert set(ga["grads"]) == set(gb["grads"])
        for k, x in ga["grads"].items():
            y = gb["grads"][k]
            rel = float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30))
            assert rel <= 1e-5, (i, k, rel)
    assert native_step.STATS["native"] - n0 == 8 and len(a.__dict__["_adk_step_plans"]) == 1
    for ka, kb in zip(a.keyframes, b.keyframes):
        assert ka.depth_loss_weight == kb.depth_loss_weight


@pytest.mark.gpu
def test_fused_forward_of_the_native_step_is_bit_identical_to_the_two_kernels(dev, monkeypatch):
    """Inside adk_mapper_step the LoD / mlp_cov forward and the projection forward are ONE kernel (DESIGN finding 39); the per-stage chain runs the
    two stand-alone kernels.  Effective opacity / scale / quaternion, the packed splat records, the radii and the rendered image must agree to the
    BIT over consecutive steps from synchronised states: the poses move every step, and the first builds of the fused kernel differed by an ulp in
    the fade factor of a few Gaussians at SOME poses only (an FMA formed in one kernel and not in the other)."""
    from test_fused_glue import _sync_state
    from artdeco_amd import fused, rasterizer
    a, b = _scene(dev, N=6000, seed=4), _scene(dev, N=6000, seed=4)
    assert fused.patch_scene_model(a) and fused.patch_scene_model(b)
    stash = {}
    lod_fwd, ras_fwd = fused.FusedLodParams.forward, rasterizer.RasterizeGaussians.forward

    def lod_spy(ctx, *args):
        out = lod_fwd(ctx, *args)
        stash["opac"], stash["scale"], stash["quat"] = out[0].clone(), out[1].clone(), out[2].clone()
        return out

    def ras_spy(ctx, *args):
        out = ras_fwd(ctx, *args)
        stash["render_colors"], stash["radii"], stash["rec"] = out[0].clone(), out[2].clone(), out[3].clone()
        return out
    monkeypatch.setattr(fused.FusedLodParams, "forward", staticmethod(lod_spy))
    monkeypatch.setattr(rasterizer.RasterizeGaussians, "forward", staticmethod(ras_spy))
    for i in range(10):
        _sync_state(a, b)
        imp, kid = i % 3 != 0, i % 2
        monkeypatch.setenv("ARTDECO_AMD_NATIVE_STEP", "1")
        torch.manual_seed(40 + i)
        a.optimization_step(kid, is_important=imp)
        plan = next(iter(a.__dict__["_adk_step_plans"].values()))
        native = {k: plan.t[k][:plan.n].clone() for k in ("opac", "scale", "quat", "rec", "radii")}
        native["render_colors"] = plan.t["render_colors"].clone()
        stash.clear()
        monkeypatch.setenv("ARTDECO_AMD_NATIVE_STEP", "0")
        torch.manual_seed(40 + i)
        b.optimization_step(kid, is_important=imp)
        assert set(stash) == set(native)
        for k, x in native.items():
            assert torch.equal(x.view(torch.int32), stash[k].view(torch.int32)), (i, k, int((x.view(torch.int32) != stash[k].view(torch.int32)).sum()))


@pytest.mark.gpu
def test_native_step_grows_its_lists_and_retries(dev, monkeypatch):
    """A frame with more intersections than the plan's capacity: ADK_STEP_ECAPACITY before anything was modified, lists grown, same step again."""
    from artdeco_amd import fused, native_step
    a, b = _scene(dev), _scene(dev)
    assert fused.patch_scene_model(a) and fused.patch_scene_model(b)
    monkeypatch.setattr(native_step, "_GRAIN", 1024)   # capacities round to 1 024 entries instead of 2^20
    monkeypatch.setattr(native_step.rasterizer, "_CAPACITY_HINT", {})
    real = native_step.StepPlan.__init__

    def tiny(self, lib, dev_, N, V, W, H, tile_px, capacity, shared=None):
        real(self, lib, dev_, N, V, W, H, tile_px, 1024, shared)
    monkeypatch.setattr(native_step.StepPlan, "__init__", tiny)
    r0 = native_step.STATS["capacity_retries"]
    ga = _one_step(a, monkeypatch, True, True)
    assert native_step.STATS["capacity_retries"] == r0 + 1 and ga["native_calls"] == 1
    plan = next(iter(a.__dict__["_adk_step_plans"].values()))
    assert plan.args.isect_capacity >= plan.out.n_isects > 1024
    gb = _one_step(b, monkeypatch, False, True)
    assert torch.equal(ga["loss"], gb["loss"]) and torch.equal(ga["invdepth"], gb["invdepth"])
    for k, x in ga["grads"].items():
        y = gb["grads"][k]
        assert float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30)) <= 1e-5, k


def _pile_up(sc, dev, n_pile):
    """Pile n_pile Gaussians into a pixel or two in front of keyframe 0; the rest keep covering the frame."""
    with torch.no_grad():
        kf = sc.keyframes[0]
        Rt = kf.get_Rt().detach()
        centre = -Rt[:3, :3].T @ Rt[:3, 3]
        fwd = Rt[2, :3]
        g = torch.Generator().manual_seed(1)
        sc.gaussian_params["xyz"]["val"][:n_pile] = (centre + 3.0 * fwd)[None] + 2e-3 * torch.randn(n_pile, 3, generator=g).to(dev)


@pytest.mark.gpu
def test_tile_lists_above_8192_entries_stay_on_the_native_step(dev, monkeypatch):
    """More than 8 192 Gaussians on one tile.  Until round 4 adk_mapper_step handed such a frame back (ADK_STEP_EROUTE) and the per-stage
    chain ran it through the global radix route; since round 5 the list is sorted inside the call (bin_tile_sort_long_kernel) -- gsplat's
    own sort has no list-length limit (h3dgsv3.py:664-680).  Same results as the per-stage chain, which ALSO takes the long-list sort now,
    and as the per-stage chain on the old global route (ADK_BIN_LONG=0), which shares no sorting code with it."""
    from artdeco_amd import fused, native_step
    a, b, c = (_scene(dev, N=20000, seed=2, lod=False) for _ in range(3))
    for sc in (a, b, c):
        assert fused.patch_scene_model(sc)
        _pile_up(sc, dev, 9000)
    f0, n0, l0 = native_step.STATS["fallback_route"], native_step.STATS["native"], native_step.STATS["long_list_steps"]
    ga = _one_step(a, monkeypatch, True, True, kid=0)
    assert native_step.STATS["fallback_route"] == f0 and native_step.STATS["native"] == n0 + 1 and ga["native_calls"] == 1
    assert native_step.STATS["long_list_steps"] == l0 + 1
    plan = next(iter(a.__dict__["_adk_step_plans"].values()))
    assert plan.out.max_tile > 8192
    gb = _one_step(b, monkeypatch, False, True, kid=0)
    monkeypatch.setenv("ADK_BIN_LONG", "0")
    gc = _one_step(c, monkeypatch, False, True, kid=0)
    monkeypatch.delenv("ADK_BIN_LONG")
    for other in (gb, gc):
        assert torch.equal(ga["loss"], other["loss"]) and torch.equal(ga["invdepth"], other["invdepth"])
        for k, x in ga["grads"].items():
            y = other["grads"][k]
            assert float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30)) <= 1e-5, k


@pytest.mark.gpu
def test_native_step_without_the_second_key_buffer_still_hands_long_lists_back(dev, monkeypatch):
    """An ABI-v16 caller's plan has no `pairs2`: adk_mapper_step must decide ADK_STEP_EROUTE BEFORE it has modified anything, and the per-stage
    chain runs the step (with the same random background)."""
    from artdeco_amd import fused, native_step
    a, b = _scene(dev, N=20000, seed=2, lod=False), _scene(dev, N=20000, seed=2, lod=False)
    for sc in (a, b):
        assert fused.patch_scene_model(sc)
        _pile_up(sc, dev, 9000)
    real = native_step.StepPlan.set_capacity

    def no_second_buffer(self, capacity):
        real(self, capacity)
        self.args.pairs2 = None
    monkeypatch.setattr(native_step.StepPlan, "set_capacity", no_second_buffer)
    f0, n0 = native_step.STATS["fallback_route"], native_step.STATS["native"]
    ga = _one_step(a, monkeypatch, True, True, kid=0)
    assert native_step.STATS["fallback_route"] == f0 + 1 and native_step.STATS["native"] == n0 and ga["native_calls"] == 0
    gb = _one_step(b, monkeypatch, False, True, kid=0)
    assert torch.equal(ga["loss"], gb["loss"]) and torch.equal(ga["invdepth"], gb["invdepth"])
    for k, x in ga["grads"].items():
        y = gb["grads"][k]
        assert float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30)) <= 1e-5, k


@pytest.mark.gpu
def test_native_step_on_a_test_keyframe_leaves_the_map_alone(dev, monkeypatch):
    """A test keyframe trains its own pose only (keyframe.py:114-123, h3dgsv3.py:458): no colour Adam inside the backward, the colour
    gradients land in `.grad`, the Gaussians and their moments do not move."""
    from artdeco_amd import fused, native_step
    monkeypatch.setenv("ARTDECO_AMD_NATIVE_STEP", "1")
    sc = _scene(dev)
    assert fused.patch_scene_model(sc)
    kf = sc.keyframes[1]
    kf.is_test = True
    kf.optimizer.params.pop("exposure", None)
    for pd in kf.optimizer.params.values():
        pd["lr"] = 1e-4
    snap = {k: pd["val"].detach().clone() for k, pd in sc.optimizer.params.items()}
    moments = {k: pd["exp_avg"].clone() for k, pd in sc.optimizer.params.items() if "exp_avg" in pd}
    pose = kf.rW2C.detach().clone(), kf.tW2C.detach().clone()
    n0 = native_step.STATS["native"]
    torch.manual_seed(3)
    sc.optimization_step(1, is_important=True)
    assert native_step.STATS["native"] == n0 + 1
    for k, v in snap.items():
        assert torch.equal(sc.optimizer.params[k]["val"].detach(), v), k
    for k, v in moments.items():
        assert torch.equal(sc.optimizer.params[k]["exp_avg"], v), k
    assert not torch.equal(kf.rW2C.detach(), pose[0]) and not torch.equal(kf.tW2C.detach(), pose[1])
    for k in ("f_dc", "f_rest"):
        g = sc.gaussian_params[k]["val"].grad
        assert g is not None and g.shape == sc.gaussian_params[k]["val"].shape and float(g.abs().max()) > 0
    # the keyframe's optimiser has no exposure entry, so nothing clears exposure.grad: the next step on this keyframe must still be native,
    # and the gradient accumulates as autograd's would (never into the plan's own buffer)
    g1 = kf.exposure.grad.clone()
    plan = next(iter(sc.__dict__["_adk_step_plans"].values()))
    assert kf.exposure.grad.data_ptr() != plan.t["v_exposure"].data_ptr()
    sc.optimization_step(1, is_important=True)
    assert native_step.STATS["native"] == n0 + 2
    assert torch.allclose(kf.exposure.grad, g1 + plan.grads["exposure"], rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
def test_the_plan_follows_a_growing_map(dev, monkeypatch):
    """The frame loop with densification: the map grows on every important frame.  A step plan is sized by capacity (N rounded up), so most
    densifications keep it and only re-point the leaves' `.grad` at shorter / longer views; crossing a capacity line builds a new one.  With the
    capacity grain shrunk to 512 Gaussians both happen within sixteen frames; every optimisation step stays on the one-call path and trains."""
    from artdeco_amd import fused, native_step
    from harness import mapper, stream
    monkeypatch.setenv("ARTDECO_AMD_NATIVE_STEP", "1")
    monkeypatch.setattr(native_step, "_N_GRAIN", 512)
    monkeypatch.setattr(native_step, "_V_GRAIN", 64)
    sc = mapper.build_synthetic_mapper(6000, 160, 112, dev, seed=1, n_keyframes=0, targets="random")
    assert fused.patch_scene_model(sc)
    frames = stream.synthetic_frames(sc, 16, seed=3, texture=0.08, slam_hw=(56, 80))
    before = dict(native_step.STATS)
    out = stream.run_stream(sc, frames, start_index=0)
    torch.cuda.synchronize()
    d = {k: native_step.STATS[k] - before[k] for k in before}
    assert out["steps"] > 100 and d["native"] == out["steps"] and d["fallback_layout"] == 0 and d["fallback_route"] == 0
    assert out["gaussians_added"] > 0 and sc.xyz.shape[0] != 6000
    plan = next(iter(sc.__dict__["_adk_step_plans"].values()))
    assert 2 <= d["plans_built"] < out["densified_frames"] + 2          # rebuilt when a capacity line was crossed, not on every densification
    assert plan.N >= plan.n == sc.xyz.shape[0] and plan.N % 512 == 0
    for k in ("xyz", "opacity", "scaling", "local_feat"):
        g = sc.gaussian_params[k]["val"].grad
        assert g is not None and g.shape == sc.gaussian_params[k]["val"].shape and bool(torch.isfinite(g).all())
    assert all(bool(torch.isfinite(pd["val"]).all()) for pd in sc.optimizer.params.values() if pd["val"].is_floating_point())


@pytest.mark.gpu
def test_plans_of_two_resolutions_share_the_per_map_buffers(dev, monkeypatch):
    """A scene trained at two pyramid levels has two step plans (image-sized buffers per resolution) over ONE set of per-Gaussian / per-voxel
    buffers; steps alternate between them and each still matches the per-stage chain."""
    import torch.nn.functional as F
    from test_fused_glue import _sync_state
    from artdeco_amd import fused, native_step
    a, b = _scene(dev, N=6000, seed=4), _scene(dev, N=6000, seed=4)
    for sc in (a, b):
        assert fused.patch_scene_model(sc)
        kf = sc.keyframes[1]
        kf.image_pyr = [kf.image_pyr[0], F.avg_pool2d(kf.image_pyr[0], 2).contiguous()]
        kf.idepth_pyr = [kf.idepth_pyr[0], F.avg_pool2d(kf.idepth_pyr[0], 2).contiguous()]
        kf.pyr_lvl = 1
    for i in range(4):
        _sync_state(a, b)
        ga = _one_step(a, monkeypatch, True, True, kid=i % 2, seed=30 + i)
        gb = _one_step(b, monkeypatch, False, True, kid=i % 2, seed=30 + i)
        assert ga["native_calls"] == 1 and torch.equal(ga["loss"], gb["loss"]) and torch.equal(ga["invdepth"], gb["invdepth"]), i
        for k, x in ga["grads"].items():
            y = gb["grads"][k]
            assert float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30)) <= 1e-5, (i, k)
    plans = list(a.__dict__["_adk_step_plans"].values())
    assert len(plans) == 2 and {(p.W, p.H) for p in plans} == {(a.width, a.height), (a.width // 2, a.height // 2)}
    for name in native_step.StepPlan.PER_MAP - {"v_dc", "v_rest"}:
        assert plans[0].t[name].data_ptr() == plans[1].t[name].data_ptr(), name
    for name in ("render_colors", "v_col", "image", "cam_grad", "pairs"):
        assert plans[0].t[name].data_ptr() != plans[1].t[name].data_ptr(), name
    one = sum(t.numel() * t.element_size() for t in plans[0].t.values())
    assert native_step.STATS["plan_bytes"] < 2 * one


@pytest.mark.gpu
def test_stage_timer_sees_the_stages_inside_the_native_call(dev, monkeypatch):
    from artdeco_amd import fused, rasterizer
    monkeypatch.setenv("ARTDECO_AMD_NATIVE_STEP", "1")
    sc = _scene(dev, N=4000)
    assert fused.patch_scene_model(sc)
    sc.optimization_step(0)
    t = rasterizer.StageTimer(only=("raster_bwd",))
    rasterizer.set_stage_timer(t)
    try:
        for i in range(3):
            sc.optimization_step(i % 2)
        s = t.summary_ms()
    finally:
        rasterizer.set_stage_timer(None)
    assert set(s) == {"raster_bwd"} and s["raster_bwd"]["count"] == 3 and 0 < s["raster_bwd"]["min_ms"] <= s["raster_bwd"]["mean_ms"]
    full = rasterizer.StageTimer()
    rasterizer.set_stage_timer(full)
    try:
        sc.optimization_step(0)
        s = full.summary_ms()
    finally:
        rasterizer.set_stage_timer(None)
    from artdeco_amd import native_step
    # the LoD forward and the projection forward are ONE kernel inside the call, timed as the projection's stage
    assert set(native_step.STAGES) - {"lod_params_fwd"} <= set(s) and "adam_multi" in s and all(v["count"] == 1 for v in s.values())
