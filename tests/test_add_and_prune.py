"""fused_add_and_prune (adk_compact_plan / adk_compact_apply) vs the torch restatement of
SparseGaussianAdam.add_and_prune (Reconstruct/scene/optimizers.py:163-219): bit-identical tensors, same dtypes,
same requires_grad, for prune+append, prune only (the weed_out_gaussians call, h3dgsv3.py:953) and append only."""
import types

import pytest
import torch


def _optimizer(dev, N, V, seed, with_meta=True):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    shapes = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (15, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,), "local_feat": (16,)}
    params = {}
    if with_meta:  # the reference keeps these in the optimiser's dict too (optimizers.py:66-68)
        params["id"] = {"val": torch.arange(N, device=dev).view(N, 1)}
        params["cls_id"] = {"val": torch.randint(0, V, (N, 1), generator=g).to(dev)}
        params["d_max"] = {"val": torch.rand(N, 1, generator=g).to(dev)}
    for k, sh in shapes.items():
        params[k] = {"val": r(N, *sh).requires_grad_(True), "exp_avg": r(N, *sh), "exp_avg_sq": r(N, *sh).abs(), "lr": torch.tensor(1e-3, device=dev)}
    params["global_feat"] = {"val": r(V, 16).requires_grad_(True), "exp_avg": r(V, 16), "exp_avg_sq": r(V, 16).abs(), "lr": torch.tensor(4e-3, device=dev)}
    params["xyz"]["lr"] = torch.rand(N, 3, generator=g).to(dev) * 5e-5
    return types.SimpleNamespace(params=params, lr_dict={"xyz": {"lr_init": 5e-5, "lr_decay": 1 - 2e-5}})


def _clone(opt):
    return types.SimpleNamespace(lr_dict=opt.lr_dict, params={k: {n: (t.detach().clone().requires_grad_(t.requires_grad) if torch.is_tensor(t) else t)
                                                                   for n, t in d.items()} for k, d in opt.params.items()})


def _extension(dev, E, Vn, seed, keys):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    shapes = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (15, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,), "local_feat": (16,), "d_max": (1,)}
    ext = {k: r(E, *shapes[k]) for k in keys if k in shapes}
    if "id" in keys:
        ext["id"] = torch.arange(10**6, 10**6 + E, device=dev).view(E, 1)
    if "cls_id" in keys:
        ext["cls_id"] = torch.randint(0, 50, (E, 1), generator=g).to(dev)
    if "global_feat" in keys:
        ext["global_feat"] = r(Vn, 16)
    return ext


ALL = ("id", "cls_id", "d_max", "xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "local_feat", "global_feat")


@pytest.mark.gpu
@pytest.mark.parametrize("N,E,keep_frac", [(10_000, 3_000, 0.8), (777, 0, 0.5), (5_000, 1_234, 1.0), (300, 50, 0.0), (1, 1, 1.0)])
def test_fused_add_and_prune_matches_torch(N, E, keep_frac, dev):
    from artdeco_amd import fused
    from harness import mapper
    ref = _optimizer(dev, N, 64, seed=N)
    got = _clone(ref)
    g = torch.Generator().manual_seed(E + 1)
    mask = (torch.rand(N, generator=g) < keep_frac).to(dev) if 0.0 < keep_frac < 1.0 else torch.full((N,), keep_frac >= 1.0, device=dev)
    ext = _extension(dev, E, 7 if E else 0, seed=3, keys=ALL)
    with torch.no_grad():
        mapper._add_and_prune(ref, ext, mask)
        fused.fused_add_and_prune(got, ext, mask)
    for k in ref.params:
        for n, t in ref.params[k].items():
            u = got.params[k][n]
            if not torch.is_tensor(t):
                assert u == t
                continue
            assert u.dtype == t.dtype and u.shape == t.shape and u.is_contiguous(), (k, n, u.shape, t.shape)
            assert torch.equal(u, t), (k, n)
            assert u.requires_grad == t.requires_grad, (k, n)
    assert got.params["xyz"]["val"].shape[0] == int(mask.sum()) + E


@pytest.mark.gpu
def test_fused_add_and_prune_with_one_empty_extension(dev):
    """A key whose extension is empty while the others add rows: its moments must grow by ITS OWN (zero) rows, exactly as
    optimizers.py:205-219 (`zeros_like(extension_tensors[key])`) does -- not by the rows of the first non-empty key."""
    from artdeco_amd import fused
    from harness import mapper
    N, E = 2_000, 300
    ref = _optimizer(dev, N, 64, seed=5)
    got = _clone(ref)
    mask = (torch.rand(N, generator=torch.Generator().manual_seed(9)) < 0.7).to(dev)
    ext = _extension(dev, E, 0, seed=3, keys=tuple(k for k in ALL if k != "global_feat"))
    ext["local_feat"] = ext["local_feat"][:0]   # this key adds nothing
    with torch.no_grad():
        mapper._add_and_prune(ref, ext, mask)
        fused.fused_add_and_prune(got, ext, mask)
    for k in ref.params:
        for n, t in ref.params[k].items():
            if torch.is_tensor(t):
                assert got.params[k][n].shape == t.shape and torch.equal(got.params[k][n], t), (k, n)
    assert got.params["local_feat"]["exp_avg"].shape[0] == got.params["local_feat"]["val"].shape[0] == int(mask.sum())


@pytest.mark.gpu
def test_patch_installs_add_and_prune(dev):
    from artdeco_amd import fused
    from tests.test_fused_glue import _scene
    sc = _scene(dev, N=2000, seed=2)
    assert fused.patch_scene_model(sc)
    assert sc.optimizer.add_and_prune.__func__ is fused.fused_add_and_prune


@pytest.mark.gpu
def test_fused_weed_out_gaussians_matches_the_keyframe_loop(dev):
    """adk_lod_visible_count over all keyframes == the reference's per-keyframe torch loop (h3dgsv3.py:943-950), and the
    patched weed_out_gaussians prunes the same Gaussians (every parameter, moment and learning rate)."""
    from artdeco_amd import fused
    from tests.test_fused_glue import _scene
    a, b = _scene(dev, N=30_000, seed=4), _scene(dev, N=30_000, seed=4)
    g = torch.Generator().manual_seed(0)
    for sc in (a, b):
        with torch.no_grad():  # LoD ranges such that a good part of the cloud is out of range of every keyframe
            sc.gaussian_params["d_max"]["val"].copy_((0.8 + 2.5 * torch.rand(30_000, 1, generator=torch.Generator().manual_seed(1))).to(dev))
        for k in range(40):    # 42 keyframes: more than one would ever unroll by hand, fewer than one LDS chunk
            Rt = torch.eye(4)
            Rt[:3, 3] = torch.randn(3, generator=g) * (0.5 if sc is a else 0.0)
            sc.add_keyframe(type(sc.keyframes[0])(sc.keyframes[0].image_pyr[0], sc.keyframes[0].idepth_pyr[0], Rt.to(dev), dev))
    for ka, kb in zip(a.keyframes, b.keyframes):  # same poses in both scenes
        with torch.no_grad():
            kb.rW2C.copy_(ka.rW2C); kb.tW2C.copy_(ka.tW2C)
    # counts
    ref = torch.zeros(30_000, dtype=torch.int, device=dev)
    for kf in a.keyframes:
        c = kf.get_Rt().transpose(0, 1).detach().inverse()[3, :3]
        ref += ((a.xyz - c).norm(dim=1, keepdim=True) < 2 * a.d_max).squeeze(-1).int()
    got = fused.lod_visible_count(a.xyz, a.d_max, a.keyframes, dev)
    assert float((got != ref).float().mean()) <= 1e-4 and int((got - ref).abs().max()) <= 1  # only exactly-on-the-boundary cases
    assert 0.05 < float((ref == 0).float().mean()) < 0.95
    # end to end
    assert fused.patch_scene_model(b)
    a.weed_out_gaussians()
    b.weed_out_gaussians()
    na, nb = a.xyz.shape[0], b.xyz.shape[0]
    assert abs(na - nb) <= 3 and na < 30_000
    if na == nb:
        for k in ("xyz", "f_rest", "cls_id", "d_max"):
            assert torch.equal(a.gaussian_params[k]["val"], b.gaussian_params[k]["val"]), k
        assert torch.equal(a.gaussian_params["xyz"]["exp_avg_sq"], b.gaussian_params["xyz"]["exp_avg_sq"])
        assert torch.equal(a.gaussian_params["xyz"]["lr"], b.gaussian_params["xyz"]["lr"])
    # the scene still trains after pruning
    torch.manual_seed(0)
    assert torch.isfinite(b.optimization_step(0))
