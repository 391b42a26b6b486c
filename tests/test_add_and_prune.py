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
    from artdeco_amd import fused, mapper
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
def test_patch_installs_add_and_prune(dev):
    from artdeco_amd import fused
    from tests.test_fused_glue import _scene
    sc = _scene(dev, N=2000, seed=2)
    assert fused.patch_scene_model(sc)
    assert sc.optimizer.add_and_prune.__func__ is fused.fused_add_and_prune
