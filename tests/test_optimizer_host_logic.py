"""The mirror of ARTDECO's optimiser host logic (harness/mapper.py: BaseAdam, SparseGaussianAdam.step / add_and_prune)
against the REFERENCE's own classes (Reconstruct/scene/optimizers.py), run side by side on CPU.

The reference's classes are imported from /root/reference and executed unmodified; only the two native entry points they
call (`adamUpdate`, `adamUpdateBasic`: a pip extension that is not in the tree) are bound to the CPU oracle in BOTH
modules, so every difference would be a difference in host logic: which tensors are stepped, with which visibility mask
and learning rate, how the per-element learning rates decay and clamp, how rows are pruned and appended.  The GPU tests
(tests/test_fused_glue.py, tests/test_add_and_prune.py) then tie the fused HIP paths to this mirror bit for bit.
Runs only where the reference tree is mounted (the build container)."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

from oracle import adam_oracle

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")


def _adam_update(param, grad, exp_avg, exp_avg_sq, visible, lr, b1, b2, eps, N, M):
    lr_np = lr.detach().numpy() if torch.is_tensor(lr) else np.float32(lr)
    p, m, v = adam_oracle.adam_update_oracle(param.detach().numpy(), grad.numpy(), exp_avg.numpy(), exp_avg_sq.numpy(),
                                             visible.numpy(), lr_np, b1, b2, eps, N, M)
    param.data.copy_(torch.from_numpy(p)); exp_avg.copy_(torch.from_numpy(m)); exp_avg_sq.copy_(torch.from_numpy(v))


def _adam_update_basic(param, grad, exp_avg, exp_avg_sq, lr, b1, b2, eps):
    p, m, v = adam_oracle.adam_update_basic_oracle(param.detach().numpy(), grad.numpy(), exp_avg.numpy(), exp_avg_sq.numpy(), lr, b1, b2, eps)
    param.data.copy_(torch.from_numpy(p)); exp_avg.copy_(torch.from_numpy(m)); exp_avg_sq.copy_(torch.from_numpy(v))


@pytest.fixture()
def both(monkeypatch):
    import artdeco_amd
    artdeco_amd.install_dropins()
    sys.path.insert(0, REF)
    ref = __import__("Reconstruct.scene.optimizers", fromlist=["SparseGaussianAdam"])
    from harness import mapper
    for mod in (ref, mapper):
        monkeypatch.setattr(mod, "adamUpdate", _adam_update)
        monkeypatch.setattr(mod, "adamUpdateBasic", _adam_update_basic)
    yield ref, mapper
    sys.path.remove(REF)


def _params(g, n, n_vox):
    """The parameter dictionary of SceneModel (h3dgsv3.py: per-Gaussian tensors, per-voxel global_feat, mlp_* weights,
    bookkeeping ids), with n Gaussians."""
    r = lambda *s: torch.randn(*s, generator=g)
    P = {"xyz": {"val": r(n, 3).requires_grad_(True), "lr": 1.6e-4},
         "f_dc": {"val": r(n, 1, 3).requires_grad_(True), "lr": 2.5e-3},
         "f_rest": {"val": r(n, 15, 3).requires_grad_(True), "lr": 1.25e-4},
         "scaling": {"val": r(n, 3).requires_grad_(True), "lr": 5e-3},
         "rotation": {"val": r(n, 4).requires_grad_(True), "lr": 1e-3},
         "opacity": {"val": r(n, 1).requires_grad_(True), "lr": 5e-2},
         "local_feat": {"val": r(n, 16).requires_grad_(True), "lr": 7.5e-3},
         "global_feat": {"val": r(n_vox, 16).requires_grad_(True), "lr": 7.5e-3},
         "mlp_w1": {"val": r(32, 32).requires_grad_(True), "lr": 4e-3},
         "mlp_b1": {"val": r(32).requires_grad_(True), "lr": 4e-3},
         "id": {"val": torch.arange(n), "lr": 0.0},
         "cls_id": {"val": torch.randint(0, max(n_vox, 1), (n,), generator=g), "lr": 0.0},
         "d_max": {"val": r(n).abs(), "lr": 0.0}}
    return P


LR_DICT = {"xyz": {"lr_init": 1.6e-4, "lr_decay": 0.97}, "mlp_w1": {"lr_init": 4e-3, "lr_decay": 0.99}}


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        for f in a[k]:
            x, y = a[k][f], b[k][f]
            if torch.is_tensor(x) or torch.is_tensor(y):
                assert torch.is_tensor(x) and torch.is_tensor(y), (k, f)
                assert x.shape == y.shape or (x.numel() == 0 and y.numel() == 0), (k, f, x.shape, y.shape)
                assert torch.equal(x.reshape(-1), y.reshape(-1)), (k, f)
                assert x.dtype == y.dtype, (k, f)
            else:
                assert x == y, (k, f)


def _grads(P, g, step_keys):
    for k in step_keys:
        P[k]["val"].grad = torch.randn(P[k]["val"].shape, generator=g)


def test_sparse_adam_and_add_and_prune_match_the_reference_classes(both):
    ref, mapper = both
    g = torch.Generator().manual_seed(0)
    # the scene starts EMPTY and grows through add_and_prune, exactly as SceneModel does (h3dgsv3.py:938,953)
    Pa = _params(g, 0, 0)
    Pb = copy.deepcopy(Pa)
    A = ref.SparseGaussianAdam(Pa, lr_dict=copy.deepcopy(LR_DICT), device="cpu")
    B = mapper.SparseGaussianAdam(Pb, lr_dict=copy.deepcopy(LR_DICT), device="cpu")
    keys = [k for k in Pa if k not in ("id", "cls_id", "d_max")]
    n, n_vox = 0, 0
    for rnd, (add, add_vox, prune_frac) in enumerate([(300, 40, 0.0), (120, 10, 0.2), (0, 0, 0.3), (57, 0, 0.0)]):
        ext = _params(torch.Generator().manual_seed(100 + rnd), add, add_vox)
        ext_t = {k: v["val"].detach() for k, v in ext.items() if not k.startswith("mlp")}
        ext_t["id"] = ext_t["id"] + 1000 * rnd
        if add_vox == 0:
            ext_t["global_feat"] = torch.empty(0)
        valid = torch.rand(n, generator=g) >= prune_frac
        A.add_and_prune({k: v.clone() for k, v in ext_t.items()}, valid.clone())
        B.add_and_prune({k: v.clone() for k, v in ext_t.items()}, valid.clone())
        n, n_vox = int(valid.sum()) + add, n_vox + add_vox
        _same(Pa, Pb)
        assert Pa["xyz"]["val"].shape[0] == n and Pa["global_feat"]["val"].shape[0] == n_vox
        for it in range(3):
            gg = torch.Generator().manual_seed(1000 * rnd + it)
            vis, gvis = torch.rand(n, generator=gg) < 0.7, torch.rand(n_vox, generator=gg) < 0.5
            step_keys = [k for k in keys if not (it == 1 and k in ("opacity", "mlp_b1"))]  # a tensor without a gradient is skipped
            for P in (Pa, Pb):
                for k in keys:
                    P[k]["val"].grad = None
                _grads(P, torch.Generator().manual_seed(7 + 1000 * rnd + it), step_keys)
            A.step(vis.clone(), n, gvis.clone(), n_vox)
            B.step(vis.clone(), n, gvis.clone(), n_vox)
            _same(Pa, Pb)


def test_base_adam_matches_the_reference_class(both):
    ref, mapper = both
    g = torch.Generator().manual_seed(3)
    mk = lambda: {"rW2C": {"val": torch.randn(3, 2, generator=g).requires_grad_(True), "lr": 1e-4},
                  "tW2C": {"val": torch.randn(3, generator=g).requires_grad_(True), "lr": 1e-4},
                  "exposure": {"val": torch.randn(3, 4, generator=g).requires_grad_(True), "lr": 1e-3}}
    Pa = mk()
    Pb = copy.deepcopy(Pa)
    A, B = ref.BaseAdam(Pa), mapper.BaseAdam(Pb)
    for it in range(4):
        for P in (Pa, Pb):
            gg = torch.Generator().manual_seed(50 + it)
            for k in P:
                P[k]["val"].grad = None if (it == 2 and k == "tW2C") else torch.randn(P[k]["val"].shape, generator=gg)
        A.step(); B.step()
        _same(Pa, Pb)
    A.zero_grad(); B.zero_grad()
    assert all(p["val"].grad is None for p in Pa.values()) and all(p["val"].grad is None for p in Pb.values())


def test_patch_survives_reset_optimizer():
    """SceneModel.reset_optimizer (h3dgsv3.py:317-330, start of every finetune epoch) REPLACES the optimiser object; the fused
    step / add_and_prune installed by patch_scene_model must be installed on the replacement too (method swaps only: no GPU)."""
    import types
    from artdeco_amd import fused
    from harness import mapper
    sc = mapper.MapperScene(64, 48, 50.0, "cpu")
    c = mapper.synthetic_cloud(50, 64, 48, 0)
    sc.set_gaussians(c["means"], c["quats"], torch.log(c["scales"]), torch.zeros(50), c["sh"])
    assert fused.patch_scene_model(sc)
    first = sc.optimizer
    assert first.step.__func__ is fused.fused_optimizer_step
    sc.reset_optimizer()
    assert sc.optimizer is not first
    assert sc.optimizer.step.__func__ is fused.fused_optimizer_step
    assert sc.optimizer.add_and_prune.__func__ is fused.fused_add_and_prune
    assert isinstance(sc.reset_optimizer, types.MethodType)
