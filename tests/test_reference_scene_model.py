"""ARTDECO's REAL scene model -- the class `run_system.py:113-114` instantiates, imported from
/root/reference/Reconstruct/scene/scene_models/h3dgsv3.py, nothing extracted or copied -- against this package:

 1. its module imports with `gsplat`, `fused_ssim`, `simple_knn._C`, `torch_scatter` and (through scene/optimizers.py)
    `diff_gaussian_rasterization` all resolving to the drop-ins;
 2. the post-import hook the drop-ins install (artdeco_amd/autoinstall.py) wraps `SceneModel.__init__`, so that an instance
    constructed by UNCHANGED host code carries the fused paths (and `ARTDECO_AMD_AUTOFUSE=0` leaves it alone);
 3. the instance, populated through the reference's own `SparseGaussianAdam.add_and_prune` and stepped with the reference's
    own `optimization_step` (natives bound to the CPU oracles: no GPU here), ends three optimisation steps with every
    parameter, moment and learning rate bit-identical to the harness mirror the GPU parity tests are written against.

CPU container only (the reference tree is not on the GPU box).  Third-party packages that are absent here and irrelevant to
the path (cv2, torchvision, plyfile, lpips, kornia, pypose, open3d, e3nn, cupy ...) are stubbed.
"""
import contextlib
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
MODULE = "Reconstruct.scene.scene_models.h3dgsv3"


class _Stub(types.ModuleType):
    """An importable nothing: attribute access yields sub-stubs, calls return the stub."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = _Stub(self.__name__ + "." + name)
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):
        return self


@pytest.fixture()
def ref_module(monkeypatch):
    import artdeco_amd
    artdeco_amd.install_dropins()
    monkeypatch.syspath_prepend(REF)
    for m in ("cv2", "torchvision", "torchvision.utils", "plyfile", "lpips", "kornia", "pypose", "open3d", "trimesh", "imageio", "roma",
              "e3nn", "e3nn.o3", "cupy"):
        if m not in sys.modules:
            monkeypatch.setitem(sys.modules, m, _Stub(m))
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)
    saved = {k: v for k, v in sys.modules.items() if k.startswith(("Reconstruct", "dataloaders"))}
    for k in saved:
        del sys.modules[k]
    yield lambda: importlib.import_module(MODULE)
    for k in [k for k in sys.modules if k.startswith(("Reconstruct", "dataloaders"))]:
        del sys.modules[k]
    sys.modules.update(saved)


def _args(**over):
    a = dict(num_prev_keyframes_check=5, sh_degree=3, lambda_dssim=0.2, init_proba_scaler=2.0, max_active_keyframes=200,
             use_last_frame_proba=0.0, scaling_reg_factor=0.0, rad_decay=float(np.sqrt(5.0)), position_lr_init=5e-5,
             position_lr_decay=1 - 2e-5, feature_lr=5e-3, scaling_lr=0.01, rotation_lr=2e-3, opacity_lr=0.1, feat_lr=4e-3,
             local_feat_dim=16, global_feat_dim=16, mlp_cov_lr_init=4e-3, mlp_cov_lr_decay=1 - 2e-5, voxel_size=0.1,
             visible_threshold=0.0, low_pass_filter_eps=0.01, gs_add_ratio=1.0)
    a.update(over)
    return types.SimpleNamespace(**a)


def _K(W, H, fx):
    return torch.tensor([[fx, 0, (W - 1) / 2], [0, fx, (H - 1) / 2], [0, 0, 1.0]])


def test_module_binds_the_dropins_and_autofuse_wraps_the_class(ref_module, monkeypatch):
    import diff_gaussian_rasterization
    import fused_ssim
    import gsplat
    import simple_knn._C
    import torch_scatter
    from artdeco_amd import autoinstall, fused
    autoinstall.on_dropin_import()
    mod = ref_module()
    here = os.path.dirname(os.path.abspath(__import__("artdeco_amd").__file__))
    assert mod.gsplat is gsplat and gsplat.__file__.startswith(here)
    assert mod.fused_ssim is fused_ssim.fused_ssim and mod.scatter_max is torch_scatter.scatter_max
    assert mod.distIndex2 is simple_knn._C.distIndex2
    opt_mod = sys.modules["Reconstruct.scene.optimizers"]
    assert opt_mod.adamUpdate is diff_gaussian_rasterization.adamUpdate
    assert mod.SparseGaussianAdam is opt_mod.SparseGaussianAdam
    # the hook wrapped the class as soon as the module had been executed
    assert getattr(mod.SceneModel, "_artdeco_amd_autofuse", False)
    scene = mod.SceneModel(64, 48, _K(64, 48, 51.2), _args(), device="cpu")
    assert scene._artdeco_amd_fused is True
    assert scene.render.__func__ is fused.fused_render and scene.optimization_step.__func__ is fused.fused_optimization_step
    assert scene.optimizer.step.__func__ is fused.fused_optimizer_step
    assert scene.weed_out_gaussians.__func__ is fused.fused_weed_out_gaussians
    # round 5: the Keyframe CLASS the module binds carries the one-launch get_Rt / set_Rt (sources pinned as the "pose" group); on CPU
    # parameters they run ARTDECO's own bodies
    assert mod.Keyframe.get_Rt is fused.fused_get_Rt and mod.Keyframe.set_Rt is fused.fused_set_Rt
    assert mod.Keyframe._unfused_get_Rt.__qualname__ == "Keyframe.get_Rt"
    scene.reset_optimizer()                     # h3dgsv3.py:317-330 replaces the optimiser; the wrapper follows it
    assert scene.optimizer.step.__func__ is fused.fused_optimizer_step
    monkeypatch.setenv("ARTDECO_AMD_AUTOFUSE", "0")
    plain = mod.SceneModel(64, 48, _K(64, 48, 51.2), _args(), device="cpu")
    assert not hasattr(plain, "_unfused_render") and plain.render.__func__ is mod.SceneModel.render


def test_real_scene_model_steps_bit_identically_to_the_mirror(ref_module, monkeypatch):
    import test_mapper_host_logic as H
    from harness import mapper
    monkeypatch.setenv("ARTDECO_AMD_AUTOFUSE", "0")   # the reference's OWN methods are what is compared here
    mod = ref_module()
    opt_mod = sys.modules["Reconstruct.scene.optimizers"]
    # natives -> CPU oracles, on both sides (the drop-ins run on the GPU only)
    fake_gsplat = types.SimpleNamespace(rendering=types.SimpleNamespace(rasterization=H._rasterization))
    for m in (mod, mapper):
        monkeypatch.setattr(m, "gsplat", fake_gsplat)
        monkeypatch.setattr(m, "fused_ssim", H._fused_ssim)
    for m in (opt_mod, mapper):
        monkeypatch.setattr(m, "adamUpdate", H._adam_update)
        monkeypatch.setattr(m, "adamUpdateBasic", H._adam_update_basic)
    monkeypatch.setattr(mod, "torch", H._TorchNoCuda())   # `device="cuda"` literals in render() mean the default device here

    W, Hh = 48, 32
    mirror = H._scene(mapper, seed=1)
    fx = W / (2 * mirror.tanfovx)
    real = mod.SceneModel(W, Hh, _K(W, Hh, fx), _args(), device="cpu")
    assert abs(real.tanfovx - mirror.tanfovx) < 1e-6 and abs(real.tanfovy - mirror.tanfovy) < 1e-6   # init_intrinsics (:968-979)
    # populate the empty real scene through the reference's own add_and_prune (optimizers.py:163-219)
    P = mirror.gaussian_params
    N = P["xyz"]["val"].shape[0]
    ext = {k: P[k]["val"].detach().clone() for k in ("cls_id", "d_max", "xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation",
                                                     "local_feat", "global_feat")}
    ext["id"] = torch.zeros(N, 1, dtype=torch.long)
    real.optimizer.add_and_prune(ext, torch.ones(0, dtype=torch.bool))
    assert real.xyz.shape[0] == N and real.global_feat.shape == mirror.global_feat.shape
    with torch.no_grad():
        for pr, pm in zip(real.mlp_cov.parameters(), mirror.mlp_cov.parameters()):
            pr.copy_(pm)
    # keyframes: the mirror's Keyframe carries what optimization_step / render_from_id read (keyframe.py:95-191)
    real.keyframes = []
    for i, kf in enumerate(mirror.keyframes):
        k2 = mapper.Keyframe(kf.image_pyr[0].clone(), kf.idepth_pyr[0].clone(), kf.get_Rt().detach().clone(), "cpu")
        with torch.no_grad():
            k2.rW2C.copy_(kf.rW2C); k2.tW2C.copy_(kf.tW2C); k2.exposure.copy_(kf.exposure)
        k2.index = i   # first/last_active_frame (:381-387); get_training_id (:393-399) samples only "cuda" keyframes, see below
        real.keyframes.append(k2)
    real.valid_Rt_cache = torch.ones(len(real.keyframes), dtype=torch.bool)
    real.last_trained_id = 0
    real.lock = contextlib.nullcontext() if not hasattr(real, "lock") else real.lock

    for step in range(3):
        kid = step % 2
        monkeypatch.setattr(real, "get_training_id", lambda kid=kid: kid, raising=False)   # (CPU keyframes would never be drawn)
        important = step != 1
        torch.manual_seed(100 + step)
        mirror.optimization_step(kid, is_important=important)
        torch.manual_seed(100 + step)
        real.optimization_step(is_important=important)
        assert real.last_trained_id == kid and not bool(real.valid_Rt_cache[kid])
    for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "local_feat", "global_feat"):
        for name in ("val", "exp_avg", "exp_avg_sq"):
            a, b = real.gaussian_params[k][name], mirror.gaussian_params[k][name]
            assert torch.equal(a.detach(), b.detach()), (k, name, float((a - b).abs().max()))
    assert torch.equal(real.gaussian_params["xyz"]["lr"], mirror.gaussian_params["xyz"]["lr"])
    for pr, pm in zip(real.mlp_cov.parameters(), mirror.mlp_cov.parameters()):
        assert torch.equal(pr.detach(), pm.detach())
    for kr, km in zip(real.keyframes, mirror.keyframes):
        assert torch.equal(kr.rW2C.detach(), km.rW2C.detach()) and torch.equal(kr.exposure.detach(), km.exposure.detach())
    # weed_out_gaussians (:942-953) on the real instance prunes the same rows as the mirror's
    with torch.no_grad():
        for sc in (real, mirror):
            sc.gaussian_params["d_max"]["val"][::3] = 0.05
    real.weed_out_gaussians()
    mirror.weed_out_gaussians()
    assert real.xyz.shape == mirror.xyz.shape and real.xyz.shape[0] < N
    assert torch.equal(real.xyz.detach(), mirror.xyz.detach())
    assert torch.equal(real.gaussian_params["f_rest"]["exp_avg_sq"], mirror.gaussian_params["f_rest"]["exp_avg_sq"])


def test_step_oracle_is_pinned_to_the_real_scene_model(ref_module, monkeypatch):
    """oracle/step_oracle.py (the fp64 restatement of the WHOLE optimisation step that the GPU tests hold `adk_mapper_step` to at the
    BASELINE sizes) against the reference's REAL `SceneModel.optimization_step` executed here on CPU with the natives bound to the fp32
    oracles: loss, both visibility masks, the inverse depth and every gradient the real class leaves in `.grad` (fp32 vs fp64: 2e-5)."""
    import test_mapper_host_logic as H
    from harness import mapper
    from oracle import step_oracle as SO
    monkeypatch.setenv("ARTDECO_AMD_AUTOFUSE", "0")
    mod = ref_module()
    opt_mod = sys.modules["Reconstruct.scene.optimizers"]
    monkeypatch.setattr(mod, "gsplat", types.SimpleNamespace(rendering=types.SimpleNamespace(rasterization=H._rasterization)))
    monkeypatch.setattr(mod, "fused_ssim", H._fused_ssim)
    for m in (opt_mod, mapper):           # the keyframes are the mirror's Keyframe objects (their BaseAdam calls mapper.adamUpdateBasic)
        monkeypatch.setattr(m, "adamUpdate", H._adam_update)
        monkeypatch.setattr(m, "adamUpdateBasic", H._adam_update_basic)
    monkeypatch.setattr(mod, "torch", H._TorchNoCuda())

    W, Hh = 48, 32
    mirror = H._scene(mapper, seed=2)
    fx = W / (2 * mirror.tanfovx)
    real = mod.SceneModel(W, Hh, _K(W, Hh, fx), _args(scaling_reg_factor=0.05), device="cpu")
    P = mirror.gaussian_params
    N = P["xyz"]["val"].shape[0]
    g = torch.Generator().manual_seed(4)
    ext = {k: P[k]["val"].detach().clone() for k in ("cls_id", "d_max", "xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation",
                                                     "local_feat", "global_feat")}
    ext["local_feat"] = 0.5 * torch.randn(N, 16, generator=g)
    ext["global_feat"] = 0.5 * torch.randn(ext["global_feat"].shape[0], 16, generator=g)
    dist = ext["xyz"].norm(dim=1, keepdim=True)
    ext["d_max"] = dist * (0.45 + 0.6 * torch.rand(N, 1, generator=g))       # some culled (dist >= 2 d_max), some fading, some plain
    ext["id"] = torch.zeros(N, 1, dtype=torch.long)
    real.optimizer.add_and_prune(ext, torch.ones(0, dtype=torch.bool))
    with torch.no_grad():
        for pr, pm in zip(real.mlp_cov.parameters(), mirror.mlp_cov.parameters()):
            pr.copy_(pm + 0.2 * torch.randn(pm.shape, generator=g))
    real.keyframes = []
    for i, kf in enumerate(mirror.keyframes):
        k2 = mapper.Keyframe(kf.image_pyr[0].clone(), kf.idepth_pyr[0].clone(), kf.get_Rt().detach().clone(), "cpu")
        with torch.no_grad():
            k2.rW2C.add_(0.01 * torch.randn(3, 2, generator=g))
            k2.exposure.add_(0.03 * torch.randn(3, 4, generator=g))
        k2.index = i
        real.keyframes.append(k2)
    real.valid_Rt_cache = torch.ones(len(real.keyframes), dtype=torch.bool)
    real.last_trained_id = 0
    real.lock = contextlib.nullcontext() if not hasattr(real, "lock") else real.lock
    for step, important in enumerate((True, False)):
        kid = step % 2
        kf = real.keyframes[kid]
        monkeypatch.setattr(real, "get_training_id", lambda kid=kid: kid, raising=False)
        state, kfd, cfg = SO.snapshot(real, kid)
        torch.manual_seed(200 + step)
        bg = torch.rand(3)
        o = SO.optimisation_step(state, kfd, cfg, bg, important, workers=1)
        assert 0 < int(o["selected"].sum()) < N
        got = {}
        orig = real.optimizer.step

        def spy(*a, _o=orig, **k):
            got.update({kk: real.gaussian_params[kk]["val"].grad.clone() for kk in SO.GAUSS_KEYS})
            got.update({"mlp." + n: p.grad.clone() for n, p in real.mlp_cov.named_parameters()})
            got.update({"kf." + n: getattr(kf, n).grad.clone() for n in ("rW2C", "tW2C", "exposure")})
            got["vis"], got["gvis"] = a[0].clone(), a[2].clone()
            return _o(*a, **k)
        monkeypatch.setattr(real.optimizer, "step", spy, raising=False)
        torch.manual_seed(200 + step)
        real.optimization_step(is_important=important)
        monkeypatch.setattr(real.optimizer, "step", orig, raising=False)
        assert torch.equal(got["vis"], o["visibility"]) and torch.equal(got["gvis"], o["global_visibility"])
        inv = kf.latest_invdepth.double()
        fin = torch.isfinite(o["invdepth"])
        assert torch.equal(torch.isfinite(inv), fin) and float((inv - o["invdepth"])[fin].abs().max()) <= 1e-5 * float(o["invdepth"][fin].abs().max())
        for k in SO.GAUSS_KEYS + SO.MLP_KEYS + SO.KF_KEYS:
            x, y = got[k].double(), o["grads"][k]
            assert x.shape == y.shape and float(y.abs().max()) > 0, k
            assert float((x - y).norm() / y.norm()) <= 2e-5 and float((x - y).abs().max() / y.abs().max()) <= 2e-5, (step, k)


def test_autofuse_refuses_a_host_method_that_moved(ref_module, monkeypatch, tmp_path, capsys):
    """A perturbed copy of the reference module (one constant of optimization_step changed): the post-import hook must leave
    the whole "step" group as ARTDECO wrote it -- natives only -- warn once, and still install the untouched "densify" group."""
    import importlib.util
    from artdeco_amd import autoinstall, fused, pins
    autoinstall.on_dropin_import()
    ref_module()   # the package context (Reconstruct.scene.optimizers, keyframe, utils) of the copy
    src = open(os.path.join(REF, "Reconstruct/scene/scene_models/h3dgsv3.py")).read()
    needle = "error_map[0] > 0.2"
    assert src.count(needle) == 1
    path = tmp_path / "h3dgsv3_moved.py"
    path.write_text(src.replace(needle, "error_map[0] > 0.25"))
    name = "Reconstruct.scene.scene_models.h3dgsv3_moved"
    spec = importlib.util.spec_from_file_location(name, str(path))
    mod = importlib.util.module_from_spec(spec)
    monkeypatch.setitem(sys.modules, name, mod)
    spec.loader.exec_module(mod)
    autoinstall._wrap_scene_model(mod)
    pins._warned.clear()
    capsys.readouterr()
    scene = mod.SceneModel(64, 48, _K(64, 48, 51.2), _args(), device="cpu")
    mod.SceneModel(64, 48, _K(64, 48, 51.2), _args(), device="cpu")
    ours = [ln for ln in capsys.readouterr().err.splitlines() if "[artdeco_amd] WARNING" in ln]
    assert len(ours) == 1 and "SceneModel.optimization_step" in ours[0]      # once, naming the method
    assert scene._artdeco_amd_skipped == {"step": ["SceneModel.optimization_step"]}
    for attr in ("render", "render_from_id", "optimization_step"):
        assert getattr(scene, attr).__func__ is getattr(mod.SceneModel, attr), attr
    assert scene.optimizer.step.__func__ is mod.SparseGaussianAdam.step
    assert scene.update_voxel.__func__ is fused.fused_update_voxel
    assert scene.optimizer.add_and_prune.__func__ is fused.fused_add_and_prune
    # round 6: a deployment that counts on the fused path stops at start-up instead of running eight times slower behind a warning
    monkeypatch.setenv("ARTDECO_AMD_REQUIRE_FUSED", "1")
    with pytest.raises(RuntimeError, match="SceneModel.optimization_step"):
        mod.SceneModel(64, 48, _K(64, 48, 51.2), _args(), device="cpu")
    monkeypatch.delenv("ARTDECO_AMD_REQUIRE_FUSED")
    # the unperturbed class verifies clean
    good = ref_module().SceneModel(64, 48, _K(64, 48, 51.2), _args(), device="cpu")
    assert good._artdeco_amd_skipped == {} and good.optimization_step.__func__ is fused.fused_optimization_step


def test_gc_freeze_is_opt_in(monkeypatch):
    import gc
    from artdeco_amd import fused
    from harness import mapper
    monkeypatch.delenv("ARTDECO_AMD_GC_FREEZE", raising=False)
    monkeypatch.setattr(fused, "_GC_FROZEN", False)
    before = gc.get_freeze_count()
    import test_mapper_host_logic as H
    sc = H._scene(mapper, seed=1)
    assert fused.patch_scene_model(sc)
    assert gc.get_freeze_count() == before and fused._GC_FROZEN is False
