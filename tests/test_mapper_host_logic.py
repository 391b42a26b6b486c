"""The mirror of ARTDECO's mapper host code (harness/mapper.py: MapperScene.render / render_from_id /
optimization_step, Keyframe, sixD2mtx, radial_decay_kernel) against the REFERENCE's own source, executed on CPU.

`SceneModel.render`, `render_from_id`, `optimization_step` (Reconstruct/scene/scene_models/h3dgsv3.py:401-469, 595-700) and
`Reconstruct/utils.py`'s `sixD2mtx` / `radial_decay_kernel` are compiled FROM THE REFERENCE FILES (ast, nothing is copied into
this repository) and run with `self` = a deep copy of the mirror's scene object, so both sides start from identical state.
The natives the reference would call are bound to the CPU oracles on both sides (gsplat rasterization, fused-ssim,
adamUpdate*), `device="cuda"` / `.cuda()` are neutralised (CPU-only container), and the global RNG is re-seeded before each
side (the reference draws a random background per step).  Every difference that survives would be a difference in host
logic: LoD selection and alpha ratio, the mlp_cov modulation, K, compositing, visibility masks, exposure, the outlier mask
and the loss mix, pose and Gaussian optimiser calls.  The GPU tests (tests/test_fused_glue.py) then tie the fused HIP path to
this mirror.  Runs only where the reference tree is mounted."""
import ast
import contextlib
import copy
import os
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import adam_oracle, gsplat_oracle, ssim_oracle

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")


def _compile_defs(path, names, namespace, cls=None):
    """exec the named function definitions of a reference file (optionally methods of `cls`) into `namespace`."""
    tree = ast.parse(open(path).read())
    body = tree.body
    if cls is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == cls).body
    found = [n for n in body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in found} == set(names), (names, [n.name for n in found])
    mod = ast.Module(body=found, type_ignores=[])
    exec(compile(mod, path, "exec"), namespace)
    return [namespace[n] for n in names]


class _TorchNoCuda:
    """`torch` as the reference methods see it in this CPU-only container: device="cuda" means the default device."""

    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def tensor(*a, **k):
        if k.get("device") == "cuda":
            k.pop("device")
        return torch.tensor(*a, **k)

    @staticmethod
    def zeros(*a, **k):
        if k.get("device") == "cuda":
            k.pop("device")
        return torch.zeros(*a, **k)


def _rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, render_mode, rasterize_mode, absgrad, packed,
                   sh_degree, eps2d):
    assert (render_mode, rasterize_mode, absgrad, packed) == ("RGB+D", "classic", False, False)   # h3dgsv3.py:673-677
    r, a, meta = gsplat_oracle.rasterization(means, quats, scales, opacities, colors, viewmats[0], Ks[0], width, height, sh_degree=sh_degree,
                                             eps2d=eps2d, grad_dtype=torch.float32)
    return r[None], a[None], {"radii": meta["radii"][None]}


def _fused_ssim(img1, img2, padding="same", train=True):
    return ssim_oracle.fused_ssim_oracle(img1, img2, padding)


def _adam_update(param, grad, exp_avg, exp_avg_sq, visible, lr, b1, b2, eps, N, M):
    lr_np = lr.detach().numpy() if torch.is_tensor(lr) else np.float32(lr)
    p, m, v = adam_oracle.adam_update_oracle(param.detach().numpy(), grad.numpy(), exp_avg.numpy(), exp_avg_sq.numpy(),
                                             visible.numpy(), lr_np, b1, b2, eps, N, M)
    param.data.copy_(torch.from_numpy(p)); exp_avg.copy_(torch.from_numpy(m)); exp_avg_sq.copy_(torch.from_numpy(v))


def _adam_update_basic(param, grad, exp_avg, exp_avg_sq, lr, b1, b2, eps):
    p, m, v = adam_oracle.adam_update_basic_oracle(param.detach().numpy(), grad.numpy(), exp_avg.numpy(), exp_avg_sq.numpy(), lr, b1, b2, eps)
    param.data.copy_(torch.from_numpy(p)); exp_avg.copy_(torch.from_numpy(m)); exp_avg_sq.copy_(torch.from_numpy(v))


@pytest.fixture()
def world(monkeypatch):
    from harness import mapper
    fake_gsplat = types.SimpleNamespace(rendering=types.SimpleNamespace(rasterization=_rasterization))
    monkeypatch.setattr(mapper, "gsplat", fake_gsplat)
    monkeypatch.setattr(mapper, "fused_ssim", _fused_ssim)
    monkeypatch.setattr(mapper, "adamUpdate", _adam_update)
    monkeypatch.setattr(mapper, "adamUpdateBasic", _adam_update_basic)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    ns = {"torch": _TorchNoCuda(), "F": F, "np": np, "gsplat": fake_gsplat, "fused_ssim": _fused_ssim}
    ref_six, ref_rdk = _compile_defs(os.path.join(REF, "Reconstruct", "utils.py"), ["sixD2mtx", "radial_decay_kernel"], ns)
    ref_render, ref_render_from_id, ref_step = _compile_defs(
        os.path.join(REF, "Reconstruct", "scene", "scene_models", "h3dgsv3.py"), ["render", "render_from_id", "optimization_step"], ns,
        cls="SceneModel")
    return types.SimpleNamespace(mapper=mapper, six=ref_six, rdk=ref_rdk, render=ref_render, render_from_id=ref_render_from_id, step=ref_step)


def _scene(mapper, seed=0, d_max=None):
    torch.manual_seed(seed)
    sc = mapper.build_synthetic_mapper(260, 48, 32, "cpu", seed=seed, n_keyframes=2, targets="random")
    g = torch.Generator().manual_seed(seed + 5)
    with torch.no_grad():  # non-trivial features / LoD distances / poses / exposure so that every branch of the glue matters
        sc.gaussian_params["local_feat"]["val"].copy_(0.5 * torch.randn(sc.gaussian_params["local_feat"]["val"].shape, generator=g))
        sc.gaussian_params["global_feat"]["val"].copy_(0.5 * torch.randn(sc.gaussian_params["global_feat"]["val"].shape, generator=g))
        sc.gaussian_params["d_max"]["val"].copy_(2.0 + 3.0 * torch.rand(sc.gaussian_params["d_max"]["val"].shape, generator=g))
        for kf in sc.keyframes:
            kf.rW2C.add_(0.05 * torch.randn(3, 2, generator=g))
            kf.exposure.add_(0.05 * torch.randn(3, 4, generator=g))
    return sc


def _as_reference_self(sc):
    """The attributes the reference's methods read that the mirror does not carry."""
    sc.lock = contextlib.nullcontext()
    sc.args = types.SimpleNamespace(low_pass_filter_eps=sc.eps2d)
    sc.use_last_frame_proba, sc.last_trained_id, sc.valid_Rt_cache = 1.0, 0, {}
    sc.get_training_id = lambda: -1   # the keyframe CHOICE (h3dgsv3.py:383-399) is scheduling, not part of the step under test
    for m in ("render", "render_from_id"):  # the reference's optimization_step must call the reference's render
        setattr(sc, m, types.MethodType({"render": _as_reference_self.render, "render_from_id": _as_reference_self.render_from_id}[m], sc))
    return sc


def _state(sc):
    out = {}
    for k, p in sc.optimizer.params.items():
        for f in ("val", "exp_avg", "exp_avg_sq", "lr"):
            if f in p:
                out[f"{k}.{f}"] = p[f].detach().clone() if torch.is_tensor(p[f]) else p[f]
    for i, kf in enumerate(sc.keyframes):
        for k, p in kf.optimizer.params.items():
            for f in ("val", "exp_avg", "exp_avg_sq"):
                out[f"kf{i}.{k}.{f}"] = p[f].detach().clone()
        out[f"kf{i}.depth_loss_weight"] = kf.depth_loss_weight
    return out


def test_small_helpers_are_the_references(world):
    g = torch.Generator().manual_seed(0)
    r = torch.randn(5, 3, 2, generator=g)
    assert torch.equal(world.mapper.sixD2mtx(r), world.six(r))
    for H, W, s in ((32, 48, 0.5), (31, 7, 5 ** 0.5)):
        assert torch.equal(world.mapper.radial_decay_kernel(H, W, s), world.rdk(H, W, s))


def test_render_and_render_from_id_match_the_reference_methods(world):
    _as_reference_self.render, _as_reference_self.render_from_id = world.render, world.render_from_id
    a = _scene(world.mapper)
    b = _as_reference_self(copy.deepcopy(a))
    bg = torch.tensor([0.2, 0.5, 0.7])
    pa = a.render_from_id(1, 0, bg)
    pb = world.render_from_id(b, 1, 0, bg)
    assert pa.keys() == pb.keys()
    assert 0 < int(pa["visibility_filter"].sum()) < pa["visibility_filter"].numel()   # some culled by LoD / frustum, some not
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k
    # and the gradients of a scalar of the outputs w.r.t. every parameter
    (pa["render"].square().sum() + pa["invdepth"].clamp(max=10).sum() + pa["scale"].sum()).backward()
    (pb["render"].square().sum() + pb["invdepth"].clamp(max=10).sum() + pb["scale"].sum()).backward()
    for k in a.optimizer.params:
        ga, gb = a.optimizer.params[k]["val"].grad, b.optimizer.params[k]["val"].grad
        assert (ga is None) == (gb is None), k
        if ga is not None:
            assert torch.equal(ga, gb), k
    assert torch.equal(a.keyframes[1].rW2C.grad, b.keyframes[1].rW2C.grad) and torch.equal(a.keyframes[1].exposure.grad, b.keyframes[1].exposure.grad)


@pytest.mark.parametrize("important", [True, False])
def test_optimization_step_matches_the_reference_method(world, important):
    _as_reference_self.render, _as_reference_self.render_from_id = world.render, world.render_from_id
    a = _scene(world.mapper, seed=1)
    b = _as_reference_self(copy.deepcopy(a))
    for it in range(3):
        torch.manual_seed(100 + it)   # the per-step random background (h3dgsv3.py:424)
        a.optimization_step(-1, is_important=important)
        torch.manual_seed(100 + it)
        np.random.seed(it)
        world.step(b, is_important=important)
        sa, sb = _state(a), _state(b)
        assert sa.keys() == sb.keys()
        for k in sa:
            if torch.is_tensor(sa[k]):
                assert torch.equal(sa[k], sb[k]), (it, k)
            else:
                assert sa[k] == sb[k], (it, k)
        assert torch.equal(a.keyframes[-1].latest_invdepth, b.keyframes[-1].latest_invdepth)
    assert b.last_trained_id == -1


def test_weed_out_gaussians_matches_the_reference_method(world):
    ns = {"torch": _TorchNoCuda()}
    ref_weed, ref_dummy = _compile_defs(os.path.join(REF, "Reconstruct", "scene", "scene_models", "h3dgsv3.py"),
                                        ["weed_out_gaussians", "make_dummy_ext_tensor"], ns, cls="SceneModel")
    a = _scene(world.mapper, seed=2)
    with torch.no_grad():  # a third of the Gaussians are out of every keyframe's LoD range
        a.gaussian_params["d_max"]["val"][::3] = 0.4
    b = copy.deepcopy(a)
    b.args = types.SimpleNamespace(visible_threshold=a.visible_threshold)
    b.make_dummy_ext_tensor = types.MethodType(ref_dummy, b)
    n0 = a.xyz.shape[0]
    a.weed_out_gaussians()
    ref_weed(b)
    assert 0 < a.xyz.shape[0] < n0
    sa, sb = _state(a), _state(b)
    assert sa.keys() == sb.keys()
    for k in sa:
        if torch.is_tensor(sa[k]):
            assert sa[k].shape == sb[k].shape and torch.equal(sa[k], sb[k]), k
    for k in ("cls_id", "d_max"):
        assert torch.equal(a.gaussian_params[k]["val"], b.gaussian_params[k]["val"]), k


def test_rasteriser_oracle_uses_the_references_quaternion_and_covariance_convention():
    """gsplat itself is not in the reference tree, but ARTDECO's own 3-D covariance helpers are (Reconstruct/utils.py:651-689:
    build_rotation, build_scaling_rotation, build_covariance_from_scaling_rotation): quaternions are (w, x, y, z), normalised
    inside, Sigma = R S S^T R^T.  The rasteriser oracle must build the same rotation and covariance from the same parameters."""
    ns = {"torch": _TorchNoCuda()}
    build_rotation, build_scaling_rotation, strip_lowerdiag, strip_symmetric, build_cov = _compile_defs(
        os.path.join(REF, "Reconstruct", "utils.py"),
        ["build_rotation", "build_scaling_rotation", "strip_lowerdiag", "strip_symmetric", "build_covariance_from_scaling_rotation"], ns)
    g = torch.Generator().manual_seed(0)
    q = torch.randn(200, 4, generator=g) * torch.rand(200, 1, generator=g) * 3   # un-normalised, like the raw parameter
    s = torch.exp(torch.randn(200, 3, generator=g))
    Ro = torch.stack([torch.stack(row, -1) for row in gsplat_oracle._quat_to_rotmat(q)], -2)
    assert (Ro - build_rotation(q)).abs().max() < 2e-6
    M = Ro * s[:, None, :]
    cov = M @ M.transpose(1, 2)
    ref6 = build_cov(s, 1.0, q)
    mine6 = torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], -1)
    assert (mine6 - ref6).abs().max() <= 1e-5 * ref6.abs().max()
