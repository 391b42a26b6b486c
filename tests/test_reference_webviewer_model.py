"""ARTDECO's REAL web-viewer scene model -- `Reconstruct/webviewer/scene_models.py`, imported from /root/reference, nothing extracted or
copied -- driving the two natives only IT consumes (SURVEY.md 8 rows a6 and a5's second call site):

 * `diff_gaussian_rasterization.GaussianRasterizationSettings / GaussianRasterizer` (scene_models.py:559-605): `SceneModel.render` and
   `render_from_id` run through the drop-in ADAPTER (`artdeco_amd/dropin/diff_gaussian_rasterization/_rasterizer.py` -- its translation of
   the Inria conventions is the code under test: transposed view matrix, intrinsics from tan(fov) with a centred principal point, dc ++ rest
   as SH, `scale_modifier`, background compositing, radii = max(rx, ry), the 0.3 px^2 low-pass), with the ONE call below the adapter
   (`rasterizer.render_camera`, HIP only) bound to the CPU oracle; the result must be the oracle's render of the same Gaussians computed
   directly from the model's raw parameters;
 * `simple_knn._C.distIndex2` (scene_models.py:1003): `SceneModel.place_anchor_if_needed` runs through the drop-in's own glue (argument
   checks, output allocation, the flat [P*K] return convention the caller `.view(-1, k)`s) with the native call `adk_knn_index2` bound to the
   REFERENCE's `simple_knn.cu` compiled for the host (oracle/_ref).

CPU container only (the reference tree is not on the GPU box); the GPU-side parity of both natives is `tests/test_raster.py::
test_gaussian_rasterizer_adapter` and `tests/test_knn.py` / `tests/test_ref_pinning.py`.  Third-party packages that are absent here and
irrelevant to the path are stubbed, as in tests/test_reference_scene_model.py.
"""
import contextlib
import ctypes
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

from test_reference_scene_model import REF, _Stub, _args

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
MODULE = "Reconstruct.webviewer.scene_models"


def _cpu_device_kw(fn):
    def wrapped(*a, **k):
        if str(k.get("device", "")).startswith("cuda"):
            k["device"] = "cpu"
        return fn(*a, **k)
    return wrapped


class _TorchNoCuda:
    """`torch` as the web-viewer module sees it in this CPU-only container: a literal device="cuda" (scene_models.py:592, :1037) means the
    default device."""

    def __getattr__(self, name):
        return getattr(torch, name)

    tensor = staticmethod(_cpu_device_kw(torch.tensor))
    zeros = staticmethod(_cpu_device_kw(torch.zeros))
    full = staticmethod(_cpu_device_kw(torch.full))


@pytest.fixture()
def wv_module(monkeypatch):
    import artdeco_amd
    artdeco_amd.install_dropins()
    monkeypatch.syspath_prepend(REF)
    for m in ("cv2", "torchvision", "torchvision.utils", "plyfile", "lpips", "kornia", "pypose", "open3d", "trimesh", "imageio", "roma",
              "e3nn", "e3nn.o3", "cupy"):
        if m not in sys.modules:
            monkeypatch.setitem(sys.modules, m, _Stub(m))
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    saved = {k: v for k, v in sys.modules.items() if k.startswith(("Reconstruct", "dataloaders"))}
    for k in saved:
        del sys.modules[k]

    def load():
        # webviewer/anchors.py:33 evaluates `torch.zeros(3, device="cuda")` as a default argument AT IMPORT: on a box without a GPU the
        # import itself needs the default device there
        real_zeros = torch.zeros
        torch.zeros = _cpu_device_kw(real_zeros)
        try:
            mod = importlib.import_module(MODULE)
        finally:
            torch.zeros = real_zeros
        monkeypatch.setattr(mod, "torch", _TorchNoCuda())
        return mod
    yield load
    for k in [k for k in sys.modules if k.startswith(("Reconstruct", "dataloaders"))]:
        del sys.modules[k]
    sys.modules.update(saved)


def _oracle_render_camera(calls):
    """`artdeco_amd.rasterizer.render_camera`'s contract (rasterizer.py:528-538) on the CPU oracle, for the arguments the adapter passes."""
    from oracle import gsplat_oracle as go

    def render_camera(means, quats, scales, opacities, colors, viewmat, K, width, height, *, sh_degree, eps2d=0.3, inv_depth=False,
                      want_main_ids=False, sh_rest=None, **kw):
        assert not kw and inv_depth and want_main_ids and sh_rest is not None
        calls.append(dict(N=means.shape[0], width=width, height=height, sh_degree=sh_degree, eps2d=eps2d, K=K.clone(), viewmat=viewmat.clone()))
        cols = torch.cat([colors, sh_rest], dim=1)
        r, a, meta = go.rasterization(means, quats, scales, opacities, cols, viewmat, K, width, height, sh_degree=sh_degree, eps2d=eps2d,
                                      render_mode="RGB", grad_dtype=torch.float32)
        p = meta["p32"]
        inv = torch.where(p["valid"], 1.0 / p["depths"].clamp(min=1e-9), torch.zeros_like(p["depths"]))
        ex = {}
        rid, _, _ = go.rasterize_to_pixels(p["means2d"], p["conics"], inv[:, None], opacities.detach(), width, height, meta["isects"], extras=ex)
        main = torch.where(a.detach()[..., 0] > 0, ex["main_ids"].to(torch.int64), torch.full_like(ex["main_ids"], -1).to(torch.int64)).to(torch.int32)
        return (torch.cat([r, rid], dim=-1), a, meta["radii"], None, None, None, None, None, None, main)
    return render_camera


def _wv_args():
    return _args(anchor_overlap=0.3)


def _populate(scene, sc):
    """The empty model filled through the reference's own SparseGaussianAdam.add_and_prune (optimizers.py:163-219) with raw parameters whose
    activations (scene_models.py:219-245: exp / normalise / sigmoid) are the oracle scene's."""
    N = sc["means"].shape[0]
    op = sc["opacities"].clamp(1e-4, 1 - 1e-4)
    ext = {"xyz": sc["means"].clone(), "f_dc": sc["colors"][:, :1].clone(), "f_rest": sc["colors"][:, 1:].clone(), "scaling": sc["scales"].log(),
           "rotation": sc["quats"].clone(), "opacity": (op / (1 - op)).log()[:, None], "id": torch.zeros(N, 1, dtype=torch.long)}
    scene.optimizer.add_and_prune(ext, torch.ones(0, dtype=torch.bool))
    assert scene.xyz.shape[0] == N and scene.n_active_gaussians == N


def test_webviewer_module_binds_the_dropins(wv_module):
    import diff_gaussian_rasterization as dgr
    import simple_knn._C
    mod = wv_module()
    here = os.path.dirname(os.path.abspath(__import__("artdeco_amd").__file__))
    assert mod.GaussianRasterizer is dgr.GaussianRasterizer and mod.GaussianRasterizationSettings is dgr.GaussianRasterizationSettings
    assert dgr.__file__.startswith(here)
    assert mod.distIndex2 is simple_knn._C.distIndex2
    assert len(dgr.GaussianRasterizationSettings._fields) == 11        # scene_models.py:559-571 passes eleven positional values


def test_webviewer_render_goes_through_the_adapter(wv_module, monkeypatch):
    from harness import mapper
    from oracle import gsplat_oracle as go
    mod = wv_module()
    adapter = sys.modules["diff_gaussian_rasterization._rasterizer"]
    calls = []
    monkeypatch.setattr(adapter, "render_camera", _oracle_render_camera(calls))
    W, H, N = 96, 64, 1500
    sc = go.synthetic_scene(N, W, H, seed=5)
    fx = float(sc["K"][0, 0])
    K = torch.tensor([[fx, 0, (W - 1) / 2], [0, fx, (H - 1) / 2], [0, 0, 1.0]])
    scene = mod.SceneModel(W, H, K, _wv_args(), device="cpu")
    assert scene.optimizer.__class__.__name__ == "SparseGaussianAdam"
    _populate(scene, sc)
    Rt = sc["viewmat"].clone()
    Rt[:3, 3] += torch.tensor([0.05, -0.03, 0.1])
    bg = torch.tensor([0.1, 0.2, 0.3])
    pkg = scene.render(W, H, Rt.transpose(0, 1), 1.0, bg)               # scene_models.py:518: the caller hands the TRANSPOSED matrix
    assert len(calls) == 1 and calls[0]["N"] == N and calls[0]["sh_degree"] == 3 and calls[0]["eps2d"] == 0.3
    assert torch.equal(calls[0]["viewmat"], Rt)
    # what the model means: its activated parameters, a pinhole camera of focal W / (2 tan(fov_x / 2)) = f with the principal point at the
    # image centre (the rasteriser family's projection matrix), SH degree 3, the 0.3 px^2 dilation
    Kc = torch.tensor([[W / (2 * scene.tanfovx), 0, W / 2.0], [0, H / (2 * scene.tanfovy), H / 2.0], [0, 0, 1.0]])
    assert abs(float(Kc[0, 0]) - fx) < 1e-3 * fx and torch.allclose(calls[0]["K"], Kc)
    with torch.no_grad():
        P = scene.gaussian_params
        cols = torch.cat([P["f_dc"]["val"], P["f_rest"]["val"]], dim=1)
        r, a, meta = go.rasterization(scene.xyz, scene.rotation, scene.scaling, scene.opacity[:, 0], cols, Rt, Kc, W, H, sh_degree=3, eps2d=0.3,
                                      render_mode="RGB", backgrounds=bg)
    assert float(a.max()) > 0.5 and float((a[..., 0] == 0).float().mean()) < 0.9           # a real picture, not an empty frame
    assert pkg["render"].shape == (3, H, W) and pkg["invdepth"].shape == (1, H, W) and pkg["mainGaussID"].shape == (1, H, W)
    assert torch.allclose(pkg["render"].detach(), r.permute(2, 0, 1), atol=2e-6)
    assert pkg["radii"].shape == (N,) and torch.equal(pkg["radii"], meta["radii"].max(dim=1).values)
    assert torch.equal(pkg["visibility_filter"], meta["radii"].max(dim=1).values > 0) and 0 < int(pkg["visibility_filter"].sum()) <= N
    assert pkg["mainGaussID"].dtype == torch.int32 and bool(((pkg["mainGaussID"][0] == -1) == (a[..., 0] == 0)).all())
    assert float(pkg["invdepth"].min()) >= 0 and float(pkg["invdepth"].max()) > 0
    assert torch.equal(pkg["scale"], scene.scaling)
    # gradients reach the model's raw parameters through the adapter (what optimization_step's loss.backward() relies on, :372)
    (pkg["render"].sum() + pkg["invdepth"].sum()).backward()
    for k in ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity"):
        g = scene.gaussian_params[k]["val"].grad
        assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0, k

    # top view (:585-588): unit opacity, constant scale = the modifier, scale_modifier 1 in the settings
    calls.clear()
    top = scene.render(W, H, Rt.transpose(0, 1), 0.02, bg, top_view=True)
    with torch.no_grad():
        rt, _, _ = go.rasterization(scene.xyz, scene.rotation, torch.full_like(scene.scaling, 0.02), torch.ones(N), cols, Rt, Kc, W, H,
                                    sh_degree=3, eps2d=0.3, render_mode="RGB", backgrounds=bg)
    assert torch.allclose(top["render"].detach(), rt.permute(2, 0, 1), atol=2e-6)
    # a scaling modifier in the ordinary view multiplies the scales inside the adapter (GaussianRasterizationSettings.scale_modifier)
    half = scene.render(W, H, Rt.transpose(0, 1), 0.5, bg)
    with torch.no_grad():
        rh, _, _ = go.rasterization(scene.xyz, scene.rotation, 0.5 * scene.scaling, scene.opacity[:, 0], cols, Rt, Kc, W, H, sh_degree=3,
                                    eps2d=0.3, render_mode="RGB", backgrounds=bg)
    assert torch.allclose(half["render"].detach(), rh.permute(2, 0, 1), atol=2e-6)
    assert not torch.allclose(half["render"].detach(), pkg["render"].detach(), atol=1e-3)

    # render_from_id (:505-526): the keyframe's pose, transposed by the caller, and its exposure applied to the adapter's colour
    kf = mapper.Keyframe(torch.rand(3, H, W), torch.rand(1, H, W), Rt.clone(), "cpu")
    with torch.no_grad():
        kf.exposure.add_(0.05 * torch.randn(3, 4, generator=torch.Generator().manual_seed(1)))
    scene.keyframes = [kf]
    out = scene.render_from_id(0, bg=bg)
    with torch.no_grad():
        Rk = kf.get_Rt()
        rk, _, _ = go.rasterization(scene.xyz, scene.rotation, scene.scaling, scene.opacity[:, 0], cols, Rk, Kc, W, H, sh_degree=3, eps2d=0.3,
                                    render_mode="RGB", backgrounds=bg)
        want = ((kf.exposure[:3, :3] @ rk.permute(2, 0, 1).reshape(3, -1)) + kf.exposure[:3, 3, None]).clamp(0, 1).view(3, H, W)
    assert torch.allclose(out["render"].detach(), want, atol=5e-6)

    # an empty model (:606-615) never reaches the native
    calls.clear()
    empty = mod.SceneModel(W, H, K, _wv_args(), device="cpu")
    e = empty.render(W, H, Rt.transpose(0, 1), 1.0, bg)
    assert not calls and e["render"].shape == (3, H, W) and float(e["render"].abs().max()) == 0


class _FakeKnnLib:
    """libartdeco_hip.so's two KNN entry points the drop-in's `distIndex2` calls, on HOST pointers, computed by the reference's own
    simple_knn.cu compiled for the host (oracle/_ref): what reaches the native and what comes back is the drop-in glue's doing."""

    def __init__(self):
        self.calls = []

    def adk_knn_workspace_bytes(self, n):
        return 64

    def adk_knn_index2(self, pts, P, K, dists, idx, ws, ws_bytes, stream):
        from oracle import ref_native
        xyz = np.ctypeslib.as_array((ctypes.c_float * (3 * P)).from_address(pts)).reshape(P, 3)
        d, i = ref_native.knn_index2(xyz.copy(), K)
        np.ctypeslib.as_array((ctypes.c_float * (P * K)).from_address(dists))[:] = d.reshape(-1)
        np.ctypeslib.as_array((ctypes.c_int32 * (P * K)).from_address(idx))[:] = i.reshape(-1)
        self.calls.append((P, K, xyz.copy(), i.copy()))
        return 0


class _KfStub:
    def __init__(self, index):
        self.index, self.moved = index, []

    def to(self, device, *a, **k):
        self.moved.append(str(device))
        return self


def test_webviewer_anchor_merge_calls_distindex2_through_the_dropin(wv_module, monkeypatch):
    from oracle import ref_native
    if not ref_native.available():
        pytest.skip("oracle/_ref (the reference's simple_knn.cu built for the host) is not available")
    import simple_knn._C as knn
    from artdeco_amd import _lib
    from oracle import gsplat_oracle as go
    mod = wv_module()
    fake = _FakeKnnLib()
    monkeypatch.setattr(_lib, "load", lambda: fake)
    monkeypatch.setattr(_lib, "require_cuda", lambda *t: None)
    monkeypatch.setattr(_lib, "stream_of", lambda t: 0)
    monkeypatch.setattr(knn.torch.cuda, "device", lambda dev: contextlib.nullcontext())
    W, H, N = 96, 64, 2003
    sc = go.synthetic_scene(N, W, H, seed=7)
    fx = float(sc["K"][0, 0])
    K = torch.tensor([[fx, 0, (W - 1) / 2], [0, fx, (H - 1) / 2], [0, 0, 1.0]])
    scene = mod.SceneModel(W, H, K, _wv_args(), device="cpu")
    _populate(scene, sc)
    # 45 keyframes in the active anchor, the newest camera far enough for every Gaussian to look smaller than a pixel
    n_kf = 45
    scene.active_anchor.keyframes, scene.active_anchor.keyframe_ids = [], []
    scene.keyframes = [_KfStub(i) for i in range(n_kf)]
    for kf in scene.keyframes:
        scene.active_anchor.add_keyframe(kf)
    centres = torch.zeros(n_kf, 3)
    centres[:, 2] = torch.linspace(-60.0, -40.0, n_kf)
    scene.approx_cam_centres = centres
    with torch.no_grad():
        dist = (scene.xyz - centres[-1][None]).norm(dim=-1)
        screen = scene.f * scene.scaling.mean(dim=-1) / dist
    assert float((screen < 1).float().mean()) > 0.4 and bool((screen < 1.5).all())      # :984-989: the merge will run, on every Gaussian
    xyz_before = scene.xyz.detach().clone()
    torch.manual_seed(3)
    scene.place_anchor_if_needed()
    # the native was reached once, with every small Gaussian and K = 3 (:1003), and the caller could `.view(-1, k)` what came back
    assert len(fake.calls) == 1
    P, Kq, pts_seen, idx_ref = fake.calls[0]
    assert (P, Kq) == (N, 3) and np.array_equal(pts_seen, xyz_before.numpy())
    assert idx_ref.min() >= 0 and idx_ref.max() < N and not (idx_ref == np.arange(N)[:, None]).any()
    n_merged = N // 4
    assert scene.xyz.shape[0] == n_merged and len(scene.anchors) == 2 and scene.active_anchor is scene.anchors[-1]
    assert bool((scene.gaussian_params["id"]["val"] == n_kf - 1).all())
    lo, hi = xyz_before.min(dim=0).values, xyz_before.max(dim=0).values
    got = scene.xyz.detach()
    assert bool(torch.isfinite(got).all()) and bool((got >= lo - 1e-5).all()) and bool((got <= hi + 1e-5).all())   # weighted means of 4 members
    for k in ("f_dc", "f_rest", "opacity", "scaling", "rotation"):
        v = scene.gaussian_params[k]["val"]
        assert v.shape[0] == n_merged and bool(torch.isfinite(v).all()), k
        assert scene.gaussian_params[k]["exp_avg"].shape == v.shape
    assert scene.anchors[0].gaussian_params["xyz"]["val"].shape[0] == N                   # the previous set went to the first anchor
    assert len(scene.active_anchor.keyframes) == 20 and all("cpu" in kf.moved for kf in scene.keyframes[:25])
    # and the drop-in's own argument checks still stand in front of the native
    with pytest.raises(ValueError):
        knn.distIndex2(torch.zeros(5, 2), 3)
    with pytest.raises(TypeError):
        knn.distIndex2(torch.zeros(5, 3, dtype=torch.float64), 3)
    with pytest.raises(NotImplementedError):
        knn.distIndex2(torch.zeros(5, 3), 9)
