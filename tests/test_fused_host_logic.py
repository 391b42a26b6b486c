"""Host-side pieces of the fused mapper step that need no GPU: the stand-in ctx that lets the four autograd Functions be
driven by hand, gradient hand-over to the leaves, the camera configuration checks, the record-and-defer context of
adamUpdateBasic (artdeco_amd/fused.py, rasterizer.py, dropin/diff_gaussian_rasterization)."""
import pytest
import torch

import artdeco_amd

artdeco_amd.install_dropins()


class _Scale(torch.autograd.Function):
    """A Function written the way the package's own are: saves tensors, marks by-products, reads needs_input_grad."""

    @staticmethod
    def forward(ctx, x, w, k):
        y = x * w
        n = torch.tensor(float(k))
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, w)
        ctx.k = k
        ctx.mark_non_differentiable(n)
        return y, n

    @staticmethod
    def backward(ctx, v_y, _v_n):
        x, w = ctx.saved_tensors
        return (v_y * w if ctx.needs_input_grad[0] else None), (v_y * x if ctx.needs_input_grad[1] else None), None


def test_hand_ctx_drives_a_function_like_the_engine():
    from artdeco_amd.rasterizer import HandCtx
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, generator=g, requires_grad=True)
    w = torch.randn(5, generator=g, requires_grad=True)
    y, n = _Scale.apply(x, w, 3)
    v = torch.randn(5, generator=g)
    (y * v).sum().backward()
    ctx = HandCtx((True, True, False))
    with torch.no_grad():
        y2, n2 = _Scale.forward(ctx, x, w, 3)
        vx, vw, vk = _Scale.backward(ctx, v, None)
    assert torch.equal(y2, y.detach()) and float(n2) == 3.0 and ctx.k == 3 and vk is None
    assert torch.equal(vx, x.grad) and torch.equal(vw, w.grad)
    ctx = HandCtx((False, True, False))
    with torch.no_grad():
        _Scale.forward(ctx, x, w, 3)
        vx, vw, _ = _Scale.backward(ctx, v, None)
    assert vx is None and torch.equal(vw, w.grad)


def test_gradient_hand_over_follows_accumulate_grad():
    from artdeco_amd.fused import _accumulate
    leaf = torch.zeros(3, 2, requires_grad=True)
    g = torch.arange(6.0)
    _accumulate(leaf, None)
    assert leaf.grad is None
    _accumulate(leaf, g)                       # a flat gradient is viewed to the leaf's shape, not copied
    assert leaf.grad.shape == (3, 2) and leaf.grad.data_ptr() == g.data_ptr()
    _accumulate(leaf, torch.ones(3, 2))        # a second contribution is added, as AccumulateGrad does
    assert torch.equal(leaf.grad, torch.arange(6.0).view(3, 2) + 1)
    frozen = torch.zeros(2)
    _accumulate(frozen, torch.ones(2))
    assert frozen.grad is None


def test_camera_config_checks_and_modes():
    from artdeco_amd import rasterizer as R
    f_dc, f_rest = torch.zeros(7, 1, 3), torch.zeros(7, 15, 3)
    cfg, cols, rest = R.camera_config(f_dc, 64, 48, sh_degree=3, eps2d=0.01, sh_rest=f_rest)
    assert (cfg.width, cfg.height, cfg.sh_degree, cfg.sh_K, cfg.color_mode) == (64, 48, 3, 16, R._COLOR_SH) and cols is f_dc and rest is f_rest
    cfg, cols, rest = R.camera_config(torch.zeros(7, 16, 3), 64, 48, sh_degree=2)
    assert cfg.sh_K == 16 and cfg.sh_degree == 2 and rest is None
    cfg, cols, rest = R.camera_config(torch.zeros(7, 3), 64, 48, sh_degree=None)
    assert cfg.color_mode == R._COLOR_RGB and cfg.sh_K == 0
    cfg, cols, rest = R.camera_config(None, 64, 48, sh_degree=None, depth_only=True, sh_rest=f_rest)
    assert cfg.color_mode == R._COLOR_DEPTH and cols is None and rest is None
    with pytest.raises(ValueError):
        R.camera_config(torch.zeros(7, 2, 3), 64, 48, sh_degree=3, sh_rest=f_rest)      # band 0 must be [N,1,3]
    with pytest.raises(ValueError):
        R.camera_config(f_dc, 64, 48, sh_degree=3, sh_rest=torch.zeros(7, 8, 3))        # 9 coefficients cannot hold degree 3
    with pytest.raises(ValueError):
        R.camera_config(torch.zeros(7, 4, 3), 64, 48, sh_degree=2)
    with pytest.raises(NotImplementedError):
        R.camera_config(torch.zeros(7, 16, 3), 64, 48, sh_degree=4)
    with pytest.raises(NotImplementedError):
        R.camera_config(torch.zeros(7, 4), 64, 48, sh_degree=None)


def test_deferred_basic_updates_scopes_and_restores():
    import diff_gaussian_rasterization as dgr
    assert getattr(dgr._tls, "queue", None) is None
    with dgr.deferred_basic_updates() as outer:
        assert outer == [] and dgr._tls.queue is outer
        with dgr.deferred_basic_updates() as inner:
            assert inner is not outer and dgr._tls.queue is inner
        assert dgr._tls.queue is outer
    assert dgr._tls.queue is None
    # the fused step looks the context up through the name the scene's optimizers imported
    from artdeco_amd import fused
    with fused._deferred_basic_updates() as q:
        assert q == [] and dgr._tls.queue is q
    assert dgr._tls.queue is None


def test_deferred_basic_updates_is_per_thread():
    import threading
    import diff_gaussian_rasterization as dgr
    seen = {}
    with dgr.deferred_basic_updates() as q:
        t = threading.Thread(target=lambda: seen.setdefault("other", getattr(dgr._tls, "queue", None)))
        t.start(); t.join()
        assert dgr._tls.queue is q
    assert seen["other"] is None      # a viewer / frontend thread calling adamUpdateBasic is not captured by the mapper's block


def test_basic_update_still_refuses_cpu_tensors_inside_the_block():
    import diff_gaussian_rasterization as dgr
    p = torch.zeros(4)
    with dgr.deferred_basic_updates() as q:
        with pytest.raises(Exception):
            dgr.adamUpdateBasic(p, torch.ones(4), torch.zeros(4), torch.zeros(4), 1e-3, 0.9, 0.999, 1e-15)
        assert q == []                # validation happens at record time: nothing invalid is queued
