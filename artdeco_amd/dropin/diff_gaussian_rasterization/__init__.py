"""Drop-in for `diff_gaussian_rasterization` (on-the-fly-nvs fork [UPSTREAM, not vendored]).

Symbols ARTDECO imports:
  * adamUpdate, adamUpdateBasic           -- Reconstruct/scene/optimizers.py:14
  * GaussianRasterizationSettings,
    GaussianRasterizer                    -- Reconstruct/webviewer/scene_models.py:33-36

All of them run on the HIP kernels of libartdeco_hip.so; there is no CPU path.
"""
from __future__ import annotations

import contextlib
import threading

import torch

from artdeco_amd import _lib
from artdeco_amd import autoinstall as _autoinstall

_autoinstall.on_dropin_import()  # post-import hook: fused mapper paths on every SceneModel (ARTDECO_AMD_AUTOFUSE=0 disables)

from ._rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401


def _check_adam(param, grad, exp_avg, exp_avg_sq):
    _lib.require_cuda(param, grad, exp_avg, exp_avg_sq)
    for t, n in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        if t.dtype != torch.float32:
            raise TypeError(f"adamUpdate: {n} must be float32")
        if t.numel() != param.numel():
            raise ValueError(f"adamUpdate: {n} has {t.numel()} elements, param has {param.numel()}")
    for t, n in ((param, "param"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        if not t.is_contiguous():
            raise ValueError(f"adamUpdate: {n} must be contiguous (it is updated in place)")


@torch.no_grad()
def adamUpdate(param, param_grad, exp_avg, exp_avg_sq, visible, lr, b1, b2, eps, N, M):
    """In-place sparse Adam: only rows with visible[row] are touched (optimizers.py:116-128,144-156).

    lr: 0-dim tensor, [N] tensor or tensor with param.numel() elements (per-Gaussian lr of
    optimizers.py:212-219 / h3dgsv3.py:1240-1247).  No bias correction.
    """
    _check_adam(param, param_grad, exp_avg, exp_avg_sq)
    N, M = int(N), int(M)
    if N * M != param.numel():
        raise ValueError(f"adamUpdate: N*M = {N * M} != param.numel() = {param.numel()}")
    if N == 0 or M == 0:
        return
    _lib.require_cuda(visible)
    if visible.numel() != N:
        raise ValueError(f"adamUpdate: visible has {visible.numel()} entries, expected N = {N}")
    if visible.dtype != torch.bool:
        visible = visible != 0
    visible = visible.contiguous()
    if not torch.is_tensor(lr):
        lr = torch.tensor(float(lr), dtype=torch.float32, device=param.device)
    lr = lr.to(device=param.device, dtype=torch.float32).contiguous()
    if lr.numel() not in (1, N, N * M):
        raise ValueError(f"adamUpdate: lr has {lr.numel()} elements; expected 1, N or N*M")
    grad = param_grad.contiguous()
    lib = _lib.load()
    with _lib.on_device(param.device):
        rc = lib.adk_adam_update(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                 visible.data_ptr(), lr.data_ptr(), lr.numel(), float(b1), float(b2), float(eps),
                                 N, M, _lib.stream_of(param))
    _lib.check(rc, "adk_adam_update")


_tls = threading.local()


def _launch_basic(param, grad, exp_avg, exp_avg_sq, lr, b1, b2, eps):
    lib = _lib.load()
    with _lib.on_device(param.device):
        rc = lib.adk_adam_update_basic(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                       float(lr), float(b1), float(b2), float(eps), param.numel(),
                                       _lib.stream_of(param))
    _lib.check(rc, "adk_adam_update_basic")


@contextlib.contextmanager
def deferred_basic_updates():
    """Inside the block (on this thread) adamUpdateBasic validates and RECORDS its arguments instead of launching; the
    block yields the list of (param, grad, exp_avg, exp_avg_sq, lr, b1, b2, eps) records.  A caller that is about to
    launch a multi-tensor update anyway (artdeco_amd.fused: the Gaussians' step right after Keyframe.step(), whose three
    calls are 6, 3 and 12 floats) takes the records out of the list and packs them into that launch.  Whatever is
    still in the list when the block ends is launched then, one kernel each, so no update can be lost."""
    queue: list = []
    prev = getattr(_tls, "queue", None)
    _tls.queue = queue
    try:
        yield queue
    finally:
        _tls.queue = prev
        for rec in queue:
            _launch_basic(*rec)
        queue.clear()


@torch.no_grad()
def adamUpdateBasic(param, param_grad, exp_avg, exp_avg_sq, lr, b1, b2, eps):
    """In-place dense Adam with a python-float lr (optimizers.py:48-57, :90-99)."""
    _check_adam(param, param_grad, exp_avg, exp_avg_sq)
    if param.numel() == 0:
        return
    grad = param_grad.contiguous()
    rec = (param, grad, exp_avg, exp_avg_sq, float(lr), float(b1), float(b2), float(eps))
    queue = getattr(_tls, "queue", None)
    if queue is not None:
        queue.append(rec)
        return
    _launch_basic(*rec)
