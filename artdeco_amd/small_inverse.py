"""torch.linalg.inv / torch.inverse / Tensor.inverse of 4x4 fp32 matrices on the GPU without torch's LU + `info` read-back.

run_system.py:194-227 re-reads every mapper keyframe's pose on a SLAM keyframe and, per keyframe, inverts three 4x4 matrices
(`view_matrix.detach().inverse()` :221, `torch.linalg.inv(old_Rt)` :222, `torch.linalg.inv(new_Rt)` :223); h3dgsv3.py:1000 inverts one more
per `add_keyframe`.  On the device each of those is a batched LU, a solve and a blocking read of `info` (torch raises on a singular
input): ~190 us and a host synchronisation per call, 27.6 ms per SLAM keyframe at 48 keyframes -- a tenth of the headline's frame time
and a third of a 1 000-frame sequence (DESIGN findings 43, 48).  The loop lives in ARTDECO's script, which the scene-model hooks do not
reach; the operator it calls is torch's.  `install()` therefore wraps the three entry points: a call whose argument is a plain CUDA
float32 tensor of shape [..., 4, 4] that does not take part in autograd goes to ONE launch of `adk_inv4x4` (Gauss-Jordan with partial
pivoting, fp64 inside, rounded once: at least as accurate as the LU it replaces); every other call -- CPU tensors, other dtypes or sizes,
tensor subclasses (pypose's LieTensor), `out=`, inputs that require grad -- goes to torch's own function, untouched.

One deviation, by construction: there is no read-back, so a SINGULAR 4x4 gives a NaN-filled result instead of `torch.linalg.LinAlgError`
(the behaviour of `torch.linalg.inv_ex` without `check_errors`).  `ARTDECO_AMD_FAST_INV4=0` keeps torch's functions; `uninstall()` restores
them.  Installed by `fused.patch_scene_model` (i.e. with the other drop-ins), never by importing this package.
"""
from __future__ import annotations

import os

import torch

from . import _lib

_ORIG: dict = {}
STATS = {"fast": 0, "torch": 0}


def _eligible(A) -> bool:
    return (type(A) in (torch.Tensor, torch.nn.Parameter) and A.is_cuda and A.dtype == torch.float32 and A.dim() >= 2
            and A.shape[-1] == 4 and A.shape[-2] == 4 and A.numel() > 0 and not (A.requires_grad and torch.is_grad_enabled()))


def inv4x4(A: torch.Tensor, info: torch.Tensor | None = None) -> torch.Tensor:
    """Inverse of every 4x4 matrix of A [..., 4, 4] (CUDA, float32, any strides): one launch, no synchronisation.  `info` (int32, one entry
    per matrix) receives 0, or 1 + the column at which a matrix turned out singular (its result is NaN)."""
    lib = _lib.load()
    _lib.require_cuda(A)
    if A.dtype != torch.float32 or A.dim() < 2 or A.shape[-2:] != (4, 4):
        raise ValueError("inv4x4: float32 [..., 4, 4] expected")
    dev = A.device
    src = A.detach()
    if src.dim() == 2:
        n, (sr, sc), sb = 1, src.stride(), 0
    else:
        src = src.reshape(-1, 4, 4)          # a view where the batch dimensions collapse, a copy otherwise
        n, (sb, sr, sc) = src.shape[0], src.stride()
    if info is not None and (info.dtype != torch.int32 or info.numel() != n or not info.is_contiguous() or info.device != dev):
        raise ValueError("inv4x4: info must be a contiguous int32 tensor with one entry per matrix on the input's device")
    with _lib.on_device(dev):
        out = torch.empty(A.shape, dtype=torch.float32, device=dev)
        rc = lib.adk_inv4x4(src.data_ptr(), out.data_ptr(), n, sb, sr, sc, info.data_ptr() if info is not None else None, _lib.raw_stream(dev))
    _lib.check(rc, "adk_inv4x4")
    return out


def _linalg_inv(A, *args, **kwargs):
    if not args and not kwargs and _eligible(A):
        STATS["fast"] += 1
        return inv4x4(A)
    STATS["torch"] += 1
    return _ORIG["linalg.inv"](A, *args, **kwargs)


def _torch_inverse(input, *args, **kwargs):      # noqa: A002 (torch's own parameter name)
    if not args and not kwargs and _eligible(input):
        STATS["fast"] += 1
        return inv4x4(input)
    STATS["torch"] += 1
    return _ORIG["inverse"](input, *args, **kwargs)


def _tensor_inverse(self, *args, **kwargs):
    if not args and not kwargs and _eligible(self):
        STATS["fast"] += 1
        return inv4x4(self)
    STATS["torch"] += 1
    return _ORIG["Tensor.inverse"](self, *args, **kwargs)


def installed() -> bool:
    return bool(_ORIG)


def install(force: bool = False) -> bool:
    """Wrap torch.linalg.inv, torch.inverse and Tensor.inverse (idempotent).  False when ARTDECO_AMD_FAST_INV4=0 (unless `force`)."""
    if _ORIG:
        return True
    if not force and os.environ.get("ARTDECO_AMD_FAST_INV4", "1") == "0":
        return False
    _ORIG["linalg.inv"], _ORIG["inverse"], _ORIG["Tensor.inverse"] = torch.linalg.inv, torch.inverse, torch.Tensor.inverse
    torch.linalg.inv, torch.inverse, torch.Tensor.inverse = _linalg_inv, _torch_inverse, _tensor_inverse
    return True


def uninstall() -> None:
    if not _ORIG:
        return
    torch.linalg.inv, torch.inverse, torch.Tensor.inverse = _ORIG["linalg.inv"], _ORIG["inverse"], _ORIG["Tensor.inverse"]
    _ORIG.clear()
