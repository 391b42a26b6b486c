"""torch.linalg.inv / torch.inverse / Tensor.inverse of 4x4 fp32 matrices on the GPU without torch's LU + `info` read-back -- for
run_system.py's SLAM-keyframe loop and nothing else.

run_system.py:194-227 re-reads every mapper keyframe's pose on a SLAM keyframe and, per keyframe, inverts three 4x4 matrices
(`view_matrix.detach().inverse()` :221, `torch.linalg.inv(old_Rt)` :222, `torch.linalg.inv(new_Rt)` :223); h3dgsv3.py:1000 inverts one more
per `add_keyframe`.  On the device each of those is a batched LU, a solve and a blocking read of `info` (torch raises on a singular
input): ~190 us and a host synchronisation per call, 27.6 ms per SLAM keyframe at 48 keyframes (DESIGN findings 43, 48).  The loop lives in
ARTDECO's script, which the scene-model hooks do not reach; the operator it calls is torch's.  `install()` therefore wraps the three entry
points -- and (round 6) takes the fast path ONLY when

  * the CALLER is one of ARTDECO's own two files that invert poses (`run_system.py`, `h3dgsv3.py`; `ALLOWED_CALLERS`, matched on the calling
    frame's file name -- the bench's mirror of that loop registers itself with `allow_caller`), so that pypose, kornia, the viewers, the pose
    initialiser and any third party in the process keep torch's own functions, error behaviour included
    (`ARTDECO_AMD_FAST_INV4=all` lifts the restriction, `=0` leaves torch alone altogether), and
  * the argument is a plain CUDA float32 tensor of shape [..., 4, 4] (no `out=`, no tensor subclass).

An argument that requires grad (run_system.py:222: `old_Rt = frame.get_Rt()` hangs off the keyframe's pose parameters, and the loop runs
with grad enabled) goes through `_Inv4x4` -- the same launch with the textbook backward -A^-T G A^-T -- so all THREE inversions of a keyframe
take one launch each in the script as written (round 5 sent that one to torch's LU).

Singular input: there is no read-back in the call, so the result is NaN-filled there and then -- but the kernel bumps a counter in host-mapped
pinned memory, and `check()` raises `torch.linalg.LinAlgError` at the process's next host wait (`native_step.train_on_keyframe` calls it behind
`adk_mapper_step`'s own wait, the per-stage rasteriser behind its intersection-count wait): within one optimisation step of the inversion.
The exception arrives late, never not at all.
`uninstall()` restores torch's functions.  Installed by `fused.patch_scene_model` (with the `pose` pin group verified), never by importing
this package.
"""
from __future__ import annotations

import os
import sys

import torch

from . import _lib

_ORIG: dict = {}
STATS = {"fast": 0, "fast_grad": 0, "torch": 0, "foreign_caller": 0}
ALLOWED_CALLERS = {"run_system.py", "h3dgsv3.py"}
_ANY_CALLER = False
_COUNTER: dict = {}       # {"t": pinned int32[1], "seen": int}


def allow_caller(filename: str) -> None:
    """Let calls made from a file of this base name take the fast path (the bench's mirror of run_system.py's loop: harness/stream.py)."""
    ALLOWED_CALLERS.add(os.path.basename(filename))


def _counter() -> torch.Tensor:
    if "t" not in _COUNTER:
        _COUNTER["t"] = torch.zeros(1, dtype=torch.int32).pin_memory()     # host-mapped: the kernel's system-scope atomic lands here
        _COUNTER["np"] = _COUNTER["t"].numpy()                             # the same memory: check() reads it without building a tensor
        _COUNTER["seen"] = 0
    return _COUNTER["t"]


def singular_seen() -> int:
    """Singular matrices met by the fast path since the process started (a plain read of pinned memory: exact after any wait on the stream)."""
    return int(_COUNTER["np"][0]) if "t" in _COUNTER else 0


def check() -> None:
    """Raise torch.linalg.LinAlgError if a fast-path inversion met a singular matrix since the last check.  Costs one read of a pinned int;
    call it behind a host wait that already exists (it does not synchronise)."""
    if "t" not in _COUNTER:
        return
    n = int(_COUNTER["np"][0])
    if n > _COUNTER["seen"]:
        new = n - _COUNTER["seen"]
        _COUNTER["seen"] = n
        raise torch.linalg.LinAlgError(f"artdeco_amd.small_inverse: {new} singular 4x4 matri{'x' if new == 1 else 'ces'} inverted since the last "
                                       "host wait (their results are NaN; torch.linalg.inv would have raised at the call -- "
                                       "ARTDECO_AMD_FAST_INV4=0 restores that)")


def _plain(A) -> bool:
    return (type(A) in (torch.Tensor, torch.nn.Parameter) and A.is_cuda and A.dtype == torch.float32 and A.dim() >= 2
            and A.shape[-1] == 4 and A.shape[-2] == 4 and A.numel() > 0)


def _caller_ok() -> bool:
    if _ANY_CALLER:
        return True
    try:
        f = sys._getframe(2)      # 0 = here, 1 = the wrapper, 2 = whoever called torch.linalg.inv / torch.inverse / Tensor.inverse
    except ValueError:            # called from the top of a stack (an embedding interpreter): nobody we know
        return False
    return os.path.basename(f.f_code.co_filename) in ALLOWED_CALLERS


def inv4x4(A: torch.Tensor, info: torch.Tensor | None = None, count: bool = True) -> torch.Tensor:
    """Inverse of every 4x4 matrix of A [..., 4, 4] (CUDA, float32, any strides): one launch, no synchronisation.  `info` (int32, one entry
    per matrix) receives 0, or 1 + the column at which a matrix turned out singular (its result is NaN); `count`: singular matrices are also
    counted for `check()`."""
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
        rc = lib.adk_inv4x4(src.data_ptr(), out.data_ptr(), n, sb, sr, sc, info.data_ptr() if info is not None else None,
                            _counter().data_ptr() if count else None, _lib.raw_stream(dev))
    _lib.check(rc, "adk_inv4x4")
    return out


class _Inv4x4(torch.autograd.Function):
    """inv4x4 for an argument that takes part in autograd: d(A^-1) = -A^-1 dA A^-1  =>  grad_A = -A^-T G A^-T."""

    @staticmethod
    def forward(ctx, A):
        out = inv4x4(A)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        ot = out.transpose(-1, -2)
        return -(ot @ g @ ot)


def _fast(A):
    if A.requires_grad and torch.is_grad_enabled():
        STATS["fast_grad"] += 1
        return _Inv4x4.apply(A)
    STATS["fast"] += 1
    return inv4x4(A)


def _linalg_inv(A, *args, **kwargs):
    if not args and not kwargs and _plain(A):
        if _caller_ok():
            return _fast(A)
        STATS["foreign_caller"] += 1
    STATS["torch"] += 1
    return _ORIG["linalg.inv"](A, *args, **kwargs)


def _torch_inverse(input, *args, **kwargs):      # noqa: A002 (torch's own parameter name)
    if not args and not kwargs and _plain(input):
        if _caller_ok():
            return _fast(input)
        STATS["foreign_caller"] += 1
    STATS["torch"] += 1
    return _ORIG["inverse"](input, *args, **kwargs)


def _tensor_inverse(self, *args, **kwargs):
    if not args and not kwargs and _plain(self):
        if _caller_ok():
            return _fast(self)
        STATS["foreign_caller"] += 1
    STATS["torch"] += 1
    return _ORIG["Tensor.inverse"](self, *args, **kwargs)


def installed() -> bool:
    return bool(_ORIG)


def install(force: bool = False, any_caller: bool | None = None) -> bool:
    """Wrap torch.linalg.inv, torch.inverse and Tensor.inverse (idempotent).  False when ARTDECO_AMD_FAST_INV4=0 (unless `force`).
    `any_caller` (default: ARTDECO_AMD_FAST_INV4=all) drops the restriction to ARTDECO's own files."""
    global _ANY_CALLER
    env = os.environ.get("ARTDECO_AMD_FAST_INV4", "1")
    if any_caller is not None:
        _ANY_CALLER = bool(any_caller)
    elif not _ORIG:
        _ANY_CALLER = env == "all"
    if _ORIG:
        return True
    if not force and env == "0":
        return False
    _ORIG["linalg.inv"], _ORIG["inverse"], _ORIG["Tensor.inverse"] = torch.linalg.inv, torch.inverse, torch.Tensor.inverse
    torch.linalg.inv, torch.inverse, torch.Tensor.inverse = _linalg_inv, _torch_inverse, _tensor_inverse
    return True


def uninstall() -> None:
    if not _ORIG:
        return
    torch.linalg.inv, torch.inverse, torch.Tensor.inverse = _ORIG["linalg.inv"], _ORIG["inverse"], _ORIG["Tensor.inverse"]
    _ORIG.clear()
