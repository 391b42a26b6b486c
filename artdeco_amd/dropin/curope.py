"""Drop-in for the `curope` extension module (croco/models/curope/curope.cpp:67-69): `rope_2d`.

Picked up by the reference's own wrapper `import curope as _kernels` (curope2d.py:6-9), which makes
MASt3R's RoPE2D run on the HIP kernel instead of the slow torch fallback (pos_embed.py:106-158).
"""
from __future__ import annotations

import torch

from artdeco_amd import _lib


_DTYPES = {torch.float16: 0, torch.float32: 1, torch.bfloat16: 2}


def rope_2d(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    """In place.  tokens [B,N,H,D] (float32/float16, D-contiguous, head stride D), positions [B,N,2] int64."""
    if tokens.dim() != 4:
        raise RuntimeError("tokens must have 4 dimensions")
    if positions.dim() != 3:
        raise RuntimeError("positions must have 3 dimensions")
    if tokens.size(0) != positions.size(0):
        raise RuntimeError("batch size differs between tokens & positions")
    if tokens.size(1) != positions.size(1):
        raise RuntimeError("seq_length differs between tokens & positions")
    if positions.size(2) != 2:
        raise RuntimeError("positions.shape[2] must be equal to 2")
    _lib.require_cuda(tokens, positions)
    B, N, H, D = tokens.shape
    # same layout contract as kernels.cu:90: only the last two dims must be dense; the wrapper passes
    # q/k as a transposed view of the fused qkv projection, so batch/token strides are arbitrary
    if not (tokens.stride(3) == 1 and tokens.stride(2) == D):
        raise RuntimeError("tokens are not contiguous")
    if not positions.is_contiguous():
        raise RuntimeError("positions are not contiguous")
    if D % 4 != 0:
        raise RuntimeError("token dim must be multiple of 4")
    if positions.dtype != torch.int64:
        raise TypeError("positions must be int64")
    if tokens.dtype not in _DTYPES:
        raise TypeError("rope_2d supports float32, float16 and bfloat16 tokens")
    lib = _lib.load()
    with torch.cuda.device(tokens.device):
        rc = lib.adk_rope_2d(tokens.data_ptr(), positions.data_ptr(), _DTYPES[tokens.dtype],
                             B, N, tokens.stride(0), tokens.stride(1), H, D, float(base), float(fwd), _lib.stream_of(tokens))
    _lib.check(rc, "adk_rope_2d")


# ---------------------------------------------------------------------------------------------------------------------------
# Not part of the reference's extension: the rotation with its trigonometry cached.  Every block of the model rotates q and k by
# the same positions tensor (croco/models/blocks.py hands `xpos` to all of them), so the (cos, sin) table of a positions tensor
# is built once (adk_rope_2d_table: the fp32 expressions of kernels.cu:38-44) and each call streams through it
# (adk_rope_2d_apply): bit-identical to rope_2d, 7.2 -> 2.7 us per call at 768 tokens.
_TABLES: "dict[tuple, tuple]" = {}
_TABLES_MAX = 8


def _table_for(positions: torch.Tensor, D: int, base: float, fwd: float) -> torch.Tensor:
    # tensors created under torch.inference_mode() carry no version counter: an in-place edit of such a positions tensor is not seen
    version = -1 if positions.is_inference() else positions._version
    key = (positions.data_ptr(), version, tuple(positions.shape), positions.device, D, float(base), float(fwd))
    hit = _TABLES.get(key)
    if hit is not None:
        # built on another stream (the model runs its two decoder branches on two streams): the first use on this one waits for it
        cur = torch.cuda.current_stream(positions.device)
        if hit[2] is not None and cur.cuda_stream not in hit[3] and not torch.cuda.is_current_stream_capturing():
            cur.wait_event(hit[2])
            hit[3].add(cur.cuda_stream)
        return hit[1]
    B, N, _ = positions.shape
    table = torch.empty(B * N, 2, D // 4, 2, dtype=torch.float32, device=positions.device)
    lib = _lib.load()
    with torch.cuda.device(positions.device):
        rc = lib.adk_rope_2d_table(positions.data_ptr(), B * N, D, float(base), float(fwd), table.data_ptr(), _lib.stream_of(positions))
    _lib.check(rc, "adk_rope_2d_table")
    while len(_TABLES) >= _TABLES_MAX:
        _TABLES.pop(next(iter(_TABLES)))
    built_on = torch.cuda.current_stream(positions.device)
    ready = None
    if not torch.cuda.is_current_stream_capturing():
        ready = torch.cuda.Event()
        ready.record(built_on)
    # the positions tensor is kept alive with its table (its address cannot be reused meanwhile); `ready` + the set of streams that
    # have already ordered themselves behind it
    _TABLES[key] = (positions, table, ready, {built_on.cuda_stream})
    return table


def rope_2d_cached(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    """rope_2d(tokens, positions, base, fwd) through the cached table; falls back to rope_2d for layouts the streaming kernel
    does not take (head dim not a multiple of 16, strides not multiples of 4 elements)."""
    B, N, H, D = tokens.shape
    if (tokens.dim() != 4 or not tokens.is_cuda or tokens.dtype not in _DTYPES or D % 16 or tokens.stride(3) != 1 or tokens.stride(2) != D
            or tokens.stride(0) % 4 or tokens.stride(1) % 4 or tokens.data_ptr() % (4 * tokens.element_size())
            or positions.dtype != torch.int64 or not positions.is_contiguous() or tuple(positions.shape) != (B, N, 2)):
        return rope_2d(tokens, positions, base, fwd)
    table = _table_for(positions, D, base, fwd)
    lib = _lib.load()
    with torch.cuda.device(tokens.device):
        rc = lib.adk_rope_2d_apply(tokens.data_ptr(), table.data_ptr(), _DTYPES[tokens.dtype], B, N, tokens.stride(0), tokens.stride(1),
                                   H, D, _lib.stream_of(tokens))
    _lib.check(rc, "adk_rope_2d_apply")
