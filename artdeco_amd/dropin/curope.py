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
