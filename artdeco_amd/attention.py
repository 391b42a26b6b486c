"""softmax(q k^T / sqrt(D)) v for the MASt3R blocks on the hand-written gfx950 kernel (csrc/attention.hip).

Arithmetic of croco/models/blocks.py:105-109 / :150-155 (the reference materialises the N x N matrix in fp32/TF32; here the
operands are fp16 and scores, softmax and accumulation fp32 -- the model's "TF32-class" mode).  Returns the result already in
the [B, N, H*D] layout blocks.py:109 produces with `.transpose(1, 2).reshape(B, N, C)`."""
from __future__ import annotations

import ctypes
import math

import torch

from artdeco_amd import _lib


def supported(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> bool:
    """fp16 CUDA tensors [B,H,N,64] whose last dim is dense and whose other strides / base addresses allow 128-bit loads."""
    for t in (q, k, v):
        if not (t.is_cuda and t.dtype == torch.float16 and t.dim() == 4 and t.shape[-1] == 64 and t.stride(3) == 1):
            return False
        if any(s % 8 for s in t.stride()[:3]) or t.data_ptr() % 16:
            return False
    return q.shape[:2] == k.shape[:2] == v.shape[:2] and k.shape[2] == v.shape[2] and q.shape[2] > 0 and k.shape[2] > 0


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float | None = None) -> torch.Tensor:
    """q [B,H,Nq,64], k / v [B,H,Nk,64] (any views accepted by `supported`) -> [B, Nq, H*64] float16."""
    if not supported(q, k, v):
        raise _lib.AdkError("adk_attention_fwd_f16: float16 CUDA tensors [B,H,N,64] with a dense last dim, strides that are "
                            "multiples of 8 elements and 16-byte aligned storage are required")
    B, H, Nq, D = q.shape
    Nk = k.shape[2]
    out = torch.empty(B, Nq, H * D, dtype=torch.float16, device=q.device)
    lib = _lib.load()
    i3 = ctypes.c_int64 * 3
    with torch.cuda.device(q.device):
        rc = lib.adk_attention_fwd_f16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, Nq, Nk,
                                       i3(*q.stride()[:3]), i3(*k.stride()[:3]), i3(*v.stride()[:3]),
                                       float(scale if scale is not None else 1.0 / math.sqrt(D)), _lib.stream_of(q))
    _lib.check(rc, "adk_attention_fwd_f16")
    return out
