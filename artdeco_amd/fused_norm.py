"""`x = x + delta; y = LayerNorm(x)` with y rounded once to the GEMM operand type, as ONE launch (csrc/layernorm.hip).

The glue between the sub-layers of croco/models/blocks.py:88-95 / :176-191 in the model's TF32-class mode (fp32 residual stream,
fp16 GEMM operands): the reference's three torch kernels per sub-layer (add, layer_norm, cast) become one."""
from __future__ import annotations

import torch

from artdeco_amd import _lib


def supported(x: torch.Tensor, delta, norm: torch.nn.LayerNorm) -> bool:
    C = x.shape[-1]
    ok = (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and C % 4 == 0 and C <= 2048
          and norm.weight is not None and norm.bias is not None and norm.weight.dtype == torch.float32
          and tuple(norm.normalized_shape) == (C,))
    if delta is not None:
        ok = ok and delta.is_cuda and delta.dtype == torch.float16 and delta.is_contiguous() and delta.shape == x.shape
    return bool(ok)


def add_layernorm(x: torch.Tensor, delta, norm: torch.nn.LayerNorm, out_f16: bool = True):
    """-> (x + delta [a NEW float32 tensor; x itself when delta is None], LayerNorm(x + delta) as float16 / float32)."""
    if not supported(x, delta, norm):
        raise _lib.AdkError("adk_add_layernorm: contiguous float32 CUDA activations (+ float16 delta) and a float32 affine "
                            "LayerNorm over the last dim (C % 4 == 0, C <= 2048) are required")
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty(x.shape, dtype=torch.float16 if out_f16 else torch.float32, device=x.device)
    x_out = torch.empty_like(x) if delta is not None else x
    lib = _lib.load()
    with torch.cuda.device(x.device):
        rc = lib.adk_add_layernorm(x.data_ptr(), _lib.ptr(delta), norm.weight.data_ptr(), norm.bias.data_ptr(), float(norm.eps),
                                   rows, C, x_out.data_ptr() if delta is not None else None, y.data_ptr(), int(out_f16),
                                   _lib.stream_of(x))
    _lib.check(rc, "adk_add_layernorm")
    return x_out, y
