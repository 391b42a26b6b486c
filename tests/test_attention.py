"""adk_attention_fwd_f16 (csrc/attention.hip) against the arithmetic of croco/models/blocks.py:105-109 / :150-155:
softmax(q k^T * scale) v, result laid out as `.transpose(1, 2).reshape(B, N, C)`.

CPU: the float64 restatement used as the checker is itself checked against torch's scaled_dot_product_attention.
GPU: the HIP kernel on fp16 inputs against that restatement evaluated on the SAME fp16 values in float64, on
transpose-detecting (asymmetric, position-dependent) data, the shapes the frontend runs (768 tokens, 16 / 12 heads, strided
views of a fused qkv projection, cross attention with different q / k lengths) and ragged sizes."""
import math

import numpy as np
import pytest
import torch


def attention_oracle(q, k, v, scale=None):
    """q [B,H,Nq,D], k / v [B,H,Nk,D] -> [B,Nq,H*D] in float64 (blocks.py:105-109)."""
    q, k, v = q.double(), k.double(), v.double()
    scale = scale if scale is not None else 1.0 / math.sqrt(q.shape[-1])
    attn = (q @ k.transpose(-2, -1)) * scale
    attn = attn.softmax(dim=-1)
    x = attn @ v
    B, H, N, D = x.shape
    return x.transpose(1, 2).reshape(B, N, H * D)


def test_oracle_matches_torch_sdpa():
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(2, 3, 37, 64, generator=g) for _ in range(3))
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(2, 37, 192)
    assert torch.allclose(attention_oracle(q, k, v).float(), ref, atol=2e-6, rtol=1e-5)


def _inputs(B, H, Nq, Nk, seed, spread=1.0):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, H, Nq, 64, generator=g) * spread
    k = torch.randn(B, H, Nk, 64, generator=g) * spread
    # position- and channel-dependent values: a swapped key order, a transposed V tile or a wrong output column all show
    v = torch.randn(B, H, Nk, 64, generator=g) + 0.01 * torch.arange(Nk).float()[None, None, :, None] \
        + 0.1 * torch.arange(64).float()[None, None, None, :]
    return q.half(), k.half(), v.half()


def _check(out, q, k, v, tol=2.5e-3):
    ref = attention_oracle(q.cpu(), k.cpu(), v.cpu())
    err = (out.cpu().double() - ref).abs()
    scale = ref.abs().max()
    assert float(err.max()) <= tol * float(scale), (float(err.max()), float(scale))
    # fp16 output rounding only: the bulk of the entries is within one fp16 ulp of the float64 result
    assert float((err <= 1.2e-3 * ref.abs().clamp_min(0.5)).double().mean()) > 0.999


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 16, 768, 768), (1, 12, 768, 768), (2, 3, 64, 64), (1, 2, 100, 77), (1, 1, 1, 1),
                                       (1, 2, 130, 768), (3, 1, 17, 200)])
def test_attention_matches_oracle(B, H, Nq, Nk):
    from artdeco_amd import attention as att
    dev = torch.device("cuda:0")
    q, k, v = (t.to(dev) for t in _inputs(B, H, Nq, Nk, seed=Nq + Nk))
    out = att.attention(q, k, v)
    assert out.shape == (B, Nq, H * 64) and out.dtype == torch.float16
    _check(out, q, k, v)


@pytest.mark.gpu
def test_attention_on_views_of_the_fused_qkv_projection():
    """The model hands over q, k, v as [B,H,N,D] views of qkv [B,N,3,H,D] (token stride 3 H D, head stride D)."""
    from artdeco_amd import attention as att
    dev = torch.device("cuda:0")
    B, N, H = 1, 768, 16
    g = torch.Generator().manual_seed(5)
    qkv5 = (torch.randn(B, N, 3, H, 64, generator=g) * 1.5).half().to(dev)
    qkv = qkv5.transpose(1, 3)  # [B,H,3,N,D]
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    assert not q.is_contiguous() and att.supported(q, k, v)
    _check(att.attention(q, k, v), q, k, v)


@pytest.mark.gpu
def test_attention_sharp_softmax_and_scale():
    """Large logits (one key dominates per row) and an explicit scale: the running-max rescale path."""
    from artdeco_amd import attention as att
    dev = torch.device("cuda:0")
    q, k, v = (t.to(dev) for t in _inputs(1, 4, 256, 320, seed=9, spread=4.0))
    out = att.attention(q, k, v, scale=0.3)
    ref = attention_oracle(q.cpu(), k.cpu(), v.cpu(), scale=0.3)
    assert float((out.cpu().double() - ref).abs().max()) <= 2.5e-3 * float(ref.abs().max())


@pytest.mark.gpu
def test_attention_rejects_what_it_cannot_run():
    from artdeco_amd import _lib, attention as att
    dev = torch.device("cuda:0")
    q = torch.zeros(1, 2, 8, 64, device=dev)  # float32
    assert not att.supported(q, q, q)
    with pytest.raises(_lib.AdkError):
        att.attention(q, q, q)
    h = torch.zeros(1, 2, 8, 32, device=dev, dtype=torch.float16)
    assert not att.supported(h, h, h)


@pytest.mark.gpu
def test_model_blocks_use_the_kernel_and_match_sdpa():
    """Attention / CrossAttention modules of the restated model: fp16 on the GPU goes through the HIP kernel and agrees with
    the same modules evaluated through torch's scaled_dot_product_attention."""
    from artdeco_amd import mast3r_model as mm
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    rope = mm.RoPE2D(100.0)
    blk = mm.Attention(1024, rope, 16).to(dev).half().eval()
    cross = mm.CrossAttention(768, rope, 12).to(dev).half().eval()
    x = torch.randn(1, 768, 1024, device=dev).half()
    y = torch.randn(1, 768, 768, device=dev).half()
    ys, xs = torch.meshgrid(torch.arange(24, device=dev), torch.arange(32, device=dev), indexing="ij")
    pos = torch.stack([ys.reshape(-1), xs.reshape(-1)], -1)[None]
    calls = []
    from artdeco_amd import attention as att
    orig = att.attention
    att.attention = lambda *a, **kw: (calls.append(1), orig(*a, **kw))[1]
    try:
        with torch.no_grad():
            a1, c1 = blk(x, pos), cross(y, y.flip(1), y.flip(1), pos, pos)
    finally:
        att.attention = orig
    assert len(calls) == 2
    sup = att.supported
    att.supported = lambda *a: False
    try:
        with torch.no_grad():
            a0, c0 = blk(x, pos), cross(y, y.flip(1), y.flip(1), pos, pos)
    finally:
        att.supported = sup
    for new, old in ((a1, a0), (c1, c0)):
        assert float((new.float() - old.float()).abs().max()) <= 4e-3 * float(old.float().abs().max())


def test_cpu_tensors_take_the_torch_path_and_the_kernel_refuses_them():
    """No GPU: `supported` says no, `attention` fails loudly (no CPU fallback inside the operator), the model's `_attend` returns
    torch's scaled_dot_product_attention in the [B, N, H*D] layout of blocks.py:109."""
    from artdeco_amd import _lib, attention as att
    from artdeco_amd.mast3r_model import _attend
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(2, 3, 10, 64, generator=g) for _ in range(3))
    assert not att.supported(q, k, v) and not att.supported(q.half(), k.half(), v.half())
    with pytest.raises(_lib.AdkError):
        att.attention(q.half(), k.half(), v.half())
    out = _attend(q, k, v)
    assert out.shape == (2, 10, 192)
    assert torch.allclose(out.double(), attention_oracle(q, k, v), atol=2e-6, rtol=1e-5)
