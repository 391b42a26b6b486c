"""adk_add_layernorm (csrc/layernorm.hip): `x = x + delta; y = LayerNorm(x)` of croco/models/blocks.py:88-95 / :176-191 in
one launch, and the model paths built on it (artdeco_amd/mast3r_model.py: _encode_image / _decoder_fused) against the same
model run sub-layer by sub-layer through torch."""
import os

import pytest
import torch


def _ref(x, delta, norm, out_f16):
    xs = x if delta is None else x + delta.float()
    y = torch.nn.functional.layer_norm(xs.double(), norm.normalized_shape, norm.weight.double(), norm.bias.double(), norm.eps)
    return xs, y


@pytest.mark.gpu
@pytest.mark.parametrize("rows,C", [(768, 1024), (768, 768), (12, 48), (5, 64), (3, 2048), (1537, 100), (1, 4)])
@pytest.mark.parametrize("with_delta", [False, True])
@pytest.mark.parametrize("out_f16", [True, False])
def test_add_layernorm_matches_torch(rows, C, with_delta, out_f16):
    from artdeco_amd.fused_norm import add_layernorm
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows * 7 + C)
    x = (torch.randn(2, rows, C, generator=g) * 3 + 0.5 * torch.arange(C).float() / C).to(dev)
    delta = (torch.randn(2, rows, C, generator=g) * 2).half().to(dev) if with_delta else None
    norm = torch.nn.LayerNorm(C, eps=1e-6).to(dev)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.3 * torch.randn(C, generator=g).to(dev))
        norm.bias.copy_(0.2 * torch.randn(C, generator=g).to(dev))
    x0 = x.clone()
    xs, y = add_layernorm(x, delta, norm, out_f16=out_f16)
    assert torch.equal(x, x0)                                   # the input is never modified
    rx, ry = _ref(x0, delta, norm, out_f16)
    assert y.dtype == (torch.float16 if out_f16 else torch.float32)
    if with_delta:
        assert torch.equal(xs, rx)                              # fp32 add of an exactly converted fp16: bit-identical
    else:
        assert xs is x
    err = (y.double() - ry).abs()
    tol = (1e-3 if out_f16 else 2e-6) * ry.abs().clamp_min(1.0)  # one fp16 ulp / a few fp32 ulps of the float64 result
    assert bool((err <= tol).all()), float((err / ry.abs().clamp_min(1.0)).max())


@pytest.mark.gpu
def test_add_layernorm_rejects_what_it_cannot_run():
    from artdeco_amd import _lib
    from artdeco_amd.fused_norm import add_layernorm, supported
    dev = torch.device("cuda:0")
    norm = torch.nn.LayerNorm(64).to(dev)
    x = torch.zeros(4, 64, device=dev)
    assert supported(x, None, norm)
    assert not supported(x.half(), None, norm) and not supported(x, x, norm) and not supported(x[:, ::2], None, torch.nn.LayerNorm(32).to(dev))
    with pytest.raises(_lib.AdkError):
        add_layernorm(x.half(), None, norm)


@pytest.mark.gpu
def test_fused_trunk_matches_the_unfused_trunk():
    """ViT-L widths (1024 / 16 heads encoder, 768 / 12 heads decoder: the HIP attention kernel's shapes) at reduced depth, TF32-class
    mode: the fused path (one launch per residual add + LayerNorm + cast, HIP attention) against the same weights run through the
    unfused torch path (ADK_MAST3R_FUSED_NORM=0, scaled_dot_product_attention)."""
    from artdeco_amd import attention as att
    from artdeco_amd.mast3r_model import AsymmetricMASt3R
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = AsymmetricMASt3R(img_size=(512, 512), enc_embed_dim=1024, enc_depth=2, enc_num_heads=16, dec_embed_dim=768, dec_depth=12,
                           dec_num_heads=12).to(dev).eval().to_inference_dtype(torch.float16, fp32_stream=True)
    img1, img2 = (torch.rand(1, 3, 384, 512, device=dev) * 2 - 1 for _ in range(2))
    shp = torch.tensor([[384, 512]])

    def run():
        with torch.inference_mode():
            f1, p1, _ = net._encode_image(img1, shp)
            f2, p2, _ = net._encode_image(img2, shp)
            d1, d2 = net._decoder(f1, p1, f2, p2)
            return [f1, f2] + list(d1) + list(d2)

    calls = []
    orig = att.attention
    att.attention = lambda *a, **kw: (calls.append(1), orig(*a, **kw))[1]
    try:
        fused = run()
    finally:
        att.attention = orig
    assert len(calls) == 2 * 2 + 12 * 2 * 2          # every self / cross attention went through the HIP kernel
    sup = att.supported
    att.supported = lambda *a: False
    os.environ["ADK_MAST3R_FUSED_NORM"] = "0"
    try:
        plain = run()
    finally:
        att.supported = sup
        del os.environ["ADK_MAST3R_FUSED_NORM"]
    assert len(fused) == len(plain) == 2 + 2 * 13
    for a, b in zip(fused, plain):
        assert a.shape == b.shape and a.dtype == b.dtype == torch.float32
        assert float((a - b).abs().max()) <= 3e-3 * float(b.abs().max()), float((a - b).abs().max() / b.abs().max())


def test_cpu_tensors_are_refused():
    from artdeco_amd import _lib
    from artdeco_amd.fused_norm import add_layernorm, supported
    norm = torch.nn.LayerNorm(64)
    x = torch.zeros(4, 64)
    assert not supported(x, None, norm)
    with pytest.raises(_lib.AdkError):
        add_layernorm(x, None, norm)
