"""curope.rope_2d: oracle vs the reference-generated goldens (CPU) and HIP vs oracle/golden (GPU).
Tolerance 2e-5 absolute on O(1) values: cosf/sinf/powf differ by ulps between libm and the device."""
import os

import numpy as np
import pytest
import torch

from oracle import rope_oracle

from conftest import GOLDEN

CASES = ["rope_vitl", "rope_small"]


def _load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    # golden tokens are [B, heads, N, D] (module layout); the kernel sees the transposed [B,N,H,D]
    return z["tokens"].transpose(0, 2, 1, 3).copy(), z["positions"], z["out"].transpose(0, 2, 1, 3).copy()


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    tok, pos, out = _load(name)
    assert np.abs(rope_oracle.rope_2d_oracle(tok, pos, 100.0, 1.0) - out).max() < 2e-5


def test_oracle_backward_is_inverse_rotation():
    tok, pos, _ = _load("rope_small")
    fwd = rope_oracle.rope_2d_oracle(tok, pos, 100.0, 1.0)
    back = rope_oracle.rope_2d_oracle(fwd, pos, 100.0, -1.0)
    assert np.abs(back - tok).max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_golden(name, dev):
    import curope
    tok, pos, out = _load(name)
    t = torch.from_numpy(tok).to(dev)
    curope.rope_2d(t, torch.from_numpy(pos).to(dev), 100.0, 1.0)
    assert np.abs(t.cpu().numpy() - out).max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,H,D", [(1, 768, 16, 64), (2, 100, 12, 64), (1, 7, 3, 20), (1, 1, 1, 4), (3, 33, 2, 48)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_hip_matches_oracle(B, N, H, D, dtype, dev):
    import curope
    g = torch.Generator().manual_seed(B * 1000 + N)
    tok = torch.randn(B, N, H, D, generator=g)
    pos = torch.randint(0, 32, (B, N, 2), generator=g)
    ref = rope_oracle.rope_2d_oracle(tok.to(dtype).float().numpy(), pos.numpy(), 100.0, 1.0)
    t = tok.to(dtype).to(dev)
    curope.rope_2d(t, pos.to(dev), 100.0, 1.0)
    tol = {torch.float32: 2e-5, torch.float16: 4e-3, torch.bfloat16: 3e-2}[dtype]
    assert np.abs(t.float().cpu().numpy() - ref).max() < tol
    # the autograd wrapper applies fwd = -F0 to the gradient: rotation by -angle undoes it
    curope.rope_2d(t, pos.to(dev), 100.0, -1.0)
    assert np.abs(t.float().cpu().numpy() - tok.to(dtype).float().numpy()).max() < 2 * tol


@pytest.mark.gpu
def test_strided_qkv_view_like_the_reference_wrapper(dev):
    """The reference feeds q = qkv.reshape(B,N,3,H,D).transpose(1,3)[:,:,0] -> [B,H,N,D] and the wrapper
    passes q.transpose(1,2) (curope2d.py:37): dense in (H,D), strided in (B,N).  Must rotate exactly those
    elements in place.  A view that is not dense in (H,D) must raise like kernels.cu:90."""
    import curope
    B, N, H, D = 2, 10, 4, 64
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B, N, 3, H, D, generator=g)
    pos = torch.randint(0, 8, (B, N, 2), generator=g)
    ref = qkv.clone().numpy()
    ref[:, :, 1] = rope_oracle.rope_2d_oracle(ref[:, :, 1], pos.numpy(), 100.0, 1.0)
    dq = qkv.to(dev)
    k_view = dq.transpose(1, 3)[:, :, 1].transpose(1, 2)  # [B,N,H,D] view of the k slice
    assert not k_view.is_contiguous() and k_view.stride(3) == 1 and k_view.stride(2) == D
    curope.rope_2d(k_view, pos.to(dev), 100.0, 1.0)
    assert np.abs(dq.cpu().numpy() - ref).max() < 2e-5       # q and v slices untouched, k rotated
    with pytest.raises(RuntimeError):
        curope.rope_2d(torch.randn(1, 4, 10, 64, device=dev).transpose(1, 2), pos[:1].to(dev), 100.0, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,H,D", [(1, 768, 32, 64), (2, 100, 12, 64), (3, 33, 2, 48), (1, 7, 3, 20)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_cached_table_variant_is_bit_identical(B, N, H, D, dtype, dev):
    """rope_2d_cached (table built once per positions tensor, streaming apply) == rope_2d, bit for bit, on dense tensors and on
    the strided q/k view of a fused qkv projection; a positions tensor edited in place gets a new table."""
    import curope
    g = torch.Generator().manual_seed(N + D)
    pos = torch.randint(0, 32, (B, N, 2), generator=g).to(dev)
    big = torch.randn(B, N, 3 * H * D, generator=g).to(dtype).to(dev)
    for view_of in ("dense", "strided"):
        if view_of == "dense":
            a = torch.randn(B, N, H, D, generator=g).to(dtype).to(dev)
            b = a.clone()
        else:
            a = big.clone().as_strided((B, N, H, D), (N * 3 * H * D, 3 * H * D, D, 1))
            b = big.clone().as_strided((B, N, H, D), (N * 3 * H * D, 3 * H * D, D, 1))
        curope.rope_2d(a, pos, 100.0, 1.0)
        curope.rope_2d_cached(b, pos, 100.0, 1.0)
        curope.rope_2d_cached(b, pos, 100.0, -1.0)   # a second table (fwd = -1) ...
        curope.rope_2d_cached(b, pos, 100.0, 1.0)    # ... and the first one again, from the cache
        curope.rope_2d(a, pos, 100.0, -1.0)
        curope.rope_2d(a, pos, 100.0, 1.0)
        assert torch.equal(a, b)
    pos.add_(1)                                       # same storage, new contents: the version counter invalidates the table
    a = torch.randn(B, N, H, D, generator=g).to(dtype).to(dev)
    b = a.clone()
    curope.rope_2d(a, pos, 100.0, 1.0)
    curope.rope_2d_cached(b, pos, 100.0, 1.0)
    assert torch.equal(a, b)
