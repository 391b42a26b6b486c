"""iter_proj / refine_matches: HIP vs the operation-for-operation numpy oracle, bit-exact."""
import numpy as np
import pytest
import torch

from oracle import matching_oracle as mo


def _ray_image(b, h, w, seed):
    """A smooth pinhole-like ray field + Scharr-free finite-difference gradients, like prep_for_iter_proj
    (VSLAM/utils_matching.py:109-133) produces; exact gradient operator does not matter for parity."""
    r = np.random.default_rng(seed)
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    out = np.zeros((b, h, w, 9), np.float32)
    for i in range(b):
        f = 0.9 * w + 5 * i
        d = np.stack([(xs - w / 2) / f, (ys - h / 2) / f, np.ones_like(xs, float)], -1)
        d += 0.01 * r.standard_normal(d.shape)
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        gx = np.zeros_like(d); gy = np.zeros_like(d)
        gx[:, 1:-1] = (d[:, 2:] - d[:, :-2]) / 2
        gy[1:-1] = (d[2:] - d[:-2]) / 2
        out[i] = np.concatenate([d, gx, gy], -1)
    return out


def _targets(rays, seed):
    b, h, w, _ = rays.shape
    r = np.random.default_rng(seed)
    n = h * w
    # targets = rays at randomly displaced pixels (what a true match looks like), some far off
    u = np.clip(np.tile(np.arange(w), h) + r.integers(-6, 7, n), 0, w - 1)
    v = np.clip(np.repeat(np.arange(h), w) + r.integers(-6, 7, n), 0, h - 1)
    pts = np.stack([rays[i, v, u, :3] for i in range(b)])
    pts += 0.002 * r.standard_normal(pts.shape).astype(np.float32)
    pts /= np.linalg.norm(pts, axis=-1, keepdims=True)
    p_init = np.stack([np.stack([np.tile(np.arange(w), h), np.repeat(np.arange(h), w)], -1)] * b).astype(np.float32)
    p_init += r.uniform(-1.5, 1.5, p_init.shape).astype(np.float32)  # also exercises the clamp at the border
    return pts.astype(np.float32), p_init


def _full_size_inputs(h=384, w=512, seed=2024):
    """The frontend's REAL size (hw = 196 608, VSLAM/utils_matching.py:152-179), built from operations that are bit-reproducible on any
    IEEE machine: PCG64 INTEGER draws scaled by powers of two, single fp32 elementwise operations, explicit left-to-right sums (no
    `standard_normal`, whose tails go through libm, and no `np.sum` / `linalg.norm`, whose order depends on the SIMD width).  The golden
    file holds only the reference's OUTPUTS for these inputs plus a SHA-256 of the inputs (tests/golden/make_golden_ref_full.py)."""
    rng = np.random.default_rng(seed)
    f32 = np.float32

    def noise(shape, pow2):     # uniform integers in [-2^15, 2^15) x 2^pow2: exact in fp32
        return rng.integers(-32768, 32768, shape).astype(f32) * f32(2.0 ** pow2)

    def unit(v):                # v / sqrt(v0^2 + v1^2 + ...), one IEEE operation at a time, left to right
        acc = v[..., 0] * v[..., 0]
        for k in range(1, v.shape[-1]):
            acc = acc + v[..., k] * v[..., k]
        return v / np.sqrt(acc)[..., None]

    ys, xs = np.meshgrid(np.arange(h, dtype=f32), np.arange(w, dtype=f32), indexing="ij")
    d = np.stack([(xs - f32(w / 2)) / f32(460.0), (ys - f32(h / 2)) / f32(460.0), np.ones_like(xs)], -1)
    d = unit(d + noise(d.shape, -22))                                   # +- 0.0078 of direction noise
    gx, gy = np.zeros_like(d), np.zeros_like(d)
    gx[:, 1:-1] = (d[:, 2:] - d[:, :-2]) * f32(0.5)
    gy[1:-1] = (d[2:] - d[:-2]) * f32(0.5)
    rays = np.ascontiguousarray(np.concatenate([d, gx, gy], -1)[None])   # [1,h,w,9]
    n = h * w
    gu, gv = np.tile(np.arange(w), h), np.repeat(np.arange(h), w)
    u = np.clip(gu + rng.integers(-6, 7, n), 0, w - 1)
    v = np.clip(gv + rng.integers(-6, 7, n), 0, h - 1)
    pts = np.ascontiguousarray(unit(d[v, u] + noise((n, 3), -24))[None])  # the rays of displaced pixels: what a true match looks like
    p_init = np.ascontiguousarray((np.stack([gu, gv], -1).astype(f32) + noise((n, 2), -14))[None])   # +- 2 px, exercises the border clamp
    D11 = unit(noise((h, w, 24), -15))
    su = np.clip(gu + rng.integers(-5, 6, n), 0, w - 1)
    sv = np.clip(gv + rng.integers(-5, 6, n), 0, h - 1)
    D21 = D11[sv, su] + noise((n, 24), -20)                              # the neighbour's descriptor + ~0.03 of noise: near-ties matter
    p1 = np.ascontiguousarray(np.stack([gu, gv], -1).astype(np.int64)[None])
    return dict(rays=rays, pts=pts, p_init=p_init, D11=np.ascontiguousarray(D11.astype(np.float16)[None]),
                D21=np.ascontiguousarray(D21.astype(np.float16)[None]), p1=p1)


def _inputs_digest(inp):
    import hashlib
    hsh = hashlib.sha256()
    for k in sorted(inp):
        hsh.update(k.encode()); hsh.update(str(inp[k].dtype).encode()); hsh.update(str(inp[k].shape).encode()); hsh.update(inp[k].tobytes())
    return hsh.hexdigest()


def test_oracle_iter_proj_converges_on_identity():
    rays = _ray_image(1, 24, 32, 0)
    pts = rays[:, :, :, :3].reshape(1, -1, 3).copy()
    p0 = np.stack([np.tile(np.arange(32), 24), np.repeat(np.arange(24), 32)], -1)[None].astype(np.float32)
    p, c = mo.iter_proj_oracle(rays, pts, p0, 10, 1e-8, 1e-6)
    inner = (p0[0, :, 0] >= 1) & (p0[0, :, 0] <= 30) & (p0[0, :, 1] >= 1) & (p0[0, :, 1] <= 22)
    assert np.abs(p[0][inner] - p0[0][inner]).max() < 0.05 and c[0][inner].mean() > 0.99


def test_oracle_refine_prefers_matching_descriptor():
    r = np.random.default_rng(0)
    D11 = r.standard_normal((1, 20, 20, 24)).astype(np.float16)
    D11 /= np.linalg.norm(D11.astype(np.float32), axis=-1, keepdims=True).astype(np.float16)
    tgt = np.array([[[7, 9]]], np.int64)
    D21 = D11[:, 9, 7][:, None].copy()
    # dense window (dilation 1): the query's own pixel (dot = 1) wins over random neighbours
    out = mo.refine_matches_oracle(D11, D21, np.array([[[5, 8]]], np.int64), 4, 1)
    assert (out == tgt).all()
    # nothing scores above the positive-min threshold -> match stays put (:47 quirk)
    out2 = mo.refine_matches_oracle(D11, -D21 * 0, np.array([[[5, 8]]], np.int64), 4, 5)
    assert (out2 == np.array([[[5, 8]]])).all()


@pytest.mark.gpu
@pytest.mark.parametrize("b,h,w,iters", [(1, 48, 64, 10), (2, 37, 53, 10), (1, 24, 32, 0), (1, 96, 128, 3)])
def test_iter_proj_bit_exact(b, h, w, iters, dev):
    import mast3r_slam_backends as msb
    rays = _ray_image(b, h, w, 1)
    pts, p0 = _targets(rays, 2)
    po, co = mo.iter_proj_oracle(rays, pts, p0, iters, 1e-8, 1e-6)
    ph, ch = msb.iter_proj(torch.from_numpy(rays).to(dev), torch.from_numpy(pts).to(dev), torch.from_numpy(p0).to(dev),
                           iters, 1e-8, 1e-6)
    assert ph.dtype == torch.float32 and ch.dtype == torch.bool
    assert np.array_equal(ph.cpu().numpy(), po)
    assert np.array_equal(ch.cpu().numpy(), co)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float16, np.float32])
@pytest.mark.parametrize("b,h,w,fdim,radius,dil", [(1, 48, 64, 24, 4, 5), (2, 30, 41, 24, 3, 2), (1, 33, 20, 16, 2, 3), (1, 16, 16, 24, 0, 1)])
def test_refine_matches_bit_exact(dtype, b, h, w, fdim, radius, dil, dev):
    import mast3r_slam_backends as msb
    r = np.random.default_rng(3)
    D11 = r.standard_normal((b, h, w, fdim)).astype(np.float32)
    D11 /= np.linalg.norm(D11, axis=-1, keepdims=True)
    # queries = descriptors of nearby pixels + noise (ties and near-ties are what make rounding matter)
    n = h * w
    u = np.clip(np.tile(np.arange(w), h) + r.integers(-5, 6, n), 0, w - 1)
    v = np.clip(np.repeat(np.arange(h), w) + r.integers(-5, 6, n), 0, h - 1)
    D21 = np.stack([D11[i, v, u] for i in range(b)]) + 0.05 * r.standard_normal((b, n, fdim)).astype(np.float32)
    p1 = np.stack([np.stack([np.tile(np.arange(w), h), np.repeat(np.arange(h), w)], -1)] * b).astype(np.int64)
    D11, D21 = D11.astype(dtype), D21.astype(dtype)
    po = mo.refine_matches_oracle(D11, D21, p1, radius, dil)
    (ph,) = msb.refine_matches(torch.from_numpy(D11).to(dev), torch.from_numpy(D21).to(dev), torch.from_numpy(p1).to(dev), radius, dil)
    assert ph.dtype == torch.int64
    assert np.array_equal(ph.cpu().numpy(), po)


@pytest.mark.gpu
def test_matching_full_size_properties(dev):
    """512x384 (BASELINE config 1/2 frontend size): iter_proj output stays inside the clamp box and
    is a fixed point of a second call with 0 iterations; refine never leaves the image."""
    import mast3r_slam_backends as msb
    h, w = 384, 512
    rays = torch.from_numpy(_ray_image(1, h, w, 5)).to(dev)
    pts, p0 = _targets(rays.cpu().numpy(), 6)
    p, c = msb.iter_proj(rays, torch.from_numpy(pts).to(dev), torch.from_numpy(p0).to(dev), 10, 1e-8, 1e-6)
    assert float(p[..., 0].min()) >= 1 and float(p[..., 0].max()) <= w - 2
    assert float(p[..., 1].min()) >= 1 and float(p[..., 1].max()) <= h - 2
    p2, _ = msb.iter_proj(rays, torch.from_numpy(pts).to(dev), p, 0, 1e-8, 1e-6)
    assert torch.equal(p2, p)
    assert c.dtype == torch.bool and c.shape == (1, h * w)
    D11 = torch.randn(1, h, w, 24, device=dev).half()
    D21 = torch.randn(1, h * w, 24, device=dev).half()
    (q,) = msb.refine_matches(D11, D21, p.long(), 4, 5)
    assert int(q[..., 0].min()) >= 0 and int(q[..., 0].max()) < w and int(q[..., 1].min()) >= 0 and int(q[..., 1].max()) < h
    assert int((q - p.long()).abs().max()) <= 4 * (5 + 4 + 3 + 2 + 1)
