"""adamUpdate / adamUpdateBasic: HIP vs the IEEE-fp32 oracle, bit-exact (both sides unfused fp32)."""
import numpy as np
import pytest
import torch

from oracle import adam_oracle


def _mk(N, M, seed, vis_frac):
    r = np.random.default_rng(seed)
    p = r.standard_normal((N, M)).astype(np.float32)
    g = (r.standard_normal((N, M)) * 1e-2).astype(np.float32)
    m = (r.standard_normal((N, M)) * 1e-3).astype(np.float32)
    v = (r.random((N, M)) * 1e-5).astype(np.float32)
    vis = r.random(N) < vis_frac
    return p, g, m, v, vis


def test_oracle_basic_properties():
    p, g, m, v, vis = _mk(50, 3, 0, 0.5)
    pn, mn, vn = adam_oracle.adam_update_oracle(p, g, m, v, vis, np.float32(1e-3), 0.5, 0.99, 1e-15, 50, 3)
    # invisible rows untouched, visible rows changed, moments follow the EMA definition
    assert np.array_equal(pn[~vis], p[~vis]) and np.array_equal(mn[~vis], m[~vis]) and np.array_equal(vn[~vis], v[~vis])
    assert not np.array_equal(pn[vis], p[vis])
    assert np.allclose(mn[vis], 0.5 * m[vis] + 0.5 * g[vis], rtol=1e-6)
    # zero gradient, zero moments -> no movement (eps keeps 0/0 away)
    z = np.zeros((4, 2), np.float32)
    pz, _, _ = adam_oracle.adam_update_oracle(np.ones((4, 2), np.float32), z, z, z, np.ones(4, bool), np.float32(1.0), 0.9, 0.999, 1e-15, 4, 2)
    assert np.array_equal(pz, np.ones((4, 2), np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("N,M", [(1000, 3), (777, 45), (4096, 1), (513, 16), (1, 1), (100000, 4), (33, 7)])
@pytest.mark.parametrize("lr_kind", ["scalar", "row", "elem"])
@pytest.mark.parametrize("vis_frac", [0.0, 0.3, 1.0])
def test_adam_update_bit_exact(N, M, lr_kind, vis_frac, dev):
    from diff_gaussian_rasterization import adamUpdate
    p, g, m, v, vis = _mk(N, M, N * 31 + M, vis_frac)
    r = np.random.default_rng(5)
    if lr_kind == "scalar":
        lr = np.float32(1.6e-4)
        lr_t = torch.tensor(float(lr), dtype=torch.float32, device=dev)
    elif lr_kind == "row":
        lr = (r.random(N) * 1e-3).astype(np.float32)
        lr_t = torch.from_numpy(lr).to(dev)
    else:
        lr = (r.random((N, M)) * 1e-3).astype(np.float32)
        lr_t = torch.from_numpy(lr).to(dev)
    po, mo, vo = adam_oracle.adam_update_oracle(p, g, m, v, vis, lr, 0.5, 0.99, 1e-15, N, M)
    pt, gt, mt, vt = (torch.from_numpy(a.copy()).to(dev) for a in (p, g, m, v))
    adamUpdate(pt, gt, mt, vt, torch.from_numpy(vis).to(dev), lr_t, 0.5, 0.99, 1e-15, N, M)
    assert np.array_equal(pt.cpu().numpy(), po)
    assert np.array_equal(mt.cpu().numpy(), mo)
    assert np.array_equal(vt.cpu().numpy(), vo)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(32, 32), (32,), (7, 32), (7,), (3, 2), (3,), (3, 4), (1025,)])
def test_adam_update_basic_bit_exact(shape, dev):
    """The tensors adamUpdateBasic sees in ARTDECO: mlp_cov weights, pose 6D/t, exposure."""
    from diff_gaussian_rasterization import adamUpdateBasic
    n = int(np.prod(shape))
    p, g, m, v, _ = _mk(n, 1, n, 1.0)
    po, mo, vo = adam_oracle.adam_update_basic_oracle(p, g, m, v, 2e-3, 0.8, 0.99, 1e-15)
    pt, gt, mt, vt = (torch.from_numpy(a.copy().reshape(shape)).to(dev) for a in (p, g, m, v))
    adamUpdateBasic(pt, gt, mt, vt, 2e-3, 0.8, 0.99, 1e-15)
    assert np.array_equal(pt.cpu().numpy().reshape(-1, 1), po)
    assert np.array_equal(mt.cpu().numpy().reshape(-1, 1), mo)
    assert np.array_equal(vt.cpu().numpy().reshape(-1, 1), vo)


@pytest.mark.gpu
def test_adam_unaligned_view_uses_scalar_path(dev):
    from diff_gaussian_rasterization import adamUpdateBasic
    n = 1001
    p, g, m, v, _ = _mk(n + 1, 1, 3, 1.0)
    po, mo, vo = adam_oracle.adam_update_basic_oracle(p[1:], g[1:], m[1:], v[1:], 1e-3, 0.9, 0.999, 1e-15)
    pt, gt, mt, vt = (torch.from_numpy(a.copy().reshape(-1)).to(dev) for a in (p, g, m, v))
    adamUpdateBasic(pt[1:], gt[1:], mt[1:], vt[1:], 1e-3, 0.9, 0.999, 1e-15)
    assert np.array_equal(pt[1:].cpu().numpy(), po.reshape(-1))
    assert pt[0].item() == p[0, 0]


@pytest.mark.gpu
def test_sparse_adam_through_reference_style_optimizer(dev):
    """Drive the kernels exactly as Reconstruct/scene/optimizers.py:106-161 does (0-dim lr + per-element lr)."""
    from diff_gaussian_rasterization import adamUpdate
    N = 5000
    r = np.random.default_rng(0)
    xyz = torch.from_numpy(r.standard_normal((N, 3)).astype(np.float32)).to(dev)
    xyz.grad = torch.from_numpy(r.standard_normal((N, 3)).astype(np.float32)).to(dev)
    m, v = torch.zeros_like(xyz), torch.zeros_like(xyz)
    lr = torch.ones_like(xyz) * 1e-4
    vis = torch.from_numpy(r.random(N) < 0.4).to(dev)
    before = xyz.clone()
    adamUpdate(xyz, xyz.grad, m, v, vis, lr, 0.5, 0.99, 1e-15, N, 3)
    assert torch.equal(xyz[~vis], before[~vis])
    # first step with zero moments: |step| = lr * (1-b1)|g| / (sqrt((1-b2) g^2) + eps) = lr * 0.5 / 0.1 = 5 lr
    step = (xyz - before)[vis]
    assert torch.allclose(step.abs(), torch.full_like(step, 5e-4), rtol=1e-3)
