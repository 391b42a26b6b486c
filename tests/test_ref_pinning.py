"""a5 / a9 / a10 pinned to the REFERENCE'S OWN SOURCE.

oracle/_ref holds simple_knn.cu and matching_kernels.cu compiled for the host straight from /root/reference
(oracle/ref_shim/build_ref.py: a CUDA-execution-model shim, the kernels' text untouched apart from the `<<<>>>` launch
syntax).  tests/golden/ref_knn.npz / ref_matching.npz are its outputs (tests/golden/make_golden_ref.py).

  CPU: numpy oracles == golden (bit for bit) and, where oracle/_ref is present, == the compiled reference on fresh inputs.
  GPU: HIP kernels (through the drop-in modules and the C ABI) == golden.
KNN rows are compared as (distance, index) sets sorted by (distance, index): the reference's neighbour order within a
row is unspecified (replace-the-max policy, simple_knn.cu:405-420).
"""
import os

import numpy as np
import pytest
import torch

from oracle import knn_oracle as ko
from oracle import matching_oracle as mo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _knn():
    return np.load(os.path.join(GOLD, "ref_knn.npz"))


def _mat():
    return np.load(os.path.join(GOLD, "ref_matching.npz"))


def _canon(d, i):
    """rows sorted by (distance, index)."""
    order = np.lexsort((i, d), axis=1)
    return np.take_along_axis(d, order, 1), np.take_along_axis(i, order, 1)


def _same_rows(d, i, gd, gi):
    """Distances bit-equal; indices equal except inside groups of exactly tied distances at the K-th place (any of the
    tied candidates is a correct K-th neighbour)."""
    d, i = _canon(np.asarray(d), np.asarray(i))
    gd, gi = _canon(gd, gi)
    if not np.array_equal(d, gd):
        return False
    diff = (i != gi)
    if not diff.any():
        return True
    rows = np.nonzero(diff.any(1))[0]
    for r in rows:  # a differing index must belong to a distance tie that reaches the row's last slot
        for c in np.nonzero(diff[r])[0]:
            if d[r, c] != d[r, -1]:
                return False
    return True


# ------------------------------------------------------------------------------------------- CPU: oracle vs reference
@pytest.mark.parametrize("name", ["uni", "clu", "clu8", "tiny"])
def test_knn_oracle_equals_reference_golden(name):
    g = _knn()
    d, i = ko.dist_index2_oracle(g[f"{name}_points"], int(g[f"{name}_K"]))
    assert _same_rows(d, i, g[f"{name}_dists"], g[f"{name}_idx"])
    if name == "tiny":  # unfilled slots: (FLT_MAX, -1), simple_knn.cu:440-441 + spatial.cu:36
        assert (g["tiny_idx"] == -1).sum() == 5 * (8 - 4) and (g["tiny_dists"][g["tiny_idx"] == -1] == np.finfo(np.float32).max).all()


def test_knn_mean_and_query_oracle_equal_reference_golden():
    g = _knn()
    assert np.array_equal(ko.dist_cuda2_oracle(g["uni_points"]), g["uni_mean"])
    assert np.array_equal(ko.dist_cuda2_oracle(g["clu_points"]), g["clu_mean"])
    d, i = ko.knn_oracle(g["uni_points"], g["q_q"], g["q_n"], int(g["q_K"]))
    assert _same_rows(d, i, g["q_dists"], g["q_idx"])


def test_matching_oracle_equals_reference_golden():
    g = _mat()
    for tag in ("ip10", "ip3"):
        it, lam, thr = g[f"{tag}_args"]
        p, c = mo.iter_proj_oracle(g["ip_rays"], g["ip_pts"], g["ip_pinit"], int(it), float(lam), float(thr))
        assert np.array_equal(p, g[f"{tag}_p"]) and np.array_equal(c, g[f"{tag}_conv"]), tag
    D11, D21, p1 = g["rf_D11"], g["rf_D21"], g["rf_p1"]
    assert np.array_equal(mo.refine_matches_oracle(D11, D21, p1, 4, 5), g["rf_out_r4d5"])
    assert np.array_equal(mo.refine_matches_oracle(D11, D21, p1, 2, 2), g["rf_out_r2d2"])
    assert np.array_equal(mo.refine_matches_oracle(D11.astype(np.float32), D21.astype(np.float32), p1, 3, 2), g["rf32_out_r3d2"])
    assert (g["rf_out_r4d5"] != p1).any()  # the search actually moved matches


def test_oracles_equal_compiled_reference_on_fresh_inputs():
    """Live: the reference compiled here from /root/reference (skipped on a box that has neither it nor oracle/_ref)."""
    from oracle import ref_native as rn
    if not rn.available():
        pytest.skip("oracle/_ref not present and /root/reference absent")
    r = np.random.default_rng(123)
    pts = (r.standard_normal((3000, 3)) * np.array([1.0, 0.2, 3.0])).astype(np.float32)
    d, i = rn.knn_index2(pts, 5)
    od, oi = ko.dist_index2_oracle(pts, 5)
    assert _same_rows(od, oi, d, i)
    assert np.array_equal(rn.knn_mean(pts), ko.dist_cuda2_oracle(pts))
    from test_matching import _ray_image, _targets
    rays = _ray_image(1, 32, 48, 9)
    tg, p0 = _targets(rays, 10)
    p, c = rn.ref_matching().iter_proj(torch.from_numpy(rays), torch.from_numpy(tg), torch.from_numpy(p0), 7, 1e-6, 1e-5)
    po, co = mo.iter_proj_oracle(rays, tg, p0, 7, 1e-6, 1e-5)
    assert np.array_equal(p.numpy(), po) and np.array_equal(c.numpy(), co)
    D11 = r.standard_normal((1, 32, 48, 24)).astype(np.float16)
    D21 = r.standard_normal((1, 32 * 48, 24)).astype(np.float16)
    p1 = np.stack([np.tile(np.arange(48), 32), np.repeat(np.arange(32), 48)], -1)[None].astype(np.int64)
    (o,) = rn.ref_matching().refine_matches(torch.from_numpy(D11), torch.from_numpy(D21), torch.from_numpy(p1), 3, 3)
    assert np.array_equal(o.numpy(), mo.refine_matches_oracle(D11, D21, p1, 3, 3))


# ------------------------------------------------------------------------------------------- GPU: HIP vs reference
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["uni", "clu", "clu8", "tiny"])
def test_hip_knn_equals_reference_golden(name, dev):
    from simple_knn._C import distIndex2
    g = _knn()
    pts, K = g[f"{name}_points"], int(g[f"{name}_K"])
    d, i = distIndex2(torch.from_numpy(pts).to(dev), K)
    assert _same_rows(d.cpu().numpy().reshape(-1, K), i.cpu().numpy().reshape(-1, K), g[f"{name}_dists"], g[f"{name}_idx"])


@pytest.mark.gpu
def test_hip_knn_mean_and_query_equal_reference_golden(dev):
    from simple_knn._C import distCUDA2, distIndexQ
    g = _knn()
    for n in ("uni", "clu"):
        assert np.array_equal(distCUDA2(torch.from_numpy(g[f"{n}_points"]).to(dev)).cpu().numpy(), g[f"{n}_mean"])
    K = int(g["q_K"])
    d, i = distIndexQ(torch.from_numpy(g["uni_points"]).to(dev), torch.from_numpy(g["q_q"]).to(dev),
                      torch.from_numpy(g["q_n"]).to(dev), K)
    assert _same_rows(d.cpu().numpy().reshape(-1, K), i.cpu().numpy().reshape(-1, K), g["q_dists"], g["q_idx"])


@pytest.mark.gpu
def test_hip_matching_equals_reference_golden(dev):
    import mast3r_slam_backends as be
    g = _mat()
    t = lambda a: torch.from_numpy(a).to(dev)
    for tag in ("ip10", "ip3"):
        it, lam, thr = g[f"{tag}_args"]
        p, c = be.iter_proj(t(g["ip_rays"]), t(g["ip_pts"]), t(g["ip_pinit"]), int(it), float(lam), float(thr))
        assert np.array_equal(p.cpu().numpy(), g[f"{tag}_p"]) and np.array_equal(c.cpu().numpy(), g[f"{tag}_conv"]), tag
    D11, D21, p1 = t(g["rf_D11"]), t(g["rf_D21"]), t(g["rf_p1"])
    assert np.array_equal(be.refine_matches(D11, D21, p1, 4, 5)[0].cpu().numpy(), g["rf_out_r4d5"])
    assert np.array_equal(be.refine_matches(D11, D21, p1, 2, 2)[0].cpu().numpy(), g["rf_out_r2d2"])
    assert np.array_equal(be.refine_matches(D11.float(), D21.float(), p1, 3, 2)[0].cpu().numpy(), g["rf32_out_r3d2"])


# ------------------------------------------------------------------------------------------- FULL SIZE (round 5)
def _full():
    return np.load(os.path.join(GOLD, "ref_full.npz"))


def test_full_size_inputs_are_bit_reproducible_here():
    """tests/golden/ref_full.npz holds the reference's OUTPUTS at 1 x 384 x 512 and a SHA-256 of the inputs it was run on; the inputs are
    rebuilt wherever the tests run (integer draws, single fp32 operations: tests/test_matching.py:_full_size_inputs).  If this fails on a
    host, the full-size comparisons on that host would compare different problems -- they check the digest themselves too."""
    import hashlib
    from test_knn import _clouds
    from test_matching import _full_size_inputs, _inputs_digest
    g = _full()
    assert _inputs_digest(_full_size_inputs()) == str(g["matching_inputs_sha256"])
    assert hashlib.sha256(_clouds("uniform", 1_000_000, 11).tobytes()).hexdigest() == str(g["knn_points_sha256"])


def test_matching_oracle_equals_the_reference_at_full_size():
    """iter_proj (10 iterations) on all 196 608 pixels and refine_matches (radius 4, 5 dilation levels, fp16) on a 24 576-query slice: the
    numpy oracles against the reference's own kernels' outputs at the frontend's real size."""
    from test_matching import _full_size_inputs
    g, inp = _full(), _full_size_inputs()
    p, c = mo.iter_proj_oracle(inp["rays"], inp["pts"], inp["p_init"], 10, 1e-8, 1e-6)
    assert np.array_equal(p, g["ip_p"]) and np.array_equal(np.packbits(c), g["ip_conv"])
    assert 0.2 < c.mean() < 1.0
    sl = slice(70_000, 70_000 + 24_576)
    q = mo.refine_matches_oracle(inp["D11"], inp["D21"][:, sl], inp["p1"][:, sl], 4, 5)
    assert np.array_equal(q.astype(np.int16), g["rf_out"][:, sl])


@pytest.mark.gpu
def test_hip_matching_equals_the_reference_at_full_size(dev):
    """a9 / a10 at hw = 196 608 (VSLAM/utils_matching.py:152-179), bit for bit against matching_kernels.cu:119-316 / :25-116 compiled for the
    host: every pixel's refined position and convergence flag, every query's refined integer match (fp16 accumulation included)."""
    import mast3r_slam_backends as be
    from test_matching import _full_size_inputs, _inputs_digest
    g, inp = _full(), _full_size_inputs()
    assert _inputs_digest(inp) == str(g["matching_inputs_sha256"]), "the full-size inputs are not reproduced bit for bit on this host"
    t = lambda a: torch.from_numpy(a).to(dev)
    p, c = be.iter_proj(t(inp["rays"]), t(inp["pts"]), t(inp["p_init"]), 10, 1e-8, 1e-6)
    assert np.array_equal(p.cpu().numpy(), g["ip_p"])
    assert np.array_equal(np.packbits(c.cpu().numpy()), g["ip_conv"])
    (q,) = be.refine_matches(t(inp["D11"]), t(inp["D21"]), t(inp["p1"]), 4, 5)
    assert q.dtype == torch.int64 and np.array_equal(q.cpu().numpy().astype(np.int16), g["rf_out"])
    assert float((q.cpu() != torch.from_numpy(inp["p1"])).any(-1).float().mean()) > 0.5     # the search moved most matches


@pytest.mark.gpu
def test_hip_knn_equals_the_reference_at_a_million_points(dev):
    """a5 at 10^6 points, K = 3: 20 000 rows of the reference's own run (simple_knn.cu:468-522 compiled for the host, 18 minutes of CPU),
    distances bit-equal, neighbour sets equal up to exact distance ties at the K-th place."""
    import hashlib
    from simple_knn._C import distIndex2
    from test_knn import _clouds
    g = _full()
    pts = _clouds("uniform", 1_000_000, 11)
    assert hashlib.sha256(pts.tobytes()).hexdigest() == str(g["knn_points_sha256"])
    d, i = distIndex2(torch.from_numpy(pts).to(dev), 3)
    rows = g["knn_rows"].astype(np.int64)
    assert rows.size >= 10_000
    assert _same_rows(d.view(-1, 3)[rows].cpu().numpy(), i.view(-1, 3)[rows].cpu().numpy(), g["knn_dists"], g["knn_idx"])
