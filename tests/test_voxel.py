"""SceneModel.update_voxel (densification path): the numpy oracle against goldens produced by the REFERENCE's
own method source executed on CPU (tests/golden/make_golden_voxel.py), and the DEVICE path (artdeco_amd/csrc/voxel.hip through
artdeco_amd.fused.update_voxel_device) against the same goldens and, at 1 M + 50 k points, against the oracle -- bit for bit."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import voxel_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden_voxel", os.path.join(HERE, "golden", "make_golden_voxel.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)
CASES = gen.cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference_goldens(name):
    new, xyz, cls, vs = CASES[name]
    d = np.load(os.path.join(HERE, "golden", name + ".npz"))
    assert float(d["in_sum"]) == float(new.astype(np.float64).sum() + xyz.astype(np.float64).sum())
    res = voxel_oracle.update_voxel(new, xyz, cls, vs)
    assert len(res) == sum(1 for k in d.files if k.startswith("out"))
    for i, r in enumerate(res):
        assert np.array_equal(np.asarray(r), d[f"out{i}"]), (name, i)


def test_majority_vote_ties_take_the_smallest_class():
    xyz = np.array([[0.01, 0.01, 0.01], [0.02, 0.02, 0.02], [0.03, 0.01, 0.02], [0.04, 0.03, 0.01]], np.float32)  # one voxel
    cls = np.array([[7], [3], [7], [3]], np.int64)
    new = np.array([[0.05, 0.05, 0.05], [5.0, 5.0, 5.0]], np.float32)
    orig, upd, count = voxel_oracle.update_voxel(new, xyz, cls, 0.1)
    assert (orig == 3).all() and upd[0, 0] == 3 and upd[1, 0] == 8 and count == 1


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_device_update_voxel_matches_reference_goldens(name, dev):
    import torch
    from artdeco_amd import fused
    new, xyz, cls, vs = CASES[name]
    d = np.load(os.path.join(HERE, "golden", name + ".npz"))
    res = fused.update_voxel_device(torch.from_numpy(new).to(dev), torch.from_numpy(xyz).to(dev), torch.from_numpy(cls).to(dev), vs,
                                    reciprocal=False)   # the goldens are the reference's CPU execution: true division
    assert len(res) == sum(1 for k in d.files if k.startswith("out"))
    for i, r in enumerate(res):
        got = r.cpu().numpy() if hasattr(r, "cpu") else np.int64(r)
        assert np.array_equal(got, d[f"out{i}"]), (name, i)


@pytest.mark.gpu
@pytest.mark.parametrize("recip", [False, True])
@pytest.mark.parametrize("N,M,vs,wide", [(1_000_000, 50_000, 0.1, False), (300_000, 20_000, 0.004, True), (5_000, 0, 0.2, False)])
def test_device_update_voxel_matches_oracle_at_map_size(N, M, vs, wide, recip, dev):
    """Map-sized inputs (1 M Gaussians + 50 k new points per LoD level) and a grid fine enough for hashes above 2^32 (two-word
    radix sort), against the golden-pinned oracle."""
    import torch
    from artdeco_amd import fused
    rng = np.random.default_rng(N + M)
    span = 40.0 if wide else 6.0
    centres = rng.uniform(-span, span, (400, 3))
    pts = lambda k: (centres[rng.integers(0, 400, k)] + 0.4 * rng.standard_normal((k, 3))).astype(np.float32)
    xyz, new = pts(N), pts(M) + np.float32(0.05)
    cls = rng.integers(0, max(N // 10, 1), (N, 1)).astype(np.int64)
    ro, rn, rc = voxel_oracle.update_voxel(new, xyz, cls, vs, reciprocal=recip)
    go, gn, gc = fused.update_voxel_device(torch.from_numpy(new).to(dev), torch.from_numpy(xyz).to(dev), torch.from_numpy(cls).to(dev), vs,
                                           reciprocal=recip)
    assert gc == rc
    assert np.array_equal(go.cpu().numpy(), ro) and np.array_equal(gn.cpu().numpy(), rn)


@pytest.mark.gpu
def test_device_update_voxel_equals_the_torch_chain_on_the_gpu(dev):
    """The reference's own sequence of torch operations (h3dgsv3.py:227-316, restated in tools/bench_update_voxel.py) executed
    by torch ON THE GPU -- division by the voxel size as torch's GPU kernel rounds it -- gives the device path's results."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    from bench_update_voxel import torch_chain
    from artdeco_amd import fused
    new, xyz, cls, vs = CASES["voxel_mixed"]
    rng = np.random.default_rng(5)
    xyz = np.concatenate([xyz, (rng.uniform(-3, 3, (200_000, 3))).astype(np.float32)])
    cls = np.concatenate([cls, rng.integers(0, 5000, (200_000, 1)).astype(np.int64)])
    t = lambda a: torch.from_numpy(a).to(dev)
    a = torch_chain(t(new), t(xyz), t(cls), vs)
    b = fused.update_voxel_device(t(new), t(xyz), t(cls), vs)
    assert a[2] == b[2] and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.gpu
@pytest.mark.parametrize("N,vs,wide", [(1_000_000, 0.1, False), (300_000, 0.004, True)])
def test_device_update_voxel_table_reuse_across_lod_levels(N, vs, wide, dev):
    """add_new_gaussians calls update_voxel once per LoD level with the same map (h3dgsv3.py:884-887).  With a `table` the first call
    leaves the map's voxel table behind and later calls only look the new points up (adk_voxel_assign_new) -- when the class ids are
    the table-building call's own output and the grid is the same (relabelling is idempotent on a fixed grid).  Every call must
    return what a call without the table returns; a batch that moves the grid origin, or another class tensor, must take the full
    path by itself."""
    import torch
    from artdeco_amd import fused
    import time
    rng = np.random.default_rng(N)
    span = 40.0 if wide else 6.0
    centres = rng.uniform(-span, span, (400, 3))
    pts = lambda k: (centres[rng.integers(0, 400, k)] + 0.4 * rng.standard_normal((k, 3))).astype(np.float32)
    xyz = torch.from_numpy(pts(N)).to(dev)
    cls0 = torch.from_numpy(rng.integers(0, max(N // 10, 1), (N, 1)).astype(np.int64)).to(dev)
    lo, hi = xyz.min(0).values, xyz.max(0).values
    inside = lambda k: (lo + (hi - lo) * torch.from_numpy(rng.uniform(0.05, 0.95, (k, 3)).astype(np.float32)).to(dev)).contiguous()
    # level 0 relabels the map (random labels -> majority per voxel) and builds the table; from level 1 on the labels are its output
    cls, table, reused, t_full, t_reuse = cls0, {}, [], [], []
    batches = [inside(50_000), inside(20_000), inside(120_000), inside(5_000), torch.cat([inside(1_000), (lo - 1.0)[None]]), inside(3_000), inside(2_000),
               inside(0)]
    box = lambda new: (tuple(torch.minimum(lo, new.min(0).values).tolist()), tuple(torch.maximum(hi, new.max(0).values).tolist())) if len(new) else (tuple(lo.tolist()), tuple(hi.tolist()))
    expected, table_box = [], None
    for level, new in enumerate(batches):
        ref_o, ref_n, ref_c = fused.update_voxel_device(new, xyz, cls, vs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        got_o, got_n, got_c = fused.update_voxel_device(new, xyz, cls, vs, table=table)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        assert got_c == ref_c and torch.equal(got_o, ref_o) and torch.equal(got_n, ref_n), level
        expected.append(box(new) == table_box)     # the labels are the table-building call's output by construction: same grid = reuse
        if not expected[-1]:
            table_box = box(new)
        reused.append(table["reused"])
        (t_reuse if table["reused"] else t_full).append(dt)
        cls = got_o
    # level 0 builds (and replaces the random labels by majorities), 1-3 reuse (incl. a batch larger than the first: the workspace
    # grows), 4 rebuilds (a point outside the map's box moves the grid origin), 5 rebuilds (the origin is back), 6-7 reuse
    assert reused == expected == [False, True, True, True, False, False, True, True], (reused, expected)
    print(f"update_voxel at {N}: full {np.mean(t_full):.3f} ms, table reused {np.mean(t_reuse):.3f} ms")
    # a DIFFERENT class tensor (even with equal values) never hits the table
    fused.update_voxel_device(batches[1], xyz, cls.clone(), vs, table=table)
    assert table["reused"] is False
