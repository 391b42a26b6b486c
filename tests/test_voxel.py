"""SceneModel.update_voxel (densification path): the numpy oracle against goldens produced by the REFERENCE's
own method source executed on CPU (tests/golden/make_golden_voxel.py).  The device path for this function is not built yet;
this pins the restatement it will be checked against."""
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
