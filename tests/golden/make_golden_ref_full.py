#!/usr/bin/env python
"""FULL-SIZE golden vectors from the REFERENCE's own kernels (oracle/_ref = matching_kernels.cu / simple_knn.cu compiled for the host by
oracle/ref_shim/build_ref.py), at the sizes the frontend and the scene model really call them with:

  iter_proj       1 x 384 x 512 (hw = 196 608), 10 iterations, lambda 1e-8, cost threshold 1e-6   VSLAM/utils_matching.py:152-159
  refine_matches  same size, radius 4, dilation_max 5, fp16 descriptors (scalar_t = half)           VSLAM/utils_matching.py:171-179
  distIndex2      10^6 points, K = 3: the reference's rows for 20 000 of the queries                simple_knn.cu:468-522

Run in the build container (needs /root/reference):   python tests/golden/make_golden_ref_full.py
Writes tests/golden/ref_full.npz = the reference's OUTPUTS + a SHA-256 of the inputs; the inputs themselves are regenerated on the
GPU box by tests/test_matching.py:_full_size_inputs / tests/test_knn.py:_clouds (bit-reproducible constructions) and checked against
the digest before anything is compared.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_native as rn  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
KNN_ROWS = 20000


def main():
    assert rn.available(), "oracle/_ref not built: python oracle/ref_shim/build_ref.py"
    from test_knn import _clouds
    from test_matching import _full_size_inputs, _inputs_digest
    out = {}
    inp = _full_size_inputs()
    out["matching_inputs_sha256"] = np.array(_inputs_digest(inp))
    rm = rn.ref_matching()
    t = lambda a: torch.from_numpy(a)
    t0 = time.time()
    p, c = rm.iter_proj(t(inp["rays"]), t(inp["pts"]), t(inp["p_init"]), 10, 1e-8, 1e-6)
    print(f"iter_proj {time.time() - t0:.1f} s, converged {float(c.float().mean()):.3f}")
    out["ip_p"], out["ip_conv"] = p.numpy(), np.packbits(c.numpy())
    t0 = time.time()
    (q,) = rm.refine_matches(t(inp["D11"]), t(inp["D21"]), t(inp["p1"]), 4, 5)
    q = q.numpy()
    print(f"refine_matches {time.time() - t0:.1f} s, moved {float((q != inp['p1']).any(-1).mean()):.3f}")
    assert q.min() >= 0 and q.max() < 32768
    out["rf_out"] = q.astype(np.int16)
    pts = _clouds("uniform", 1_000_000, 11)
    t0 = time.time()
    d, i = rn.knn_index2(pts, 3)
    print(f"knn_index2 10^6: {time.time() - t0:.1f} s")
    rows = np.sort(np.random.default_rng(5).choice(1_000_000, KNN_ROWS, replace=False))
    import hashlib
    out["knn_points_sha256"] = np.array(hashlib.sha256(pts.tobytes()).hexdigest())
    out["knn_rows"], out["knn_dists"], out["knn_idx"] = rows.astype(np.int32), d[rows], i[rows]
    path = os.path.join(HERE, "ref_full.npz")
    np.savez_compressed(path, **out)
    print("ref_full.npz", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
