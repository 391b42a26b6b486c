#!/usr/bin/env python
"""Golden vectors from the REFERENCE's own kernels (oracle/_ref: simple_knn.cu and matching_kernels.cu compiled for the
host by oracle/ref_shim/build_ref.py).  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_ref.py

Writes tests/golden/ref_knn.npz and tests/golden/ref_matching.npz (inputs and the reference's outputs).  The GPU box
has no /root/reference; the HIP kernels and the numpy oracles are compared against these files there.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_native as rn  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def knn_cases():
    r = np.random.default_rng(7)
    uni = r.random((6000, 3)).astype(np.float32)
    # clustered surface samples with exact duplicates and a far outlier (ties, degenerate boxes)
    c = r.standard_normal((40, 3)).astype(np.float32)
    clu = (c[r.integers(0, 40, 5000)] + 0.02 * r.standard_normal((5000, 3))).astype(np.float32)
    clu[100:140] = clu[60:100]
    clu[-1] = (50.0, -20.0, 3.0)
    tiny = r.random((5, 3)).astype(np.float32)      # fewer points than K + 1
    out = {}
    for name, pts, K in (("uni", uni, 3), ("clu", clu, 3), ("clu8", clu, 8), ("tiny", tiny, 8)):
        d, i = rn.knn_index2(pts, K)
        out[f"{name}_points"], out[f"{name}_K"], out[f"{name}_dists"], out[f"{name}_idx"] = pts, np.int32(K), d, i
    out["uni_mean"] = rn.knn_mean(uni)
    out["clu_mean"] = rn.knn_mean(clu)
    q = r.choice(6000, 900, replace=False).astype(np.int32)
    n = r.choice(6000, 2500, replace=False).astype(np.int32)
    d, i = rn.knn_indexQ(uni, q, n, 4)
    out.update(q_q=q, q_n=n, q_K=np.int32(4), q_dists=d, q_idx=i)
    return out


def matching_cases():
    from test_matching import _ray_image, _targets
    rm = rn.ref_matching()
    r = np.random.default_rng(11)
    out = {}
    b, h, w = 2, 48, 64
    rays = _ray_image(b, h, w, 3)
    pts, p_init = _targets(rays, 4)
    for tag, (it, lam, thr) in (("ip10", (10, 1e-8, 1e-6)), ("ip3", (3, 1e-4, 1e-5))):
        p, c = rm.iter_proj(torch.from_numpy(rays), torch.from_numpy(pts), torch.from_numpy(p_init), it, lam, thr)
        out.update({f"{tag}_args": np.array([it, lam, thr]), f"{tag}_p": p.numpy(), f"{tag}_conv": c.numpy()})
    out.update(ip_rays=rays, ip_pts=pts, ip_pinit=p_init)
    # refine_matches: unit-norm fp16 descriptors (what MASt3R's head emits), matches near the border as well
    D11 = r.standard_normal((b, h, w, 24)).astype(np.float32)
    D11 /= np.linalg.norm(D11, axis=-1, keepdims=True)
    src = np.stack([np.clip(np.tile(np.arange(w), h) + r.integers(-5, 6, h * w), 0, w - 1),
                    np.clip(np.repeat(np.arange(h), w) + r.integers(-5, 6, h * w), 0, h - 1)], -1)
    D21 = np.stack([D11[i, src[:, 1], src[:, 0]] for i in range(b)]) + 0.05 * r.standard_normal((b, h * w, 24)).astype(np.float32)
    p1 = np.stack([np.stack([np.tile(np.arange(w), h), np.repeat(np.arange(h), w)], -1)] * b).astype(np.int64)
    D11h, D21h = D11.astype(np.float16), D21.astype(np.float16)
    (o16,) = rm.refine_matches(torch.from_numpy(D11h), torch.from_numpy(D21h), torch.from_numpy(p1), 4, 5)   # radius / dilation_max as in VSLAM/utils_matching.py:171-179
    (o16b,) = rm.refine_matches(torch.from_numpy(D11h), torch.from_numpy(D21h), torch.from_numpy(p1), 2, 2)
    (o32,) = rm.refine_matches(torch.from_numpy(D11h.astype(np.float32)), torch.from_numpy(D21h.astype(np.float32)), torch.from_numpy(p1), 3, 2)  # scalar_t = float on the same values
    out.update(rf_D11=D11h, rf_D21=D21h, rf_p1=p1, rf_out_r4d5=o16.numpy(), rf_out_r2d2=o16b.numpy(),
               rf32_out_r3d2=o32.numpy())
    return out


if __name__ == "__main__":
    assert rn.available(), "oracle/_ref not built: python oracle/ref_shim/build_ref.py"
    np.savez_compressed(os.path.join(HERE, "ref_knn.npz"), **knn_cases())
    np.savez_compressed(os.path.join(HERE, "ref_matching.npz"), **matching_cases())
    for f in ("ref_knn.npz", "ref_matching.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
