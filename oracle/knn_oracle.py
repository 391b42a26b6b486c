"""CPU oracle for simple_knn._C (distCUDA2 / distIndex2 / distIndexQ) -- TEST INFRASTRUCTURE only.

The reference's kernels (Reconstruct/submodules/simple-knn/simple_knn.cu:391-466, :524-576,
:150-186) compute the EXACT K nearest neighbours with d = dx*dx + dy*dy + dz*dz in fp32
(:400-401), self excluded by index, output order within a row unspecified (replace-the-max
policy :405-420), unfilled slots (FLT_MAX, -1).  This oracle is the brute-force definition of
that, chunked numpy, same fp32 expression and association ((dx*dx + dy*dy) + dz*dz).

Parity: UNPINNED by the reference (no KNN test exists); cross-checked against scipy's cKDTree.
"""
from __future__ import annotations

import numpy as np

FLT_MAX = np.finfo(np.float32).max


def _sqdist(q, c):
    dx = c[None, :, 0] - q[:, None, 0]
    dy = c[None, :, 1] - q[:, None, 1]
    dz = c[None, :, 2] - q[:, None, 2]
    return (dx * dx + dy * dy) + dz * dz


def knn_oracle(points, q_idx, n_idx, K, chunk=2048):
    """-> (dists [Q,K] ascending fp32, indices [Q,K] int32 original ids; (FLT_MAX, -1) when short)."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    q_idx = np.asarray(q_idx, dtype=np.int64)
    n_idx = np.asarray(n_idx, dtype=np.int64)
    Q = len(q_idx)
    out_d = np.full((Q, K), FLT_MAX, np.float32)
    out_i = np.full((Q, K), -1, np.int32)
    if len(n_idx) == 0:
        return out_d, out_i
    cand = pts[n_idx]
    for s in range(0, Q, chunk):
        qs = q_idx[s:s + chunk]
        d = _sqdist(pts[qs], cand)
        d[qs[:, None] == n_idx[None, :]] = np.inf  # self excluded by INDEX (duplicates stay)
        k = min(K, d.shape[1])
        part = np.argpartition(d, k - 1, axis=1)[:, :k]
        pd = np.take_along_axis(d, part, 1)
        order = np.argsort(pd, axis=1, kind="stable")
        pd = np.take_along_axis(pd, order, 1)
        pi = np.take_along_axis(part, order, 1)
        ok = np.isfinite(pd)
        out_d[s:s + len(qs), :k] = np.where(ok, pd, FLT_MAX)
        out_i[s:s + len(qs), :k] = np.where(ok, n_idx[pi], -1)
    return out_d, out_i


def dist_index2_oracle(points, K):
    n = np.arange(len(points))
    return knn_oracle(points, n, n, K)


def dist_cuda2_oracle(points):
    d, _ = dist_index2_oracle(points, 3)
    return ((d[:, 0] + d[:, 1]) + d[:, 2]) / np.float32(3.0)
