"""Loader for oracle/_ref -- the reference's OWN kernels compiled for the host (oracle/ref_shim/build_ref.py).
TEST INFRASTRUCTURE ONLY: imported by tests/ and the golden-vector scripts, never by artdeco_amd/.

  ref_matching()            -> module with iter_proj(rays, pts, p_init, max_iter, lambda_init, cost_thresh) and
                               refine_matches(D11, D21, p1, radius, dilation)      (CPU torch tensors in and out;
                               VSLAM/backend/src/matching_kernels.cu:119-316, :25-116)
  knn_index2(points, K)     -> (dists [P,K] f32, idx [P,K] i32)   SimpleKNN::knn_index2  (simple_knn.cu:468-522)
  knn_indexQ(points, q, n, K)                                      SimpleKNN::knn_indexQ  (simple_knn.cu:596-651)
  knn_mean(points)          -> [P] f32                              SimpleKNN::knn         (simple_knn.cu:188-221)
"""
from __future__ import annotations

import ctypes
import importlib.util
import os

import numpy as np

from .ref_shim import build_ref

_knn = None
_match = None


def available() -> bool:
    """True when oracle/_ref can be used: built here from /root/reference, or prebuilt files present (GPU box)."""
    try:
        return build_ref.build()
    except RuntimeError:
        return False


def _knn_lib():
    global _knn
    if _knn is None:
        if not available():
            raise RuntimeError("oracle/_ref is not built (python oracle/ref_shim/build_ref.py needs /root/reference)")
        _knn = ctypes.CDLL(build_ref.targets()["knn"])
    return _knn


def ref_matching():
    global _match
    if _match is None:
        if not available():
            raise RuntimeError("oracle/_ref is not built (python oracle/ref_shim/build_ref.py needs /root/reference)")
        import torch  # noqa: F401  (libtorch must be loaded before the extension)
        spec = importlib.util.spec_from_file_location("ref_matching", build_ref.targets()["matching"])
        _match = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_match)
    return _match


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def knn_mean(points):
    pts = np.ascontiguousarray(points, np.float32)
    P = len(pts)
    out = np.zeros(P, np.float32)           # spatial.cu:21 torch::full({P}, 0.0)
    _knn_lib().ref_knn(ctypes.c_int(P), _p(pts, ctypes.c_float), _p(out, ctypes.c_float))
    return out


def knn_index2(points, K):
    pts = np.ascontiguousarray(points, np.float32)
    P = len(pts)
    d = np.zeros(P * K, np.float32)         # spatial.cu:35-36
    i = np.full(P * K, -1, np.int32)
    _knn_lib().ref_knn_index2(ctypes.c_int(K), ctypes.c_int(P), _p(pts, ctypes.c_float), _p(d, ctypes.c_float), _p(i, ctypes.c_int))
    return d.reshape(P, K), i.reshape(P, K)


def knn_indexQ(points, q_idx, n_idx, K):
    pts = np.ascontiguousarray(points, np.float32)
    q = np.ascontiguousarray(q_idx, np.int32)
    n = np.ascontiguousarray(n_idx, np.int32)
    P, Q, N = len(pts), len(q), len(n)
    d = np.zeros(Q * K, np.float32)         # spatial.cu:52-53
    i = np.full(Q * K, -1, np.int32)
    _knn_lib().ref_knn_indexQ(ctypes.c_int(K), ctypes.c_int(P), _p(pts, ctypes.c_float), ctypes.c_int(Q), _p(q, ctypes.c_int),
                              ctypes.c_int(N), _p(n, ctypes.c_int), _p(d, ctypes.c_float), _p(i, ctypes.c_int))
    return d.reshape(Q, K), i.reshape(Q, K)
