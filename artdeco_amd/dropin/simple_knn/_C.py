"""Drop-in for `simple_knn._C` (ext.cpp:15-19): distCUDA2, distIndex2, distIndexQ on the HIP kernels.

Imported by ARTDECO at Reconstruct/scene/scene_models/h3dgsv3.py:37 and
Reconstruct/webviewer/scene_models.py:37; called at webviewer/scene_models.py:1003 (`distIndex2(xyz, k)`).
"""
from __future__ import annotations

import torch

from artdeco_amd import _lib

_SUPPORTED_K = (1, 2, 3, 4, 5, 6, 7, 8, 12, 16)


def _points(points: torch.Tensor) -> torch.Tensor:
    _lib.require_cuda(points)
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError(f"points must be [P,3], got {tuple(points.shape)}")
    if points.dtype != torch.float32:
        raise TypeError("points must be float32")
    return points.contiguous()


def _ws(lib, dev, n):
    return torch.empty(int(lib.adk_knn_workspace_bytes(n)), dtype=torch.uint8, device=dev)


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """Mean squared distance to the 3 nearest neighbours, float[P] (spatial.cu:14-25)."""
    pts = _points(points)
    P = pts.shape[0]
    lib = _lib.load()
    with torch.cuda.device(pts.device):
        out = torch.zeros(P, dtype=torch.float32, device=pts.device)
        if P:
            ws = _ws(lib, pts.device, P)
            rc = lib.adk_knn_mean_dist3(pts.data_ptr(), P, out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_of(pts))
            _lib.check(rc, "adk_knn_mean_dist3")
    return out


def distIndex2(points: torch.Tensor, K: int):
    """-> [dists float[P*K] (squared), indices int32[P*K]] (spatial.cu:28-41)."""
    pts = _points(points)
    P, K = pts.shape[0], int(K)
    if K not in _SUPPORTED_K:
        raise NotImplementedError(f"distIndex2: K must be one of {_SUPPORTED_K}")
    lib = _lib.load()
    with torch.cuda.device(pts.device):
        dists = torch.zeros(P * K, dtype=torch.float32, device=pts.device)
        idx = torch.full((P * K,), -1, dtype=torch.int32, device=pts.device)
        if P:
            ws = _ws(lib, pts.device, P)
            rc = lib.adk_knn_index2(pts.data_ptr(), P, K, dists.data_ptr(), idx.data_ptr(), ws.data_ptr(), ws.numel(),
                                    _lib.stream_of(pts))
            _lib.check(rc, "adk_knn_index2")
    return [dists, idx]


def distIndexQ(points: torch.Tensor, q_indices: torch.Tensor, n_indices: torch.Tensor, K: int):
    """-> [dists float[Q*K], indices int32[Q*K]]; candidates restricted to n_indices (spatial.cu:43-58)."""
    pts = _points(points)
    _lib.require_cuda(q_indices, n_indices)
    if q_indices.dtype != torch.int32 or n_indices.dtype != torch.int32:
        raise TypeError("q_indices / n_indices must be int32 (the reference reads them as int*, spatial.cu:55)")
    q, nn = q_indices.contiguous(), n_indices.contiguous()
    P, Q, N, K = pts.shape[0], q.numel(), nn.numel(), int(K)
    if K not in _SUPPORTED_K:
        raise NotImplementedError(f"distIndexQ: K must be one of {_SUPPORTED_K}")
    lib = _lib.load()
    with torch.cuda.device(pts.device):
        dists = torch.zeros(Q * K, dtype=torch.float32, device=pts.device)
        idx = torch.full((Q * K,), -1, dtype=torch.int32, device=pts.device)
        if Q:
            ws = _ws(lib, pts.device, max(N, 1))
            rc = lib.adk_knn_indexQ(pts.data_ptr(), P, q.data_ptr(), Q, nn.data_ptr() if N else None, N, K,
                                    dists.data_ptr(), idx.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_of(pts))
            _lib.check(rc, "adk_knn_indexQ")
    return [dists, idx]
