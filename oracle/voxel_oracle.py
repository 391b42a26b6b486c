"""ORACLE (test infrastructure only): SceneModel.update_voxel, numpy restatement (densification path).

Follows Reconstruct/scene/scene_models/h3dgsv3.py:227-316 step by step: a common voxel grid over old + new points
(origin = componentwise minimum, `floor((p - min) / voxel_size)` in float32), a linear hash with strides
(vmax_y * vmax_z, vmax_z, 1), the majority class of every occupied voxel (torch.unique + torch_scatter.scatter_max: among
equally frequent classes the SMALLEST class id wins, because the (voxel, class) pairs are visited in sorted order and
scatter_max keeps the first maximum), relabelling of the old points, and for the new points either the majority class of
the voxel they fall into or a fresh id `max_cls + 1 + rank` with rank = position of their hash among the sorted distinct
unmatched hashes.  The cold start (no old points, :244-255) returns (inverse index of the distinct hashes, their count).

PINNED: tests/golden/voxel_*.npz are produced by the reference's OWN method source executed on CPU
(tests/golden/make_golden_voxel.py; scatter_max bound to oracle/scatter_oracle.py); tests/test_voxel.py checks this
restatement against them bit for bit; the HIP path (artdeco_amd/csrc/voxel.hip) is checked against both.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def _hashes(all_p, voxel_size, reciprocal=False):
    min_c = all_p.min(0)
    # torch evaluates `tensor / python_float` as a true division on CPU and as a multiplication by the fp32 reciprocal on GPU
    # (BinaryDivTrueKernel.cu); the goldens come from the CPU execution of the reference, hence the default
    q = (all_p - min_c) * (F(1.0) / F(voxel_size)) if reciprocal else (all_p - min_c) / F(voxel_size)
    v_idx = np.floor(q).astype(np.int64)
    v_max = v_idx.max(0) + 1
    stride = np.array([v_max[1] * v_max[2], v_max[2], 1], dtype=np.int64)
    return (v_idx * stride).sum(1)


def update_voxel(new_xyz, xyz, cls_id, voxel_size=0.1, reciprocal=False):
    """new_xyz [M,3] f32, xyz [N,3] f32, cls_id [N,1] int64 -> (updated_orig [N,1], updated_new [M,1], new_voxel_count),
    or (updated_new [M,1], count) when N == 0."""
    new_xyz, xyz = np.asarray(new_xyz, dtype=F), np.asarray(xyz, dtype=F).reshape(-1, 3)
    M, N = len(new_xyz), len(xyz)
    if N == 0:
        h_new = _hashes(new_xyz, voxel_size, reciprocal)
        u, inv = np.unique(h_new, return_inverse=True)
        return inv.astype(np.int64)[:, None], len(u)
    cls = np.asarray(cls_id, dtype=np.int64).reshape(-1)
    max_cls = int(cls.max())
    h_all = _hashes(np.concatenate([xyz, new_xyz], 0), voxel_size, reciprocal)
    h_orig, h_new = h_all[:N], h_all[N:]
    uniq, inv = np.unique(h_orig, return_inverse=True)
    # majority class per voxel; ties -> smallest class id
    offset = max_cls + 1
    pair_u, pair_c = np.unique(inv * offset + cls, return_counts=True)
    vox, lab = pair_u // offset, pair_u % offset
    mode = np.zeros(len(uniq), dtype=np.int64)
    best = np.zeros(len(uniq), dtype=np.int64)
    for v, l, c in zip(vox, lab, pair_c):  # sorted by (voxel, class): strict improvement keeps the first maximum
        if c > best[v]:
            best[v], mode[v] = c, l
    updated_orig = mode[inv][:, None]
    pos = np.minimum(np.searchsorted(uniq, h_new), len(uniq) - 1)
    hit = uniq[pos] == h_new
    updated_new = np.zeros(M, dtype=np.int64)
    updated_new[hit] = mode[pos[hit]]
    count = 0
    if (~hit).any():
        u_new, inv_new = np.unique(h_new[~hit], return_inverse=True)
        count = len(u_new)
        updated_new[~hit] = inv_new + max_cls + 1
    return updated_orig, updated_new[:, None], count
