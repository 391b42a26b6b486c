#!/usr/bin/env python
"""Per-call time of SceneModel.update_voxel at map size: the device path (artdeco_amd.fused.update_voxel_device) next to the
torch chain the reference runs (h3dgsv3.py:227-316 restated with the same torch ops: 3 x torch.unique, scatter_max from the
drop-in, searchsorted, boolean-mask writes).  add_new_gaussians calls it once per LoD level (4x per important frame).

    python tools/bench_update_voxel.py [N M]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd import fused
from torch_scatter import scatter_max


def torch_chain(new_xyz, xyz, cls_id, voxel_size):
    num_orig = xyz.shape[0]
    cls_id_1d = cls_id.squeeze(-1)
    max_cls = cls_id_1d.max().item()
    all_p = torch.cat([xyz, new_xyz], dim=0)
    min_c = all_p.min(dim=0).values
    v_idx_all = torch.floor((all_p - min_c) / voxel_size).long()
    v_max = v_idx_all.max(dim=0).values + 1
    stride = torch.stack([v_max[1] * v_max[2], v_max[2], torch.ones_like(v_max[2])])
    h_all = (v_idx_all * stride).sum(dim=1)
    h_orig, h_new = h_all[:num_orig], h_all[num_orig:]
    unique_voxels, inv_idx = torch.unique(h_orig, return_inverse=True)
    offset = max_cls + 1
    pair_unique_ids, pair_counts = torch.unique(inv_idx * offset + cls_id_1d, return_counts=True)
    _, max_indices = scatter_max(pair_counts, pair_unique_ids // offset)
    voxel_mode_labels = (pair_unique_ids % offset)[max_indices]
    updated_orig = voxel_mode_labels[inv_idx].unsqueeze(-1)
    pos = torch.searchsorted(unique_voxels, h_new).clamp(max=unique_voxels.shape[0] - 1)
    mask = unique_voxels[pos] == h_new
    updated_new = torch.zeros(new_xyz.shape[0], dtype=torch.long, device=new_xyz.device)
    if mask.any():
        updated_new[mask] = voxel_mode_labels[pos[mask]]
    count = 0
    if (~mask).any():
        u_new_h, u_new_inv = torch.unique(h_new[~mask], return_inverse=True)
        count = u_new_h.shape[0]
        updated_new[~mask] = u_new_inv + max_cls + 1
    return updated_orig, updated_new.unsqueeze(-1), count


def main():
    N, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 50_000)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    centres = rng.uniform(-6, 6, (400, 3))
    pts = lambda k: (centres[rng.integers(0, 400, k)] + 0.4 * rng.standard_normal((k, 3))).astype(np.float32)
    xyz, new = torch.from_numpy(pts(N)).to(dev), torch.from_numpy(pts(M)).to(dev)
    cls = torch.from_numpy(rng.integers(0, N // 10, (N, 1)).astype(np.int64)).to(dev)

    def timed(fn, reps=10):
        for _ in range(2):
            r = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, r

    t_dev, rd = timed(lambda: fused.update_voxel_device(new, xyz, cls, 0.1))
    t_ref, rt = timed(lambda: torch_chain(new, xyz, cls, 0.1))
    same = bool(torch.equal(rd[0], rt[0]) and torch.equal(rd[1], rt[1]) and rd[2] == rt[2])
    print(json.dumps({"N": N, "M": M, "device_ms": t_dev, "torch_chain_ms": t_ref, "speedup": t_ref / t_dev, "results_identical": same,
                      "new_voxels": rd[2], "per_important_frame_ms": {"device": 4 * t_dev, "torch_chain": 4 * t_ref}}))


if __name__ == "__main__":
    main()
