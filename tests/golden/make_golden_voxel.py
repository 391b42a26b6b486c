"""Generate tests/golden/voxel_*.npz with the REFERENCE's own `SceneModel.update_voxel`
(Reconstruct/scene/scene_models/h3dgsv3.py:227-316), compiled from the reference file (ast; nothing copied) and executed
on CPU.  `scatter_max` (torch_scatter, a pip dependency that is not in the tree) is bound to oracle/scatter_oracle.py.
Inputs are seeded (`cases()` below, shared with tests/test_voxel.py) and not stored.  Build container only.

    python tests/golden/make_golden_voxel.py
"""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_FILE = "/root/reference/Reconstruct/scene/scene_models/h3dgsv3.py"

from oracle import scatter_oracle  # noqa: E402


def scatter_max(src, index):
    out, arg = scatter_oracle.scatter_arg(src.numpy(), index.numpy())
    return torch.from_numpy(out), torch.from_numpy(arg)


def cases():
    """name -> (new_xyz [M,3], xyz [N,3], cls_id [N,1], voxel_size): clustered points so that voxels hold several points and
    classes, ties included; a cold start; new points that all fall into known voxels; and ones that all open new voxels."""
    out = {}
    for name, seed, N, M, vs, shift in (("voxel_mixed", 0, 4000, 900, 0.1, 0.3), ("voxel_coarse", 1, 2500, 700, 0.35, 0.0),
                                        ("voxel_all_new", 2, 1500, 400, 0.1, 50.0), ("voxel_cold", 3, 0, 1200, 0.1, 0.0),
                                        ("voxel_all_known", 4, 3000, 500, 0.5, 0.0)):
        rng = np.random.default_rng(seed)
        centres = rng.uniform(-2, 2, (40, 3))
        pts = lambda k: (centres[rng.integers(0, 40, k)] + 0.15 * rng.standard_normal((k, 3))).astype(np.float32)
        xyz = pts(N)
        new = pts(M) + np.float32(shift)
        if name == "voxel_all_known" and N:
            new = (xyz[rng.integers(0, N, M)] + 1e-4).astype(np.float32)
        cls = rng.integers(0, max(N // 12, 1), (N, 1)).astype(np.int64)
        out[name] = (new, xyz, cls, vs)
    return out


def main():
    tree = ast.parse(open(REF_FILE).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SceneModel")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "update_voxel")
    ns = {"torch": torch, "scatter_max": scatter_max}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF_FILE, "exec"), ns)
    update_voxel = ns["update_voxel"]
    for name, (new, xyz, c, vs) in cases().items():
        res = update_voxel(None, torch.from_numpy(new), torch.from_numpy(xyz), torch.from_numpy(c), vs)
        arrs = {f"out{i}": (r.numpy() if torch.is_tensor(r) else np.int64(r)) for i, r in enumerate(res)}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), in_sum=np.float64(new.astype(np.float64).sum() + xyz.astype(np.float64).sum()),
                            **arrs)
        print(name, [getattr(a, "shape", a) for a in arrs.values()], "new voxels:", int(res[-1]))


if __name__ == "__main__":
    main()
