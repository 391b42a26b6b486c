"""ORACLE (test infrastructure only): scatter_max / scatter_min with argument, sequential restatement.

pytorch_scatter is a third-party pip dependency of ARTDECO (module-level import at
Reconstruct/scene/scene_models/h3dgsv3.py:35, call :289), absent from /root/reference and not installable here.
Restated from its published CPU algorithm: walk the elements in order, replace on STRICT improvement (so the first
of equal maxima wins), groups that receive nothing get value 0 and argument n.  parity unpinned by the reference
(no test or golden vector exists for it); anchored on the call site's use (majority class per voxel).
"""
import numpy as np


def scatter_arg(src: np.ndarray, index: np.ndarray, dim_size: int | None = None, is_min: bool = False):
    n = src.shape[0]
    if dim_size is None:
        dim_size = int(index.max()) + 1 if n else 0
    out = np.zeros(dim_size, dtype=src.dtype)
    arg = np.full(dim_size, n, dtype=np.int64)
    for i in range(n):
        j = int(index[i])
        if arg[j] == n or (src[i] < out[j] if is_min else src[i] > out[j]):
            out[j] = src[i]
            arg[j] = i
    return out, arg
