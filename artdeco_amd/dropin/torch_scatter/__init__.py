"""Drop-in for the part of `torch_scatter` ARTDECO imports (pytorch_scatter is a CUDA-only wheel upstream).

`from torch_scatter import scatter_max` sits at module level of the hot scene model
(Reconstruct/scene/scene_models/h3dgsv3.py:35) and is called by `update_voxel` (:289) on 1-D int64 tensors
(counts grouped by voxel index).  `scatter_max` / `scatter_min` keep upstream's signature and return
`(out, arg)`; groups that receive nothing get `out = 0`, `arg = src.size(dim)` as upstream does.  Among equal
maxima the FIRST position is returned (upstream's CPU behaviour; its CUDA kernel picks one by a store race).
Only what the reference uses is implemented: 1-D `src`/`index`, no preallocated `out`.
"""
import torch

from artdeco_amd import _lib
from artdeco_amd import autoinstall as _autoinstall

_autoinstall.on_dropin_import()  # post-import hook: fused mapper paths on every SceneModel (ARTDECO_AMD_AUTOFUSE=0 disables)

_DTYPES = {torch.float32: 0, torch.int32: 1, torch.int64: 2}


def _scatter_arg(src, index, dim, out, dim_size, is_min):
    if out is not None:
        raise NotImplementedError("torch_scatter drop-in: preallocated `out` is not supported")
    if src.dim() != 1 or index.dim() != 1 or dim not in (0, -1):
        raise NotImplementedError("torch_scatter drop-in: only 1-D src/index (ARTDECO's call, h3dgsv3.py:289)")
    if index.shape != src.shape:
        raise ValueError("index must have the shape of src")
    if src.dtype not in _DTYPES:
        raise NotImplementedError(f"torch_scatter drop-in: dtype {src.dtype}")
    lib = _lib.load()
    _lib.require_cuda(src, index)
    src_c, idx_c = src.detach().contiguous(), index.detach().long().contiguous()
    n = src_c.numel()
    if dim_size is None:  # upstream: index.max() + 1 (a host read there as well)
        dim_size = int(idx_c.max()) + 1 if n > 0 else 0
    with torch.cuda.device(src.device):
        res = torch.empty(dim_size, dtype=src.dtype, device=src.device)
        arg = torch.empty(dim_size, dtype=torch.int64, device=src.device)
        rc = lib.adk_scatter_argmax(n, src_c.data_ptr(), _DTYPES[src.dtype], idx_c.data_ptr(), dim_size, int(is_min),
                                    res.data_ptr(), arg.data_ptr(), _lib.stream_of(src_c))
    _lib.check(rc, "adk_scatter_argmax")
    return res, arg


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    return _scatter_arg(src, index, dim, out, dim_size, False)


def scatter_min(src, index, dim=-1, out=None, dim_size=None):
    return _scatter_arg(src, index, dim, out, dim_size, True)


__all__ = ["scatter_max", "scatter_min"]
