#!/usr/bin/env python
"""Where the tile-local scatter's time goes: times adk_bin_local_scatter alone (HIP events, 20 launches) on the bench scene for the library
named by ARTDECO_HIP_LIB.  Lab builds: -DBIN_LAB=1 (LDS atomics only, no scattered stores), -DBIN_LAB=2 (same stores, consecutive addresses).
Their `pairs` are garbage: nothing downstream runs here.

    ARTDECO_HIP_LIB=... python tools/lab/bin_scatter_lab.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from artdeco_amd import _lib
from oracle import gsplat_oracle as go   # scene generator only (tool, not product)

N, W, H = 1_000_000, 1920, 1080
dev = torch.device("cuda:0")
lib = _lib.load()
sc = go.synthetic_scene(N, W, H, seed=0)
t = lambda a: a.to(dev).contiguous()
means, quats, scales, opac = t(sc["means"]), t(sc["quats"]), t(sc["scales"]), t(sc["opacities"])
viewmat, K = t(sc["viewmat"]), t(sc["K"])
i32 = dict(dtype=torch.int32, device=dev)
rec = torch.empty(N, 12, dtype=torch.float32, device=dev)
radii, depth_keys, gauss_ids, tpg = torch.empty(N, 2, **i32), torch.empty(N, **i32), torch.empty(N, **i32), torch.empty(N, **i32)
stream = torch.cuda.current_stream().cuda_stream
colors = torch.rand(N, 3, device=dev)
rc = lib.adk_project_fwd(N, means.data_ptr(), quats.data_ptr(), scales.data_ptr(), opac.data_ptr(), colors.data_ptr(), 0, 0, 0, 1, viewmat.data_ptr(),
                         K.data_ptr(), W, H, 0.01, 0.01, 1e10, 0.0, 1, rec.data_ptr(), radii.data_ptr(), depth_keys.data_ptr(), gauss_ids.data_ptr(),
                         tpg.data_ptr(), stream)
assert rc == 0, rc
tile_w, tile_h = (W + 15) // 16, (H + 15) // 16
offsets = torch.empty(tile_h, tile_w, **i32)
stats = torch.empty(2, dtype=torch.int64, device=dev)
table = torch.empty(int(lib.adk_bin_local_workspace_bytes(W, H)) + 256, dtype=torch.uint8, device=dev)
tbase = (table.data_ptr() + 255) & ~255
tbytes = table.numel() - (tbase - table.data_ptr())
assert lib.adk_bin_local_count(N, tpg.data_ptr(), rec.data_ptr(), W, H, offsets.data_ptr(), stats.data_ptr(), tbase, tbytes, stream) == 0
n_isects, max_tile = (int(x) for x in stats.cpu())
pairs = torch.empty(n_isects, dtype=torch.int64, device=dev)


def run():
    assert lib.adk_bin_local_scatter(N, n_isects, depth_keys.data_ptr(), tpg.data_ptr(), rec.data_ptr(), W, H, offsets.data_ptr(), tbase, tbytes,
                                     pairs.data_ptr(), stream) == 0


for _ in range(3):
    run()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(f"{os.environ.get('ARTDECO_HIP_LIB', 'default')}: I={n_isects} max_tile={max_tile} scatter median {ts[10]:.1f} us  min {ts[0]:.1f} us")
