"""Time SparseGaussianAdam.add_and_prune (optimizers.py:163-219) at 1 M Gaussians + 50 k new ones, 95 % kept:
the torch restatement (harness.mapper._add_and_prune) vs artdeco_amd.fused.fused_add_and_prune.
usage (GPU box, repo root): python tools/bench_add_and_prune.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from artdeco_amd import fused
from harness import mapper  # noqa: E402
from tests.test_add_and_prune import ALL, _clone, _extension, _optimizer  # noqa: E402

dev = torch.device("cuda:0")
N, E = 1_000_000, 50_000
base = _optimizer(dev, N, 125_000, seed=0)
ext = _extension(dev, E, 6_000, seed=1, keys=ALL)
mask = (torch.rand(N, device=dev) < 0.95)
for name, fn in (("torch (reference body)", mapper._add_and_prune), ("fused (adk_compact_*)", fused.fused_add_and_prune)):
    ts = []
    for _ in range(5):
        o = _clone(base)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            fn(o, ext, mask)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name}: {min(ts):.2f} ms (best of 5), rows {o.params['xyz']['val'].shape[0]}")
