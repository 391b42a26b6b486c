"""Run under `python -m torch.distributed.run --nproc-per-node 1` on a GPU box: artdeco_amd.multigpu over RCCL (backend "nccl") with one rank --
process-group creation bound to the device, the start / stop barrier and the MAX / SUM metric all-reduce on DEVICE tensors.
Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from artdeco_amd import multigpu  # noqa: E402

dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
topo = multigpu.init("nccl", dev, force=True)
import torch.distributed as dist  # noqa: E402

assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
multigpu.barrier(dev)
elapsed, sums = multigpu.aggregate(2.5, {"frames": 20.0, "steps": 270.0}, dev)
x = torch.arange(8, dtype=torch.float64, device=dev)
dist.all_reduce(x)                      # a plain RCCL all-reduce of device memory, for good measure
multigpu.barrier(dev)
out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "rank": topo.rank, "elapsed": elapsed, "sums": sums,
       "allreduce_ok": bool(torch.equal(x.cpu(), torch.arange(8, dtype=torch.float64))),
       "nccl_version": list(torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None}
multigpu.shutdown()
print(json.dumps(out), flush=True)
