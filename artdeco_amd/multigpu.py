"""Scene-per-GPU scaling harness.

ARTDECO has no collective communication; scenes are independent.  The 8 x MI355X configuration is
therefore N replicas of the single-GPU pipeline, one process per GPU, and the ONLY data that
crosses xGMI is (i) a start/stop barrier and (ii) one SUM/MAX all-reduce of a handful of fp64
scalars (~64 B -- latency-bound, so RCCL's one-shot path; ring bandwidth is irrelevant here).
Backend-agnostic so the same code is covered on CPU with gloo (tests/test_multigpu.py).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class Topology:
    rank: int
    local_rank: int
    world: int


def topology_from_env() -> Topology:
    return Topology(int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
                    int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str, device: torch.device | None = None, force: bool = False) -> Topology:
    """Join the job launched by torch.distributed.run (env:// rendezvous on 127.0.0.1).  A lone rank needs no process group and gets none,
    unless `force` (or ARTDECO_AMD_DIST_FORCE=1) asks for one -- which is how the RCCL branch is exercised on a one-GPU box
    (tests/test_multigpu.py::test_rccl_barrier_and_metric_allreduce_one_rank: backend "nccl" IS RCCL on ROCm)."""
    topo = topology_from_env()
    force = force or os.environ.get("ARTDECO_AMD_DIST_FORCE", "0") == "1"
    if (topo.world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return topo


def scene_for_rank(scenes: list, topo: Topology) -> list:
    """Static partition of independent scenes: scene i -> rank i % world (no data-path collective)."""
    return [s for i, s in enumerate(scenes) if i % topo.world == topo.rank]


def barrier(device: torch.device | None = None) -> None:
    if dist.is_initialized():
        if device is not None and device.type == "cuda" and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[device.index if device.index is not None else torch.cuda.current_device()])   # RCCL: name the device, no guess
        else:
            dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def aggregate(elapsed_s: float, sums: dict[str, float], device: torch.device) -> tuple[float, dict[str, float]]:
    """MAX over ranks of the elapsed time, SUM over ranks of the additive metrics."""
    keys = sorted(sums)
    if dist.is_initialized() and dist.get_backend() == "gloo":
        device = torch.device("cpu")            # gloo reduces host tensors; 64 bytes either way
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    v = torch.tensor([float(sums[k]) for k in keys], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return float(t.item()), {k: float(x) for k, x in zip(keys, v.tolist())}


def shutdown() -> None:
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
