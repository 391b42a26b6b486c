"""The N>1 path of bench.py (scene-per-rank partition, barrier, MAX/SUM metric all-reduce) on CPU with
gloo, world_size 2 -- the only collectives the hot path has."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from artdeco_amd import multigpu
    topo = multigpu.init("gloo")
    assert (topo.rank, topo.world) == (rank, world)
    mine = multigpu.scene_for_rank(list(range(5)), topo)
    multigpu.barrier()
    elapsed, sums = multigpu.aggregate(1.0 + rank, {"steps": 10.0 * (rank + 1), "frames": float(len(mine))}, torch.device("cpu"))
    q.put((rank, mine, elapsed, sums))
    multigpu.shutdown()


def test_scene_partition_and_metric_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]          # disjoint, complete partition
    for _, _, elapsed, sums in res:
        assert elapsed == 2.0                                       # MAX over ranks
        assert sums == {"frames": 5.0, "steps": 30.0}               # SUM over ranks


def test_single_process_is_a_noop():
    from artdeco_amd import multigpu
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    topo = multigpu.init("gloo")
    assert topo.world == 1
    e, s = multigpu.aggregate(3.0, {"steps": 7.0}, torch.device("cpu"))
    assert e == 3.0 and s == {"steps": 7.0}


def test_bench_entry_self_launches_two_ranks_on_gloo():
    """The driver's own invocation, `python bench.py --gpus N` with no launcher in front of it: bench.py re-executes itself
    under torch.distributed.run with N local ranks (127.0.0.1), every rank joins the process group, the barrier / MAX / SUM
    all-reduce path runs, and rank 0 prints exactly one JSON line with n_gpus = N.  (--cpu-dry-run replaces the GPU step by
    a constant CPU step; everything around it is the code the 8-GPU run executes.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--cpu-dry-run",
                        "--steps", "5", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["parallelism"] == "scene-per-rank x2"


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_entry_runs_two_ranks_on_one_gpu():
    """The GPU side of the N > 1 path, driver-run: `python bench.py --gpus 2 --backend gloo` started bare on the one-GPU box with
    ARTDECO_BENCH_SHARE_GPU=1 (both ranks on cuda:0): self-launch under torch.distributed.run, one scene per rank, the frame loop,
    barrier, MAX(elapsed) / SUM(frames) over ranks, ONE JSON line from rank 0 with n_gpus = 2, `value` = the whole job's frames over
    the slowest rank's wall time.  (scene per GPU, a barrier + one metric all-reduce; no scaling curve can be measured on
    one GPU -- this pins the code path the driver's 8-GPU run takes.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ARTDECO_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--gaussians", "200000", "--width", "512", "--height", "384",
                        "--no-extra-configs", "--no-cpu-baseline", "--no-frontend"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == "scene-per-gpu x2"
    # whole-job value: frames of BOTH ranks / max-over-ranks elapsed = 2 * steps / (ms_per_step * steps)
    assert abs(out["value"] - 2.0 * 1e3 / out["ms_per_step"]) <= 1e-6 * out["value"]
    assert out["config"]["optimisation_steps"] >= 2 * 3 * 10          # SUM over ranks of >= 10 optimisation steps per frame
    assert out["roofline"]["avg_launch_ms"] > 0


@pytest.mark.gpu
def test_rccl_barrier_and_metric_allreduce_one_rank():
    """The RCCL branch itself (round 6): `multigpu.init("nccl", device, force=True)` under torch.distributed.run with ONE rank on the one GPU
    (RCCL refuses two ranks on one device, so this is as far as a one-GPU box goes): process group bound to the device, barrier, the MAX / SUM
    all-reduce of the metrics on device tensors, shutdown."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "tests", "host", "rccl_one_rank.py")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["backend"] == "nccl" and out["world"] == 1 and out["allreduce_ok"]
    assert out["elapsed"] == 2.5 and out["sums"] == {"frames": 20.0, "steps": 270.0}
