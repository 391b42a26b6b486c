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
