"""bench.py's N>1 path (proof sharding, barrier-bracketed timing, max-over-ranks aggregation) on
CPU with the gloo backend, world_size 2.  The step function is a stand-in sleep: no GPU here."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.01 * (rank + 1))   # rank 1 is the slow one

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = bench.timed_region(step, 5, 2, dist.barrier, lambda: None)
    agg = bench.aggregate(elapsed, world, 5, reduce_max)
    q.put((rank, len(calls), elapsed, agg))
    dist.destroy_process_group()


def test_two_rank_sharded_bench_logic():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, e0, a0), (r1, n1, e1, a1) = res
    assert n0 == n1 == 7                        # 2 warmup + exactly 5 timed steps on each rank
    assert abs(a0["seconds"] - a1["seconds"]) < 1e-9   # both ranks agree on the max-over-ranks time
    assert a0["seconds"] >= 5 * 0.02 * 0.9      # bounded below by the slow rank
    assert abs(a0["value"] - world * 5 / a0["seconds"]) < 1e-9   # whole-job proofs/s over all ranks
