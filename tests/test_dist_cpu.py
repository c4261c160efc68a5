"""CPU, world_size 2, gloo: the multi-GPU path of bench.py / the engine is 'disjoint slot ranges per rank + ONE
all-reduce(SUM) of the int64[4] error counters' -- checked here with two processes on the host."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def slot_ranges(world, steps, warmup, batch):
    """The sharding rule bench.py uses: rank r owns slots [r*(K+W)*B, (r+1)*(K+W)*B)."""
    return [(r * (steps + warmup) * batch, (r + 1) * (steps + warmup) * batch) for r in range(world)]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = slot_ranges(world, 3, 1, 8)[rank]
    # stand-in per-slot outcome that depends only on the absolute slot index (as the device RNG keying guarantees)
    slots = np.arange(lo, hi)
    errs = (slots * 2654435761 % 97 < 13).astype(np.int64)
    counters = torch.tensor([errs.sum() * 3, len(slots) * 72, errs.sum() * 1000, len(slots) * 606504], dtype=torch.int64)
    t = torch.tensor([0.5 + 0.25 * rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(counters)                              # the path's only collective
    dist.all_reduce(t, op=dist.ReduceOp.MAX)               # bench timing: max over ranks
    q.put((rank, counters.tolist(), float(t)))
    dist.barrier()
    dist.destroy_process_group()


def test_counter_allreduce_world2():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 500
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # reference: the same statistic over the union of the ranks' slot ranges
    ranges = slot_ranges(world, 3, 1, 8)
    assert ranges[0][1] == ranges[1][0] and ranges[0][0] == 0          # disjoint + contiguous
    slots = np.arange(ranges[0][0], ranges[-1][1])
    errs = (slots * 2654435761 % 97 < 13).astype(np.int64)
    want = [int(errs.sum() * 3), len(slots) * 72, int(errs.sum() * 1000), len(slots) * 606504]
    for rank, counters, tmax in res:
        assert counters == want and tmax == 0.75
