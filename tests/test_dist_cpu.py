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


class _StubLink:
    """Stands in for PdschLink on the host: per-slot outcomes are a pure function of (absolute slot, SNR, seed), which is
    what the device generator's keying guarantees for the real engine."""
    dev = torch.device('cpu')

    def run(self, slot0, n_slots, snr_db, seed=0, counters=None):
        slots = np.arange(slot0, slot0 + n_slots)
        errs = ((slots * 2654435761 + seed * 97 + int(snr_db * 10) * 31) % 101 < 40 - int(snr_db)).astype(np.int64)
        counters += torch.tensor([errs.sum() * 2, n_slots * 16, errs.sum() * 500, n_slots * 129128], dtype=torch.int64)
        return counters


def _sweep_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from neoradium_amd.engine import run_sweep, shard_slots
    table = run_sweep(_StubLink(), [8.0, 10.5, 13.0], 37, seed=5, batch=8, slot0=100)
    q.put((rank, table.tolist(), shard_slots(100, 37, world, rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_sweep_world2_equals_single_process():
    """neoradium_amd.run_sweep (slot ranges per rank, one all-reduce of the [nSnr, 4] table) with two gloo ranks gives
    every rank the table a single process computes; the ranks' slot ranges are disjoint and cover the sweep."""
    sys.path.insert(0, ROOT)
    from neoradium_amd.engine import run_sweep
    want = run_sweep(_StubLink(), [8.0, 10.5, 13.0], 37, seed=5, batch=8, slot0=100).tolist()
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 77) % 500
    procs = [ctx.Process(target=_sweep_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[2] for r in res] == [(100, 19), (119, 18)]
    for _, table, _ in res:
        assert table == want


class _StubHarqLink:
    """Stands in for PdschLink.run_harq on the host: whether process p's transmission of round k decodes is a pure function of its
    ABSOLUTE slot (slot0 + k * n_proc_total + p) -- what the device generator's keying gives the real engine -- and the per-try
    bookkeeping is the engine's (harq.py:185-199)."""
    dev = torch.device('cpu')

    def run_harq(self, n_proc, n_rounds, snr_db, state=None, maxTries=4, slot0=0, proc_offset=0, n_proc_total=None, **kw):
        n_proc_total = n_proc if n_proc_total is None else n_proc_total
        if state is None:
            state = dict(tries=np.zeros(n_proc, dtype=np.int64), tx=torch.zeros(maxTries, dtype=torch.int64), rx=torch.zeros(maxTries, dtype=torch.int64),
                         tx_bits=torch.zeros(maxTries, dtype=torch.int64), rx_bits=torch.zeros(maxTries, dtype=torch.int64),
                         timeouts=torch.zeros(1, dtype=torch.int64), next_slot=int(slot0))
        for _ in range(n_rounds):
            s0 = state['next_slot'] + proc_offset
            slots = np.arange(s0, s0 + n_proc)
            ok = ((slots * 2654435761 + 17 * state['tries']) % 11) < (2 + 2 * state['tries'])      # later tries decode more often
            for t, o in zip(state['tries'], ok):
                state['tx'][t] += 1
                state['tx_bits'][t] += 1000
                state['rx'][t] += int(o)
                state['rx_bits'][t] += 1000 * int(o)
            nxt = state['tries'] + 1
            timeout = (~ok) & (nxt == maxTries)
            state['timeouts'] += int(timeout.sum())
            state['tries'] = np.where(ok | timeout, 0, nxt)
            state['next_slot'] = s0 - proc_offset + n_proc_total
        from neoradium_amd.engine import harq_stats
        return harq_stats(state['tx'].numpy(), state['rx'].numpy(), state['tx_bits'].numpy(), state['rx_bits'].numpy(), int(state['timeouts'])), state


def _harq_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from neoradium_amd.engine import run_harq_sharded
    link = _StubHarqLink()
    st, state = run_harq_sharded(link, 13, 5, 20.0, slot0=40)
    st2, _ = run_harq_sharded(link, 13, 4, 20.0, state=state, slot0=40)          # continued from this rank's shard
    q.put((rank, {k: np.asarray(v).tolist() for k, v in st2.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_harq_world2_equals_single_process():
    """neoradium_amd.run_harq_sharded (HARQ processes split over the ranks as independent streams, harq.py:626-631, one all-reduce of
    the per-try counters) with two gloo ranks gives every rank the statistics one process computes for all the processes."""
    sys.path.insert(0, ROOT)
    from neoradium_amd.engine import run_harq_sharded
    link = _StubHarqLink()
    _, state = run_harq_sharded(link, 13, 5, 20.0, slot0=40)
    want, _ = run_harq_sharded(link, 13, 4, 20.0, state=state, slot0=40)
    want = {k: np.asarray(v).tolist() for k, v in want.items()}
    assert sum(want['txBlocks']) == 13 * 9 and want['numTimeouts'] >= 0
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 191) % 500
    procs = [ctx.Process(target=_harq_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, got in res:
        assert got == want


def test_bench_gpus2_self_launch_gloo():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (torch.distributed.run children), each rank
    takes its own slot range, the counters are all-reduced and rank 0 prints ONE JSON line with n_gpus == 2.  Runs the
    real bench.py code path (launcher, rank bookkeeping, timed_steps, collectives) on the host: gloo backend + the
    --stub link (no GPU here)."""
    import json
    import subprocess
    env = dict(os.environ, NRX_BENCH_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '8',
                        '--stub', '--no-cpu'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1 and out['scaling'] == 'weak'
    # both ranks' slot ranges counted: 2 ranks x 3 steps x 8 slots x 72 code blocks
    assert out['bler']['blocks'] == 2 * 3 * 8 * 72
    slots = np.concatenate([np.arange(r_ * 4 * 8 + 8, (r_ + 1) * 4 * 8) for r_ in range(2)])      # timed slots of each rank
    assert out['bler']['block_errors'] == int(((slots * 2654435761 + 123) % 97 < 13).sum() * 3)
    # a launcher whose world size contradicts --gpus is an error, not a silent single-GPU run
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--stub', '--no-cpu'], env=env2,
                        capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and 'WORLD_SIZE' in r2.stderr
