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


sys.path.insert(0, ROOT)
from bench import StubHarqLink as _StubHarqLink      # noqa: E402  (the host stand-in for PdschLink.run_harq; also behind `bench.py --config cfg5 --stub`)


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


def _harq_worker8(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from neoradium_amd.engine import run_harq_sharded, shard_slots
    link = _StubHarqLink()
    st, state = run_harq_sharded(link, 13, 5, 20.0, slot0=40)               # 13 processes over 8 ranks: shares of 2 and 1
    st2, _ = run_harq_sharded(link, 13, 4, 20.0, state=state, slot0=40)
    st3, _ = run_harq_sharded(link, 5, 3, 20.0, slot0=7)                     # fewer processes than ranks: empty shards join the collective
    q.put((rank, {k: np.asarray(v).tolist() for k, v in st2.items()}, {k: np.asarray(v).tolist() for k, v in st3.items()},
           shard_slots(0, 13, world, rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_harq_world8_process_count_not_divisible_by_the_world_size():
    """First-8-GPU-run insurance: eight gloo ranks, 13 HARQ processes (shares of 2 and 1) and 5 processes (three empty
    shards): contiguous disjoint shares that cover every process, and every rank ends with the single-process statistics."""
    sys.path.insert(0, ROOT)
    from neoradium_amd.engine import run_harq_sharded
    link = _StubHarqLink()
    _, state = run_harq_sharded(link, 13, 5, 20.0, slot0=40)
    want, _ = run_harq_sharded(link, 13, 4, 20.0, state=state, slot0=40)
    want = {k: np.asarray(v).tolist() for k, v in want.items()}
    want5, _ = run_harq_sharded(link, 5, 3, 20.0, slot0=7)
    want5 = {k: np.asarray(v).tolist() for k, v in want5.items()}
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 311) % 500
    procs = [ctx.Process(target=_harq_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    shares = [r[3] for r in res]
    assert sum(c for _, c in shares) == 13 and all(shares[i][0] + shares[i][1] == shares[i + 1][0] for i in range(world - 1))
    assert max(c for _, c in shares) - min(c for _, c in shares) <= 1
    for _, got, got5, _ in res:
        assert got == want and got5 == want5


@pytest.mark.parametrize("config", ["metric", "cfg3", "cfg5"])
def test_bench_gpus8_self_launch_gloo(config):
    """`python bench.py --gpus 8 --config ...` on the host (gloo + the --stub links; no GPU, no measurement): eight ranks, local rank ->
    device index mapping, disjoint slot ranges / process streams per rank, ONE JSON line from rank 0 whose counters are 8 x a
    single rank's share -- the code path the driver's first 8-GPU run takes."""
    import json
    import subprocess
    env = dict(os.environ, NRX_BENCH_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    K, W, B = 3, 1, 8
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--config', config, '--steps', str(K), '--warmup', str(W),
                        '--batch', str(B), '--stub', '--no-cpu'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 8 and out['steps'] == K and out['scaling'] == 'weak' and out['config']['name'] == config
    if config != 'cfg5':
        assert out['bler']['blocks'] == 8 * K * B * 72
        # rank r's timed slots: [r (K + W) B + W B, (r + 1)(K + W) B) -- disjoint, the union is what the counters saw
        slots = np.concatenate([np.arange(r_ * (K + W) * B + W * B, (r_ + 1) * (K + W) * B) for r_ in range(8)])
        assert len(np.unique(slots)) == 8 * K * B
        assert out['bler']['block_errors'] == int(((slots * 2654435761 + 123) % 97 < 13).sum() * 3)
        one = int(((np.arange(W * B, (K + W) * B) * 2654435761 + 123) % 97 < 13).sum() * 3)      # (rank 0's own share differs: other slots)
        assert out['bler']['bits'] == 8 * K * B * 606504 and one >= 0
        assert abs(out['value'] - 8 * K * B / (out['ms_per_step'] * 1e-3 * K)) < 1e-6 * out['value']
    else:
        # 8 ranks x B process streams, K timed rounds: every process transmits once per round, and the sharded run equals ONE process
        # simulating all 8 B streams (same absolute slots)
        assert sum(out['harq']['txBlocks']) == 8 * B * K
        link = _StubHarqLink()
        _, st = link.run_harq(8 * B, W, 27.0, seed=123)
        before = [st['tx'].clone(), st['rx'].clone()]
        _, st = link.run_harq(8 * B, K, 27.0, seed=123, state=st)
        assert (st['tx'] - before[0]).tolist() == out['harq']['txBlocks'] and (st['rx'] - before[1]).tolist() == out['harq']['rxBlocks']
