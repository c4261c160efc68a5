"""RCCL on the one GPU a test box has: the sweep's only collective -- all_reduce(SUM) over the int64 [nSnr, 4] counter table
 -- through torch.distributed's 'nccl' backend (= RCCL on ROCm) with world_size 1, in a fresh child process, so
that the first multi-GPU run is not also the first time librccl is loaded and a communicator is built by this code."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', init_method='tcp://127.0.0.1:%(port)d', world_size=1, rank=0)
import neoradium_amd as nr
nr.random.setSeed(3)
car = nr.Carrier(numRbs=24, spacing=30)
p = nr.PDSCH(car.curBwp, numLayers=2, nID=car.cellId, modulation='16QAM')
p.setDMRS(configType=1, additionalPos=1)
ch = nr.CdlChannel(car.curBwp, 'C', delaySpread=100, carrierFreq=4e9, dopplerShift=5,
                   txAntenna=nr.AntennaPanel([1, 2], polarization='x'), rxAntenna=nr.AntennaPanel([1, 1], polarization='x'))
link = nr.PdschLink(p, ch, 0.5, baseGraphNo=1, numIter=10, freqDomain=False, chanEst='LS', decoder='f64')
snrs = [4.0, 8.0, 12.0]
table = nr.run_sweep(link, snrs, 12, seed=5, batch=8)            # all_reduce(SUM) of the device table through RCCL inside
# the same reduction spelled out on a device tensor, and a reduction that actually changes something (MAX of -x)
t = torch.from_numpy(table).to(link.dev)
dist.all_reduce(t)
m = -t.clone()
dist.all_reduce(m, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
print(json.dumps({'table': table.tolist(), 'again': t.cpu().tolist(), 'neg': m.cpu().tolist(), 'backend': dist.get_backend(),
                  'world': dist.get_world_size(), 'nccl_version': list(torch.cuda.nccl.version())}))
dist.barrier()
dist.destroy_process_group()
"""


def test_counter_all_reduce_through_rccl_world_1(dev):
    import neoradium_amd as nr
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, port=port)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().split('\n')[-1])
    assert out['backend'] == 'nccl' and out['world'] == 1
    # the same sweep in this process, no process group: identical table
    nr.random.setSeed(3)
    car = nr.Carrier(numRbs=24, spacing=30)
    p = nr.PDSCH(car.curBwp, numLayers=2, nID=car.cellId, modulation='16QAM')
    p.setDMRS(configType=1, additionalPos=1)
    ch = nr.CdlChannel(car.curBwp, 'C', delaySpread=100, carrierFreq=4e9, dopplerShift=5,
                       txAntenna=nr.AntennaPanel([1, 2], polarization='x'), rxAntenna=nr.AntennaPanel([1, 1], polarization='x'))
    link = nr.PdschLink(p, ch, 0.5, baseGraphNo=1, numIter=10, freqDomain=False, chanEst='LS', decoder='f64')
    ref = nr.run_sweep(link, [4.0, 8.0, 12.0], 12, seed=5, batch=8)
    assert np.array_equal(np.array(out['table']), ref) and np.array_equal(np.array(out['again']), ref)
    assert np.array_equal(np.array(out['neg']), -ref)
    assert ref[:, 1].min() > 0 and ref[0, 0] > ref[-1, 0]            # blocks were simulated; the low-SNR point fails more often


def test_bench_two_ranks_on_one_gpu_with_the_real_engine(dev):
    """`bench.py --gpus 2` with the REAL PdschLink: two ranks (children started before anything touches the GPU) share the one GPU of
    a test box, the collectives go through gloo (NRX_BENCH_BACKEND: two RCCL ranks cannot share a device).  One JSON line, n_gpus 2,
    every rank's slot range counted, and the counters equal those of ONE process running both ranks' timed slot ranges -- the
    device generator is keyed by the absolute slot number, so sharding must not change a single block."""
    import torch
    import neoradium_amd as nr
    import bench
    K, W, B = 2, 1, 32
    env = dict(os.environ, NRX_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', str(K), '--warmup', str(W), '--batch', str(B),
                        '--no-cpu', '--no-fast', '--no-allrows', '--no-twopass', '--no-cert', '--no-configs'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == K and out['warmup'] == W and out['scaling'] == 'weak'
    assert out['bler']['blocks'] == 2 * K * B * 72
    assert out['value'] > 0 and out['roofline']['frac'] > 0
    # one process, both ranks' timed slot ranges: rank r owns [r (K+W) B, (r+1)(K+W) B), its first W*B slots are warm-up
    link = bench.build_link(nr, decoder="f64")
    c = torch.zeros(4, dtype=torch.int64, device=link.dev)
    for rank in range(2):
        base = rank * (K + W) * B
        for k in range(K):
            link.run(base + (W + k) * B, B, out['config']['snr_db'], seed=123, counters=c)
    c = c.cpu().numpy()
    got = out['bler']
    assert [got['block_errors'], got['blocks'], got['bit_errors'], got['bits']] == [int(v) for v in c]


def test_bench_cfg5_two_ranks_shard_the_harq_processes(dev):
    """`bench.py --config cfg5 --gpus 2` with the REAL engine: two ranks on the one GPU of a test box (gloo collectives), each simulating
    its own HARQ processes (harq.py:626-631: independent streams), ONE all-reduce of the per-try counters -- the statistics of the timed
    rounds equal those of one process running all the processes (the generator is keyed by the absolute slot)."""
    import neoradium_amd as nr
    import bench
    K, W, B = 3, 1, 8
    env = dict(os.environ, NRX_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', 'cfg5', '--gpus', '2', '--steps', str(K), '--warmup', str(W),
                        '--batch', str(B)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == K and out['unit'] == 'transmissions/s' and out['config']['name'] == 'cfg5'
    assert sum(out['harq']['txBlocks']) == 2 * B * K and out['value'] > 0 and 0 < out['roofline']['frac'] < 1
    link = bench.build_link(nr, decoder="f64")
    _, st = link.run_harq(2 * B, W, out['config']['snr_db'], seed=123)
    before = [st[k].clone() for k in ('tx', 'rx')]
    stats, st = link.run_harq(2 * B, K, out['config']['snr_db'], seed=123, state=st)
    assert (st['tx'] - before[0]).cpu().tolist() == out['harq']['txBlocks'] and (st['rx'] - before[1]).cpu().tolist() == out['harq']['rxBlocks']
