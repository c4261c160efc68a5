#!/usr/bin/env python3
"""Polar control path (BASELINE cfg4) throughput: batched DCI blind-decode candidates, SCL list 8.

    python tests/tools/bench_polar.py [--A 64] [--E 864] [--n 32768] [--reps 5]

Prints decoded candidates/s for the SCL kernel alone and for rate-recover + decode, plus the oracle's (NumPy port of
the reference's recursive decoder) time per candidate on a few rows as the CPU baseline."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--A', type=int, default=64)
    ap.add_argument('--E', type=int, default=864)
    ap.add_argument('--n', type=int, default=32768)
    ap.add_argument('--L', type=int, default=8)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--snr', type=float, default=-4.0)
    ap.add_argument('--cpu-rows', type=int, default=8)
    a = ap.parse_args()
    import torch
    from neoradium_amd.polar import PolarEncoder, PolarDecoder
    from neoradium_amd import ops
    dev = torch.device('cuda:0')
    enc, dec = PolarEncoder(a.A, a.E, 'dci'), PolarDecoder(a.A, a.E, 'dci', sclListSize=a.L)
    rng = np.random.default_rng(0)
    tbn = rng.integers(0, 2, (a.n, a.A)).astype(np.uint8)
    tb = torch.from_numpy(tbn).to(dev)
    crc = ops.crc(tb, '24C')
    cbs = torch.cat([tb, crc], 1).contiguous()
    rm = enc.rateMatchDevice(enc.encodeDevice(cbs))
    sig = 10 ** (-a.snr / 20)
    g = torch.Generator(device=dev).manual_seed(7)
    llr = 2 * (1 - 2 * rm.double() + sig * torch.randn(rm.shape, dtype=torch.float64, device=dev, generator=g)) / sig ** 2
    rr = dec.recoverRateDevice(llr)
    msg, ok = dec.decodeDevice(rr)
    torch.cuda.synchronize()
    good = (msg[:, :a.A] == tb).all(1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_dec = t_all = 0.0
    for _ in range(a.reps):
        ev[0].record()
        rr = dec.recoverRateDevice(llr)
        ev[1].record()
        msg, ok = dec.decodeDevice(rr)
        ev[2].record()
        torch.cuda.synchronize()
        t_dec += ev[1].elapsed_time(ev[2]) * 1e-3
        t_all += ev[0].elapsed_time(ev[2]) * 1e-3
    from oracle.polar import PolarCode
    pc = PolarCode(a.A, a.E, 'dci', a.L)
    rows = rr[:a.cpu_rows].cpu().numpy()
    t0 = time.perf_counter()
    for r in rows:
        pc.decode(r[None])
    t_cpu = (time.perf_counter() - t0) / len(rows)
    print(json.dumps({'A': a.A, 'E': a.E, 'N': dec.polarCodeSize, 'K': dec.codeBlockSize, 'L': a.L, 'n': a.n,
                      'bler': float(1 - good.double().mean()), 'crc_fail': float(1 - ok.double().mean()),
                      'scl_cand_per_s': a.n * a.reps / t_dec, 'recover_plus_scl_cand_per_s': a.n * a.reps / t_all,
                      'scl_ms_per_launch': 1e3 * t_dec / a.reps, 'oracle_cpu_ms_per_cand': 1e3 * t_cpu}))


if __name__ == '__main__':
    main()
