"""Dev tool (GPU): the certified early exit against the fixed schedule at the metric configuration.

For every SNR: the same slots through PdschLink with the reference's fixed schedule and with `certifiedExit`; every certified
block's hard bits must equal the fixed schedule's bit for bit, all CRC verdicts must agree.  Prints the exit histogram and the
times of both decoders.  `--flags` breaks the certificate on purpose (1: no sign conditions, 2: no closure, 3: neither).

    python tools/r4/cert_gpu_check.py [--snr 29 31 33] [--slots 32] [--stages 8 16] [--flags 0] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import neoradium_amd as nr                    # noqa: E402
import bench                                  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--snr', type=float, nargs='+', default=[31.0])
    ap.add_argument('--slots', type=int, default=32)
    ap.add_argument('--batches', type=int, default=1)
    ap.add_argument('--stages', type=int, nargs='+', default=[8, 16])
    ap.add_argument('--flags', type=int, default=0)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--sweeps', type=int, default=8)
    ap.add_argument('--standalone', action='store_true', help='the certificate as its own launch on the parked states')
    ap.add_argument('--json', default='')
    a = ap.parse_args()
    fixed = bench.build_link(nr, decoder="f64", num_iter=a.iters)
    cert = bench.build_link(nr, decoder="f64", num_iter=a.iters, certifiedExit=tuple(a.stages), certFlags=a.flags, certSweeps=a.sweeps, certInKernel=not a.standalone)
    C = fixed.cfg.C
    pay = fixed.cfg.cb_len - 24
    out = {}
    for snr in a.snr:
        tot = dict(blocks=0, certified=0, mismatch_blocks=0, verdict_diff=0, crc_ok=0, hist={})
        t_fixed = t_cert = 0.0
        for b in range(a.batches):
            slot0 = 100 + b * a.slots
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, d0 = fixed.run(slot0, a.slots, snr, seed=3, details="verdicts")
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            _, d1 = cert.run(slot0, a.slots, snr, seed=3, details="verdicts")
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if b > 0 or a.batches == 1:
                t_fixed += t1 - t0
                t_cert += t2 - t1
            ok0 = d0[0][1]['cb_ok'].reshape(-1).cpu().numpy()
            ok1 = d1[0][1]['cb_ok'].reshape(-1).cpu().numpy()
            tb0 = d0[0][1]['tb_out'].reshape(-1, pay).cpu().numpy()
            tb1 = d1[0][1]['tb_out'].reshape(-1, pay).cpu().numpy()
            ex = cert.last_exit_iter.cpu().numpy()
            certd = ex > 0
            diff = (tb0 != tb1).any(1)
            tot['blocks'] += len(ex)
            tot['certified'] += int(certd.sum())
            tot['mismatch_blocks'] += int((diff & certd).sum())
            tot['verdict_diff'] += int((ok0 != ok1).sum())
            tot['crc_ok'] += int(ok0.sum())
            tot.setdefault('uncertified_diff', 0)
            tot['uncertified_diff'] += int((diff & ~certd).sum())
            for k, v in zip(*np.unique(ex, return_counts=True)):
                tot['hist'][int(k)] = tot['hist'].get(int(k), 0) + int(v)
        tot['ms_fixed_per_batch'] = 1e3 * t_fixed / max(1, a.batches - (1 if a.batches > 1 else 0))
        tot['ms_cert_per_batch'] = 1e3 * t_cert / max(1, a.batches - (1 if a.batches > 1 else 0))
        import ctypes
        from neoradium_amd import _lib
        h = (ctypes.c_ulonglong * 16)()
        try:
            _lib.lib().nrx_debug_cert_sweeps(h, 1)
            tot['sweep_hist'] = list(h)
        except Exception as e:
            tot['sweep_hist'] = str(e)
        out[str(snr)] = tot
        print(snr, json.dumps(tot), flush=True)
    if a.json:
        os.makedirs(os.path.dirname(a.json) or '.', exist_ok=True)
        json.dump(dict(stages=a.stages, flags=a.flags, slots=a.slots, batches=a.batches, results=out), open(a.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
