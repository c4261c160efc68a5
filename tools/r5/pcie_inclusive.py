"""Round-5 tool (GPU): the metric configuration with HOST buffers at the boundary -- what `value` never includes.

The throughput mode generates its inputs on the device; the class surface / parity mode takes the transport blocks and the noise draws
from the host (NumPy in, NumPy out).  This times B slots per step including the host -> device copies of (tb_bits, noise) and the
device -> host copies of the decoded transport blocks and CRC verdicts, from pageable and from pinned host memory.

    python tools/r5/pcie_inclusive.py [--batch 256] [--steps 4] [--snr 31]
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
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--snr', type=float, default=31.0)
    a = ap.parse_args()
    link = bench.build_link(nr, decoder='f64')
    B = a.batch
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(1)
    tb = torch.from_numpy(rng.integers(0, 2, (B, link.tbs)).astype(np.uint8))
    z = rng.standard_normal((B, link.nr, link.slot_len[0] + link.max_delay, 2))
    zc = torch.from_numpy(z[..., 0] + 1j * z[..., 1])
    del z
    bytes_in = tb.numel() + zc.numel() * 16
    res = {}
    for kind in ('pageable', 'pinned'):
        tbh, zch = (tb.pin_memory(), zc.pin_memory()) if kind == 'pinned' else (tb, zc)
        out_tb = torch.empty((B, link.tbs), dtype=torch.uint8, pin_memory=(kind == 'pinned'))
        out_ok = None

        def step(s):
            nonlocal out_ok
            _, dv = link.run(s * B, B, a.snr, tb_bits=tbh.to(dev, non_blocking=True), noise=zch.to(dev, non_blocking=True), details="verdicts")
            d = dv[0][1]
            out_tb.copy_(d['tb_out'][:, :link.tbs], non_blocking=True)
            out_ok = d['cb_ok'].cpu()
        step(0)
        torch.cuda.synchronize()
        t0 = time.time()
        for s in range(a.steps):
            step(1 + s)
        torch.cuda.synchronize()
        dt = time.time() - t0
        res[kind] = dict(slots_per_s=B * a.steps / dt, ms_per_step=1e3 * dt / a.steps)
    # double-buffered: step s + 1's inputs cross the bus on a copy stream while step s computes, results leave on the copy stream too
    tbh, zch = tb.pin_memory(), zc.pin_memory()
    out_tb = [torch.empty((B, link.tbs), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    out_ok = [torch.empty((B * 72,), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    cs = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    bufs = [(torch.empty_like(tb, device=dev), torch.empty_like(zc, device=dev)) for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]
    for e in free:
        e.record(main)

    def prefetch(s):
        i = s & 1
        with torch.cuda.stream(cs):
            cs.wait_event(free[i])
            bufs[i][0].copy_(tbh, non_blocking=True)
            bufs[i][1].copy_(zch, non_blocking=True)
            ready[i].record(cs)

    keep = [None, None]

    def compute(s):
        i = s & 1
        main.wait_event(ready[i])
        _, dv = link.run(s * B, B, a.snr, tb_bits=bufs[i][0], noise=bufs[i][1], details="verdicts")
        keep[i] = dv[0][1]
        free[i].record(main)
        done[i].record(main)
        with torch.cuda.stream(cs):
            cs.wait_event(done[i])
            keep[i]['tb_out'].record_stream(cs)
            keep[i]['cb_ok'].record_stream(cs)
            out_tb[i].copy_(keep[i]['tb_out'][:, :link.tbs], non_blocking=True)
            out_ok[i].copy_(keep[i]['cb_ok'].reshape(-1)[:B * 72], non_blocking=True)
    prefetch(0)
    compute(0)
    torch.cuda.synchronize()
    t0 = time.time()
    prefetch(1)
    for s in range(1, a.steps + 1):
        if s < a.steps:
            prefetch(s + 1)
        compute(s)
    torch.cuda.synchronize()
    dt = time.time() - t0
    res['pinned_double_buffered'] = dict(slots_per_s=B * a.steps / dt, ms_per_step=1e3 * dt / a.steps)
    del bufs, keep
    # the same steps with the inputs already on the device (parity mode's kernels, no copies)
    tbd, zcd = tb.to(dev), zc.to(dev)
    link.run(0, B, a.snr, tb_bits=tbd, noise=zcd)
    torch.cuda.synchronize()
    t0 = time.time()
    for s in range(a.steps):
        link.run((1 + s) * B, B, a.snr, tb_bits=tbd, noise=zcd)
    torch.cuda.synchronize()
    dt = time.time() - t0
    res['resident'] = dict(slots_per_s=B * a.steps / dt, ms_per_step=1e3 * dt / a.steps)
    out = dict(batch=B, steps=a.steps, snr_db=a.snr, host_to_device_bytes_per_step=int(bytes_in), device_to_host_bytes_per_step=int(B * link.tbs + B * 72),
               **res)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
