#!/usr/bin/env python3
"""Opt-in two-pass decoding (PdschLink(firstPassIter=...)) at the metric configuration, next to the reference schedule.
NOT the headline number: the reference runs numIter iterations on every code block; here only blocks that fail the
CRC after the short first pass are decoded (again, from scratch) with all 50.

    python tools/archive/bench_two_pass.py [--first 12] [--snr 31] [--steps 3]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--first', type=int, default=12)
    ap.add_argument('--snr', type=float, default=31.0)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64)
    a = ap.parse_args()
    import neoradium_amd as nr
    res = {}
    for name, kw in (("reference_schedule", {}), ("two_pass_opt_in", dict(first=a.first))):
        link = bench.build_link(nr)
        if kw:
            link.firstPassIter = kw['first']
        link.run(0, a.batch, a.snr, seed=123)
        torch.cuda.synchronize()
        c = torch.zeros(4, dtype=torch.int64, device=link.dev)
        t0 = time.perf_counter()
        for k in range(a.steps):
            link.run((k + 1) * a.batch, a.batch, a.snr, seed=123, counters=c)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c = c.cpu().numpy()
        res[name] = dict(slots_per_s=a.batch * a.steps / dt, block_errors=int(c[0]), blocks=int(c[1]), bit_errors=int(c[2]))
    res["first_pass_iterations"] = a.first
    res["snr_db"] = a.snr
    res["same_block_errors"] = res["reference_schedule"]["block_errors"] == res["two_pass_opt_in"]["block_errors"]
    print(json.dumps(res))


if __name__ == '__main__':
    main()
