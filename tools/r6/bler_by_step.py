#!/usr/bin/env python3
"""Round-6 tool (GPU): block-error rate of every 256-slot step of the metric configuration at 31 dB (certified schedule: the same verdicts
as the fixed one, profiles/r5_soak_400_steps.txt), steps 0 ... N-1 in bench.py's slot numbering -- to pick, for bench.py's
`certified_early_exit`, a slot range whose BLER is about 10 % beside the bench's own first steps (a deep fade: 36 %).

    python tools/r6/bler_by_step.py [--steps 400] > profiles/r6_bler_by_step.json
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import neoradium_amd as nr                    # noqa: E402
import bench                                  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--snr', type=float, default=31.0)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--window', type=int, default=8)
    a = ap.parse_args()
    link = bench.build_link(nr, decoder='f64', certifiedExit=(8, 16))
    per = []
    for k in range(a.steps):
        c = torch.zeros(4, dtype=torch.int64, device=link.dev)
        link.run(k * a.batch, a.batch, a.snr, seed=123, counters=c)
        c = c.cpu().numpy()
        per.append(int(c[0]))
    blocks = a.batch * link.cfg.C
    bler = np.asarray(per) / blocks
    w = a.window
    win = np.convolve(bler, np.ones(w) / w, mode='valid')                  # mean BLER of steps [i, i + w)
    best = int(np.argmin(np.abs(win - 0.10)))
    print(json.dumps({"snr_db": a.snr, "slots_per_step": a.batch, "blocks_per_step": blocks, "steps": a.steps,
                      "bler_all_steps": float(bler.mean()), "block_errors_per_step": per,
                      "window_steps": w, "window_closest_to_10_percent": {"first_step": best, "bler": float(win[best])},
                      "bench_first_steps": {"steps": "2..21 (2 warm-up steps)", "bler": float(bler[2:22].mean())}}))


if __name__ == '__main__':
    main()
