#!/usr/bin/env python3
"""BASELINE cfg5: HARQ-IR (rv 0,2,3,1, soft-LLR combining) at the metric configuration, batched over HARQ processes.

    python tools/r4/bench_harq.py [--proc 64] [--rounds 8] [--snr 27]

Prints transmissions/s (one transmission = one PDSCH slot of one HARQ process through the whole chain incl. the
soft-combining rate recovery), the HARQ statistics and the size of the resident soft buffers."""
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
    ap.add_argument('--proc', type=int, default=64)
    ap.add_argument('--rounds', type=int, default=8)
    ap.add_argument('--snr', type=float, default=27.0)
    ap.add_argument('--decoder', default='f64', choices=['f32', 'f64'])
    a = ap.parse_args()
    import neoradium_amd as nr
    link = bench.build_link(nr, decoder=a.decoder)
    _, st = link.run_harq(a.proc, 1, a.snr, seed=1)          # warm-up round (also allocates the soft buffers)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats, st = link.run_harq(a.proc, a.rounds, a.snr, seed=1, state=st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "cfg5 HARQ-IR at the metric configuration (273 PRB, 64-QAM, 4x4 CDL-C, BG1 72 CB)",
                      "decoder": a.decoder, "harq_processes": a.proc, "rounds": a.rounds, "snr_db": a.snr,
                      "transmissions_per_s": a.proc * a.rounds / dt, "ms_per_round": 1e3 * dt / a.rounds,
                      "soft_buffer_MB": sum(c.numel() * c.element_size() for c in st['circ']) / 1e6,
                      "txBlocks": stats['txBlocks'].tolist(), "rxBlocks": stats['rxBlocks'].tolist(),
                      "numTimeouts": stats['numTimeouts'], "throughput_pct": stats['throughput'],
                      "bler_pct": stats['bler'], "meanTries": stats['meanTries']}))


if __name__ == '__main__':
    main()
