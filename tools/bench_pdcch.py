#!/usr/bin/env python3
"""BASELINE cfg4 with a caller: PDCCH blind decoding of whole monitoring occasions (neoradium_amd/pdcch.py).

    python tools/bench_pdcch.py [--occasions 2048] [--cces 16] [--A 64]

One occasion = a 16-CCE CORESET carrying one AL-8 and one AL-4 DCI for this UE + noise; all 31 aligned candidates
(AL 1/2/4/8/16) are demapped, descrambled, rate-recovered and SCL-decoded (list 8) with the RNTI-masked CRC test.
Prints candidates/s and occasions/s."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--occasions', type=int, default=2048)
    ap.add_argument('--cces', type=int, default=16)
    ap.add_argument('--A', type=int, default=64)
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    import torch
    import neoradium_amd as nr
    dev = torch.device('cuda:0')
    pd = nr.PDCCH(a.cces, nID=11, rnti=0x4321)
    rng = np.random.default_rng(0)
    n = a.occasions
    grid = torch.zeros((n, a.cces * 54), dtype=torch.complex128, device=dev)
    p8 = rng.integers(0, 2, (n, a.A)).astype(np.uint8)
    p4 = rng.integers(0, 2, (n, a.A)).astype(np.uint8)
    grid[:, 0:8 * 54] = pd.encode(p8, 8)
    grid[:, 8 * 54:12 * 54] = pd.encode(p4, 4)
    sigma = 0.7
    g = torch.Generator(device=dev).manual_seed(3)
    noisy = grid + sigma / np.sqrt(2) * torch.complex(torch.randn(grid.shape, dtype=torch.float64, device=dev, generator=g),
                                                    torch.randn(grid.shape, dtype=torch.float64, device=dev, generator=g))
    found, bits, cands = pd.blindDecode(noisy, sigma ** 2, a.A)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        found, bits, cands = pd.blindDecode(noisy, sigma ** 2, a.A)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    i8, i4 = cands.index((8, 0)), cands.index((4, 8))
    ok8 = (found[:, i8] & (bits[:, i8] == torch.from_numpy(p8).to(dev)).all(1)).float().mean().item()
    ok4 = (found[:, i4] & (bits[:, i4] == torch.from_numpy(p4).to(dev)).all(1)).float().mean().item()
    # the AL-16 candidate at CCE 0 starts with the AL-8 DCI's 864 bits, same mother code (N = 512 repeated), same scrambling
    # prefix: it decodes that DCI too -- the well-known AL-8 / AL-16 ambiguity of NR, not a false alarm
    i16 = cands.index((16, 0)) if (16, 0) in cands else None
    alias = int((found[:, i16] & (bits[:, i16] == torch.from_numpy(p8).to(dev)).all(1)).sum().item()) if i16 is not None else 0
    false_alarms = int(found.sum().item()) - int(found[:, i8].sum().item()) - int(found[:, i4].sum().item()) - alias
    print(json.dumps({"workload": f"PDCCH blind decoding, {a.cces}-CCE CORESET, A={a.A}, {len(cands)} candidates per occasion, SCL list 8",
                      "occasions": n, "candidates_per_s": n * len(cands) / dt, "occasions_per_s": n / dt,
                      "detect_rate_AL8": ok8, "detect_rate_AL4": ok4, "al16_aliases_of_the_al8_dci": alias,
                      "false_alarms": false_alarms,
                      "candidates_tested": n * len(cands)}))


if __name__ == '__main__':
    main()
