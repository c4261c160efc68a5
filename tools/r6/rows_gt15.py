#!/usr/bin/env python3
"""Round-6 tool (GPU): the two figures of VERDICT r5 #4 -- `all_rows` (the metric steps with all 46 rows of the base graph, the 46-row hybrid) and
cfg5 (one HARQ round of 64 processes per step) -- with bench.py's own protocol, for A/B runs of the hybrid decoder.

    python tools/r6/rows_gt15.py
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import neoradium_amd as nr                    # noqa: E402
from neoradium_amd import ops                 # noqa: E402
import bench                                  # noqa: E402

al = bench.build_link(nr, decoder='f64', skipPuncturedRows=False)
dt, c, _ = bench.timed_steps(al, ops, 256, 3, 1, 31.0, 0, None, torch.cuda.synchronize, timer_enabled=False)
out = {"all_rows": {"value": 256 * 3 / dt, "unit": "slots/s", "counters": c.cpu().numpy().tolist()}}
del al
l5 = bench.build_link(nr, decoder='f64')
_, st = l5.run_harq(64, 2, 27.0, seed=123)
torch.cuda.synchronize()
t0 = time.perf_counter()
stats, st = l5.run_harq(64, 8, 27.0, seed=123, state=st)
torch.cuda.synchronize()
d = time.perf_counter() - t0
out["cfg5"] = {"value": 64 * 8 / d, "unit": "transmissions/s", "ms_per_round": 1e3 * d / 8, "txBlocks": stats['txBlocks'].tolist(), "rxBlocks": stats['rxBlocks'].tolist()}
print(json.dumps(out))
