"""Round-5 tool (GPU): the bench's certified-early-exit leg alone -- same links, slots, seed and timing protocol as bench.py's
`certified_early_exit` block -- so that a `rocprofv3 --kernel-trace` of it is a trace of exactly the timed steps.

    python tools/r5/cert_steps.py [--snr 31] [--steps 8] [--warmup 2] [--batch 256] [--stages 8 16] [--fixed]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import neoradium_amd as nr                    # noqa: E402
from neoradium_amd import ops                 # noqa: E402
import bench                                  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--snr', type=float, default=31.0)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--stages', type=int, nargs='+', default=[8, 16])
    ap.add_argument('--fixed', action='store_true', help='the fixed 50-iteration schedule instead')
    ap.add_argument('--persistent', action='store_true', help='the certified schedule as one persistent launch (round 6)')
    ap.add_argument('--slot0', type=int, default=0, help='first slot (bench.py: 0)')
    ap.add_argument('--flags', type=int, default=0, help='certificate flags (timing experiments: != 0 breaks the certificate)')
    a = ap.parse_args()
    kw = {} if a.fixed else {'certifiedExit': tuple(a.stages), 'certPersistent': a.persistent, 'certFlags': a.flags}
    link = bench.build_link(nr, decoder='f64', **kw)
    dt, c, _ = bench.timed_steps(link, ops, a.batch, a.steps, a.warmup, a.snr, a.slot0, None, torch.cuda.synchronize, timer_enabled=False)
    c = c.cpu().numpy()
    print(json.dumps({"schedule": "fixed" if a.fixed else ("certified, persistent" if a.persistent else "certified, staged"), "persistent_error": ops.persistent_error(), "snr_db": a.snr, "steps": a.steps, "batch": a.batch,
                      "ms_per_step": 1e3 * dt / a.steps, "slots_per_s": a.batch * a.steps / dt, "block_errors": int(c[0]), "blocks": int(c[1])}), flush=True)


if __name__ == '__main__':
    main()
