"""Round-5 experiment (GPU): the bench's fixed-schedule steps issued alternately on two HIP streams (two PdschLink objects, so that no
work buffer is shared): step k + 1's HBM-bound front end can share the chip with step k's VALU-bound decoder.

    python tools/r5/two_stream.py [--steps 20] [--streams 2]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import neoradium_amd as nr                    # noqa: E402
import bench                                  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--streams', type=int, default=2)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--snr', type=float, default=31.0)
    a = ap.parse_args()
    n = a.streams
    links = [bench.build_link(nr, decoder='f64') for _ in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    dev = links[0].dev
    B = a.batch
    cnt = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(n)]
    for w in range(a.warmup * n):
        with torch.cuda.stream(streams[w % n]):
            links[w % n].run(w * B, B, a.snr, seed=123)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        with torch.cuda.stream(streams[k % n]):
            links[k % n].run((a.warmup * n + k) * B, B, a.snr, seed=123, counters=cnt[k % n])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = sum(x.cpu() for x in cnt).numpy()
    print(json.dumps(dict(streams=n, steps=a.steps, batch=B, ms_per_step=1e3 * dt / a.steps, slots_per_s=B * a.steps / dt,
                          block_errors=int(c[0]), blocks=int(c[1]))), flush=True)


if __name__ == '__main__':
    main()
