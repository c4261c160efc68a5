"""Round-5 tool (GPU): the channel filter alone at the metric configuration's shapes (256 slots, 4 layers -> 4 Rx, CDL-C 300 ns:
24 paths, hist 334, 15 gain sets, 61 440 + delay samples), overlap-save against the path form.  NRX_LIB picks a variant library.

    python tools/r5/bench_filter.py [--batch 256] [--reps 10] [--paths]
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
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--paths', action='store_true', help='also time the path-form kernel')
    a = ap.parse_args()
    link = bench.build_link(nr, decoder='f64')
    dev = link.dev
    n = a.batch
    g = torch.Generator(device=dev).manual_seed(1)
    ns = link.slot_len[0] + link.max_delay
    x = torch.randn((n, link.nl, ns, 2), dtype=torch.float64, device=dev, generator=g)
    x = torch.view_as_complex(x).contiguous()
    P = link.taps.shape[0]
    gains = torch.view_as_complex(torch.randn((n, link.L + 1, link.nr, link.nl, P, 2), dtype=torch.float64, device=dev, generator=g)).contiguous()
    lens = [int(v) for v in link.sym_lens[0]]
    snr = torch.full((n,), 1000.0, dtype=torch.float64, device=dev)
    power = (link.nfft, snr, link.nfft / (12.0 * link.bwp.numRbs), float(link.nfft))
    out = {"lib": os.environ.get('NRX_LIB', 'in-tree'), "batch": n, "hist": link.td_hist, "paths": int(P), "ns": int(ns)}

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            r = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps, r

    ms, r_os = timed(lambda: ops.apply_td_os(x, gains, link.td_spec, link.td_hist, lens, power=power))
    out["os_ms"] = ms
    if a.paths:
        ms, r_pf = timed(lambda: ops.apply_td_paths(x, gains, link.taps, link.tap_off, lens, hist=link.td_hist, power=power))
        out["paths_ms"] = ms
        out["max_abs_diff_rel"] = float((r_os[0] - r_pf[0]).abs().max() / r_pf[0].abs().max())
        out["sigma_rel_diff"] = float(((r_os[1] - r_pf[1]).abs() / r_pf[1]).max())
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
