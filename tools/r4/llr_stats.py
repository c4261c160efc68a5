"""Dev tool (GPU): LLR magnitudes per code block at the metric configuration, and a sample of rate-recovered LLRs + the
transmitted code blocks for the CPU prototype of the early-termination certificate (cert_proto.py of round 4, in the history).

    python tools/r4/llr_stats.py [--snr 31] [--slots 4] [--out gpurun_out/llr_sample.npz]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import neoradium_amd as nr                    # noqa: E402
from neoradium_amd import ops                 # noqa: E402
import bench                                  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--snr', type=float, nargs='+', default=[31.0])
    ap.add_argument('--slots', type=int, default=4)
    ap.add_argument('--keep', type=int, default=144)
    ap.add_argument('--out', default='gpurun_out/llr_sample.npz')
    a = ap.parse_args()
    link = bench.build_link(nr, decoder="f64", num_iter=50)
    cw = link.cw[0]
    cfg = cw['cfg']
    res = {}
    keep = {}
    for snr in a.snr:
        counters, det = link.run(1000, a.slots, snr, seed=5, details=True)
        d = det[0][1]
        llr = d['llr']                                    # (n, G) in the reference's order
        rr = ops.ldpc_rate_recover(llr, cfg, cw['nl'], cw['qm'])          # (n*C, N) float64
        ncol = 22 - 2 + 4 + (cw['rows'] - 4)
        x = rr[:, :ncol * cfg.Zc].abs()
        x = torch.where(x >= 1e9, torch.zeros_like(x), x)
        bmax = x.max(1).values.cpu().numpy()
        par = x[:, 20 * cfg.Zc:].max(1).values.cpu().numpy()            # core parity + extension columns
        ok = d['cb_ok'].reshape(-1).cpu().numpy().astype(bool)
        q = [50, 90, 99, 100]
        res[str(snr)] = dict(blocks=int(len(bmax)), crc_ok=int(ok.sum()), max_abs_llr_percentiles={str(p): float(np.percentile(bmax, p)) for p in q},
                             parity_ext_max_percentiles={str(p): float(np.percentile(par, p)) for p in q},
                             mean_abs=float(x.mean()), rows=int(cw['rows']))
        n = min(a.keep, rr.shape[0])
        keep[f'llr_{snr}'] = rr[:n, :ncol * cfg.Zc].cpu().numpy()
        keep[f'ok_{snr}'] = ok[:n]
        tb = d['tb']
        cbs = ops.ldpc_segment(tb, cfg)
        keep[f'cbs_{snr}'] = cbs[:n].cpu().numpy().astype(np.uint8)
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    np.savez_compressed(a.out, F=cfg.F, Zc=cfg.Zc, rows=cw['rows'], **keep)
    json.dump(res, open(a.out.replace('.npz', '.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
