#!/usr/bin/env python3
"""Where does the HIP chain first leave the CPU oracle's bits?  (VERDICT r5 weak #2 / next #2c)

bench.py's `parity_vs_cpu_oracle.chain` reports that a handful of CRC verdicts differ END TO END although every stage is bit-exact
on identical inputs.  This script runs the bench's parity slots (metric configuration, identical transport blocks, noise draws and
precoders) through PdschLink(details=True) and through oracle.link.run_slot(keep=...) and prints, stage by stage in the order of the
data flow, how many elements differ in ANY bit and the largest difference in units of the stage's largest magnitude x 2^-52 -- so
"FFT / summation order" becomes a table.  Run on the GPU box:

    python tools/r6/stage_diff.py [--slots 16] [--snr 31] > profiles/r6_stage_diff.json

Two passes: the default engine (wideband precoder folded into the channel filter's gains: the Tx waveform is per LAYER and has no
counterpart in the oracle) and NRX_SEPARATE_PRECODER=1 (the reference's order: precode, modulate, filter), where the Tx waveform is
compared as well.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def stage(name, got, ref, what):
    got, ref = np.asarray(got), np.asarray(ref)
    if got.shape != ref.shape:
        return dict(stage=name, what=what, error=f"shape {got.shape} vs {ref.shape}")
    if got.dtype.kind in 'iub' or ref.dtype.kind in 'iub':
        nd = int((got.astype(np.int64) != ref.astype(np.int64)).sum())
        return dict(stage=name, what=what, elements=int(got.size), differing=nd, max_err_in_eps_of_max=0.0 if nd == 0 else None)
    g = np.ascontiguousarray(got.astype(np.complex128 if np.iscomplexobj(got) or np.iscomplexobj(ref) else np.float64))
    r = np.ascontiguousarray(ref.astype(g.dtype))
    if np.iscomplexobj(g):
        gv, rv = g.view(np.float64), r.view(np.float64)
    else:
        gv, rv = g, r
    differ = int((gv.view(np.int64) != rv.view(np.int64)).sum())
    scale = float(np.abs(r).max()) or 1.0
    err = float(np.abs(g - r).max())
    # per-element error in units of the ELEMENT's own ulp (median / max over the differing ones)
    with np.errstate(divide='ignore', invalid='ignore'):
        ulp = np.spacing(np.maximum(np.abs(rv), np.finfo(np.float64).tiny))
        eu = np.abs(gv - rv) / ulp
    eu = eu[np.isfinite(eu) & (gv != rv)]
    return dict(stage=name, what=what, elements=int(gv.size), differing=differ, frac_differing=differ / gv.size,
                max_abs_err=err, max_err_in_eps_of_max=err / (scale * 2.0 ** -52),
                own_ulp_median=float(np.median(eu)) if eu.size else 0.0, own_ulp_max=float(eu.max()) if eu.size else 0.0)


def run(args, sep):
    if sep:
        os.environ['NRX_SEPARATE_PRECODER'] = '1'
    else:
        os.environ.pop('NRX_SEPARATE_PRECODER', None)
    import torch
    import neoradium_amd as nr
    from neoradium_amd._dev import D
    from oracle import link as olink
    import bench
    link = bench.build_link(nr)
    n = args.slots
    st = olink.static_from_link(link, slots=range(n))
    rng = np.random.default_rng(2025)                      # bench.cpu_baseline's draws
    tb = rng.integers(0, 2, (n, link.tbs)).astype(np.uint8)
    z = rng.standard_normal((n, link.nr, link.slot_len[0] + link.max_delay, 2))
    zc = z[..., 0] + 1j * z[..., 1]
    _, det = link.run(0, n, args.snr, tb_bits=torch.from_numpy(tb), noise=D(zc), details=True)
    d = det[0][1]
    torch.cuda.synchronize()
    F = d['F'].cpu().numpy()
    rows, first = [], {}
    t0 = time.time()
    acc = {}
    for s in range(n):
        keep = {}
        ref = olink.run_slot(st, s, args.snr, tb[s].astype(np.int8), zc[s], F=F[s], keep=keep)
        L = link.slot_len[0] + link.max_delay
        stages = [('rate_matched_bits', d['bits'][s].cpu().numpy(), keep['bits'], 'ldpc.py:981-1159 segment / encode / rate match'),
                  ('tx_grid', d['grid'][s].cpu().numpy(), keep['pgrid'] if sep else keep['grid'],
                   'pdsch.py:855-932 scramble / modulate / map' + (' + precode (grid.py:430)' if sep else ''))]
        if sep:
            stages.append(('tx_waveform', d['tx'][s].cpu().numpy()[..., :L], keep['tx'][..., :L], 'grid.py:521-582 IFFT + CP + windowing'))
        stages += [('rx_waveform', d['ry'][s].cpu().numpy()[..., :L], keep['ry'][..., :L], 'channelmodel.py:403-448 channel filter (overlap-save FFT blocks here, 333-tap direct sums in NumPy)'),
                   ('noise_sigma', np.float64([d['sigma'][s].item()]), np.float64([keep['sigma']]), 'waveform.py:119-142 noise level'),
                   ('rx_grid', d['rxg'][s].cpu().numpy(), keep['rxg'], 'waveform.py:473-527 noise add + FFT (radix-16 passes here, pocketfft in NumPy)'),
                   ('channel_estimate', d['hest'][s].cpu().numpy(), keep['hest'], 'grid.py:740-837 DMRS LS + interpolation'),
                   ('equalised', d['eq'][s].cpu().numpy(), keep['eq'], 'grid.py:626-694 MMSE (Cholesky per RE here, np.linalg.inv in NumPy)'),
                   ('llr', d['llr'][s].cpu().numpy(), keep['llr'], 'modulation.py max-log demap'),
                   ('crc_verdicts', d['cb_ok'][s].cpu().numpy().astype(np.uint8), ref['crc'].astype(np.uint8), 'ldpc.py:1495-1619 decode + CRC'),
                   ('hard_bits', d['tb_out'][s].cpu().numpy()[:len(ref['tb_out'])], ref['tb_out'].astype(np.uint8), 'decoded transport block')]
        for name, got, want, what in stages:
            r = stage(name, got, want, what)
            a = acc.setdefault(name, dict(stage=name, what=what, slots=0, elements=0, differing=0, max_err_in_eps_of_max=0.0,
                                          own_ulp_median=[], own_ulp_max=0.0))
            if 'error' in r:
                a['error'] = r['error']
                continue
            a['slots'] += 1
            a['elements'] += r['elements']
            a['differing'] += r['differing']
            if r.get('max_err_in_eps_of_max') is not None:
                a['max_err_in_eps_of_max'] = max(a['max_err_in_eps_of_max'], r['max_err_in_eps_of_max'])
            if 'own_ulp_median' in r:
                a['own_ulp_median'].append(r['own_ulp_median'])
                a['own_ulp_max'] = max(a['own_ulp_max'], r['own_ulp_max'])
            if r['differing'] and s not in first:
                first[s] = name
        print(f"[stage_diff] {'separate' if sep else 'folded'} precoder: slot {s} done ({time.time() - t0:.0f} s), first differing stage: {first.get(s)}",
              file=sys.stderr, flush=True)
    for a in acc.values():
        a['frac_differing'] = a['differing'] / max(a['elements'], 1)
        a['own_ulp_median'] = float(np.median(a['own_ulp_median'])) if a['own_ulp_median'] else 0.0
        rows.append(a)
    hist = {}
    for s in range(n):
        hist[first.get(s, 'none')] = hist.get(first.get(s, 'none'), 0) + 1
    return dict(precoder='separate (reference order: NRX_SEPARATE_PRECODER=1)' if sep else 'folded into the channel filter gains (default)',
                slots=n, snr_db=args.snr, first_differing_stage_per_slot=hist, stages=rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--slots', type=int, default=16)
    ap.add_argument('--snr', type=float, default=31.0)
    ap.add_argument('--only', choices=['folded', 'separate'])
    args = ap.parse_args()
    out = dict(tool='tools/r6/stage_diff.py', unit="differing = elements (real and imaginary parts counted separately) whose float64 bit patterns differ; "
               "max_err_in_eps_of_max = max |got - ref| / (max |ref| x 2^-52); own_ulp = |got - ref| in units of the element's own ulp",
               runs=[run(args, sep) for sep in ((False, True) if not args.only else ((args.only == 'separate'),))])
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
