"""Round-5 evidence script (GPU box; lives under tests/ because it calls the oracle, which only tests may do): "BLER match vs CPU ref" of the headline metric on a sample larger than bench.py's parity block.

The metric configuration (273 PRB, 64-QAM, 4 x 4, BG1) over CDL through the HIP chain and through oracle/ (the NumPy float64
restatement of the reference) on IDENTICAL inputs -- transport blocks, noise draws and the precoder as data -- at several SNR points
across the waterfall.  Per point: both block error rates, the number of code-block CRC verdicts and hard bits that differ, the decoder
alone on the oracle's LLRs (must be bit-identical).  The oracle runs one single-threaded process per host core.

    python tests/bler_match.py [--snrs 29 31 33] [--slots 64] [--procs 16] [--out gpurun_out/r5/r5_bler_match.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import neoradium_amd as nr                    # noqa: E402
from neoradium_amd import ops                 # noqa: E402
from neoradium_amd._dev import D              # noqa: E402
import bench                                  # noqa: E402
from oracle import link as olink              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--snrs', type=float, nargs='+', default=[29.0, 31.0, 33.0])
    ap.add_argument('--slots', type=int, default=64)
    ap.add_argument('--procs', type=int, default=max(1, min(16, os.cpu_count() or 1)))
    ap.add_argument('--chunk', type=int, default=16)
    ap.add_argument('--out', default='gpurun_out/r5/r5_bler_match.json')
    a = ap.parse_args()
    link = bench.build_link(nr, decoder='f64')
    cw = link.cw[0]
    st = olink.static_from_link(link, slots=range(a.slots))
    points = []
    t_all = time.time()
    for pi, snr in enumerate(a.snrs):
        rng = np.random.default_rng(7000 + pi)
        acc = dict(blocks=0, gpu_fail=0, cpu_fail=0, crc_diff=0, bit_diff=0, bits=0, dec_crc_diff=0, dec_bit_diff=0, llr_err=0.0, fused_equal=True)
        t0 = time.time()
        for s0 in range(0, a.slots, a.chunk):
            n = min(a.chunk, a.slots - s0)
            tb = rng.integers(0, 2, (n, link.tbs)).astype(np.uint8)
            z = rng.standard_normal((n, link.nr, link.slot_len[0] + link.max_delay, 2))
            zc = z[..., 0] + 1j * z[..., 1]
            del z
            _, det = link.run(s0, n, snr, tb_bits=torch.from_numpy(tb), noise=D(zc), details=True)
            d = det[0][1]
            _, dv = link.run(s0, n, snr, tb_bits=torch.from_numpy(tb), noise=D(zc), details="verdicts")
            acc['fused_equal'] &= bool(torch.equal(torch.cat([x['cb_ok'].reshape(-1) for _, x in dv]), d['cb_ok'].reshape(-1)))
            torch.cuda.synchronize()
            F = d['F'].cpu().numpy()
            got_llr = d['llr'].cpu().numpy().astype(np.float64)
            got_ok = d['cb_ok'].cpu().numpy().astype(bool)
            got_tb = d['tb_out'].cpu().numpy()
            del det, d, dv
            jobs = [(st, s0 + s, snr, tb[s].astype(np.int8), zc[s], F[s]) for s in range(n)]
            refs = olink.run_slots_parallel(jobs, min(a.procs, n))
            for s, ref in enumerate(refs):
                nb = len(ref['tb_out'])
                want = ref['tb_out'].astype(np.uint8)
                acc['blocks'] += len(ref['crc'])
                acc['bits'] += nb
                acc['cpu_fail'] += int((~ref['crc']).sum())
                acc['gpu_fail'] += int((~got_ok[s]).sum())
                acc['crc_diff'] += int((got_ok[s] != ref['crc']).sum())
                acc['bit_diff'] += int((got_tb[s][:nb] != want).sum())
                acc['llr_err'] = max(acc['llr_err'], float(np.abs(got_llr[s] - ref['llr']).max() / np.abs(ref['llr']).max()))
                # the decoder alone on the oracle's LLRs: identical input, so rate recovery + decode + CRC are bit-identical
                rr = ops.ldpc_rate_recover(D(ref['llr'][None]), cw['cfg'], cw['nl'], cw['qm'])
                dec = ops.ldpc_decode(rr, cw['cfg'], link.numIter, rows=cw['rows'])
                tb_o, cb_ok, _ = ops.ldpc_crc_merge(dec, cw['cfg'], want_tb_crc=False)
                acc['dec_crc_diff'] += int((cb_ok[0].cpu().numpy().astype(bool) != ref['crc']).sum())
                acc['dec_bit_diff'] += int((tb_o[0].cpu().numpy()[:nb] != want).sum())
            print(f"snr {snr}: slots {s0 + n}/{a.slots}  {time.time() - t0:.0f} s", flush=True)
        b = acc['blocks']
        p = acc['cpu_fail'] / b
        points.append(dict(snr_db=snr, slots=a.slots, code_blocks=b, bler_hip=acc['gpu_fail'] / b, bler_oracle=p,
                           block_errors_hip=acc['gpu_fail'], block_errors_oracle=acc['cpu_fail'],
                           binomial_sigma_of_the_oracle_bler=float(np.sqrt(max(p * (1 - p), 1e-12) / b)),
                           chain=dict(crc_verdicts_differing=acc['crc_diff'], hard_bits_differing=acc['bit_diff'], hard_bits=acc['bits'],
                                      llr_max_rel_err=acc['llr_err']),
                           decoder_on_oracle_llrs=dict(crc_verdicts_differing=acc['dec_crc_diff'], hard_bits_differing=acc['dec_bit_diff'],
                                                       bit_exact=acc['dec_crc_diff'] == 0 and acc['dec_bit_diff'] == 0),
                           fused_entry_verdicts_equal_separate_stages=acc['fused_equal'], seconds=round(time.time() - t0, 1)))
        print(json.dumps(points[-1]), flush=True)
    out = dict(workload=bench.METRIC if hasattr(bench, 'METRIC') else "metric", numIter=link.numIter, host_processes=a.procs,
               inputs="transport blocks and noise draws from numpy.random.default_rng(7000 + point), the HIP chain's SVD precoder handed to the "
                      "oracle as data (SURVEY 8c: SVD-derived precoders are inputs)",
               seconds=round(time.time() - t_all, 1), points=points)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(out, open(a.out, 'w'), indent=1)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
