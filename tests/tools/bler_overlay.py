#!/usr/bin/env python3
"""BLER-vs-SNR overlay: the GPU engine against the CPU oracle (NumPy float64 restatement of the reference) on the SAME
transport blocks and noise draws, slot by slot.  Writes profiles/r1_bler_overlay.json.

    python tests/tools/bler_overlay.py [--slots 24] [--out profiles/r1_bler_overlay.json]

Configurations: BASELINE cfg1 (25 PRB, QPSK, BG2, SISO TDL-A 30 ns), a 2x2 CDL-C 16-QAM case and (--configs metric) the
bench configuration itself (273 PRB, 7 s of CPU oracle per slot: use --slots 6); all time-domain
channel + DMRS-LS + MMSE, 20 iterations.  Per SNR point: block errors of both paths and whether every per-slot CRC
vector was identical, for the engine's f32 (throughput) and f64 (bit-exact) decoders."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def build(nr, which, decoder, **kw):
    nr.random.setSeed(123)
    if which == 'cfg1':
        car = nr.Carrier(numRbs=25, spacing=15)
        bwp = car.curBwp
        p = nr.PDSCH(bwp, numLayers=1, nID=car.cellId, modulation='QPSK')
        p.setDMRS(configType=1, additionalPos=1)
        ch = nr.TdlChannel(bwp, 'A', delaySpread=30, carrierFreq=4e9, dopplerShift=5)
        return nr.PdschLink(p, ch, 0.35, baseGraphNo=2, numIter=20, freqDomain=False, chanEst="LS", decoder=decoder, **kw), \
            [0.0, 0.4, 0.8, 1.2, 1.6, 2.0]
    if which == 'metric':        # the bench configuration: 273 PRB, 64-QAM, 4 layers, 4x4 CDL-C, BG1, 50 iterations
        import bench
        return bench.build_link(nr, decoder=decoder, **kw), [29.0, 32.0, 35.0]
    car = nr.Carrier(numRbs=51, spacing=30)
    bwp = car.curBwp
    p = nr.PDSCH(bwp, numLayers=2, nID=car.cellId, modulation='16QAM')
    p.setDMRS(configType=1, additionalPos=1)
    ch = nr.CdlChannel(bwp, 'C', delaySpread=300, carrierFreq=4e9, dopplerShift=5,
                       txAntenna=nr.AntennaPanel([1, 1], polarization="x"),
                       rxAntenna=nr.AntennaPanel([1, 1], polarization="x"))
    return nr.PdschLink(p, ch, 490 / 1024, baseGraphNo=1, numIter=20, freqDomain=False, chanEst="LS", decoder=decoder, **kw), \
        [9.5, 10.0, 10.5, 11.0, 11.5]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--slots', type=int, default=24)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r1_bler_overlay.json'))
    ap.add_argument('--configs', default='cfg1,cdl_2x2_16qam', help="comma list of cfg1, cdl_2x2_16qam, metric")
    ap.add_argument('--procs', type=int, default=1, help="CPU oracle: one single-threaded process per core (oracle.link.run_slots_parallel)")
    ap.add_argument('--snrs', default='', help="comma list overriding the configuration's SNR points")
    a = ap.parse_args()
    import neoradium_amd as nr
    from neoradium_amd._dev import D
    from oracle import link as olink
    res = {}
    for which in a.configs.split(','):
        link, snrs = build(nr, which, 'f32')
        link64, _ = build(nr, which, 'f64')
        linkw, _ = build(nr, which, 'f32', waveform='f32')       # float32 decoder AND complex64 waveform chain (opt-in fast mode)
        # round 4: the float64 chain with the CERTIFIED early exit (metric configuration: the fused entry's instantiations)
        linkc = build(nr, which, 'f64', certifiedExit=(8, 16))[0] if which == 'metric' else None
        if a.snrs:
            snrs = [float(v) for v in a.snrs.split(',')]
        st = olink.static_from_link(link, slots=range(0, 20 * len(snrs) + a.slots))
        rows = []
        t_cpu = 0.0
        for si, snr in enumerate(snrs):
            rng = np.random.default_rng(1000 + si)
            n = a.slots
            tb = rng.integers(0, 2, (n, link.tbs)).astype(np.uint8)
            z = rng.standard_normal((n, link.nr, link.slot_len[0] + link.max_delay, 2))
            zc = z[..., 0] + 1j * z[..., 1]
            slots0 = 20 * si                                      # 20 slots per frame at 30 kHz, 10 at 15 kHz: same geometry
            _, det = link.run(slots0, n, snr, tb_bits=torch.from_numpy(tb), noise=D(zc), details=True)
            d = det[0][1]
            gpu_ok = d['cb_ok'].cpu().numpy().astype(bool)
            _, det64 = link64.run(slots0, n, snr, tb_bits=torch.from_numpy(tb), noise=D(zc), details=True)
            gpu64_ok = det64[0][1]['cb_ok'].cpu().numpy().astype(bool)
            # the float64 THROUGHPUT path (fused rate recovery + decode + CRC/merge where it applies) on the same slots
            _, dv = link64.run(slots0, n, snr, tb_bits=torch.from_numpy(tb), noise=D(zc), details="verdicts")
            fused_ok = torch.cat([x['cb_ok'] for _, x in dv]).cpu().numpy().astype(bool).reshape(gpu64_ok.shape)
            _, dw = linkw.run(slots0, n, snr, tb_bits=torch.from_numpy(tb), noise=D(zc), details="verdicts")
            wave_ok = torch.cat([x['cb_ok'] for _, x in dw]).cpu().numpy().astype(bool).reshape(gpu64_ok.shape)
            cert = {}
            if linkc is not None:
                _, dc = linkc.run(slots0, n, snr, tb_bits=torch.from_numpy(tb), noise=D(zc), details="verdicts")
                cert_ok = torch.cat([x['cb_ok'] for _, x in dc]).cpu().numpy().astype(bool).reshape(gpu64_ok.shape)
                same_bits = all(torch.equal(x['tb_out'], y['tb_out']) for (_, x), (_, y) in zip(dc, dv))
                cert = dict(gpu_f64_certified_block_errors=int((~cert_ok).sum()),
                            certified_crc_vectors_differ_from_f64_throughput_path_in=int((cert_ok != fused_ok).sum()),
                            certified_bits_identical_to_f64_throughput_path=bool(same_bits),
                            blocks_stopped_early=int((linkc.last_exit_iter > 0).sum()))
            F = d['F'].cpu().numpy()
            t0 = time.time()
            jobs = [(st, slots0 + i, snr, tb[i].astype(np.int8), zc[i], F[i]) for i in range(n)]
            if a.procs > 1:
                cpu_ok = [r['crc'] for r in olink.run_slots_parallel(jobs, a.procs)]
            else:
                cpu_ok = [olink.run_slot(*j)['crc'] for j in jobs]
            t_cpu += time.time() - t0
            cpu_ok = np.array(cpu_ok)
            rows.append(dict(snr_db=snr, blocks=int(cpu_ok.size), cpu_block_errors=int((~cpu_ok).sum()),
                             gpu_f32_block_errors=int((~gpu_ok).sum()), gpu_f64_block_errors=int((~gpu64_ok).sum()),
                             gpu_f32_waveform_block_errors=int((~wave_ok).sum()),
                             f32_waveform_crc_vectors_differ_in=int((cpu_ok != wave_ok).sum()),
                             f32_crc_vectors_differ_in=int((cpu_ok != gpu_ok).sum()),
                             f64_crc_vectors_differ_in=int((cpu_ok != gpu64_ok).sum()),
                             f64_throughput_path_differs_from_f64_in=int((fused_ok != gpu64_ok).sum()), **cert))
            print(which, rows[-1], flush=True)
        res[which] = dict(tbs=link.tbs, code_blocks=link.cfg.C, slots_per_point=a.slots, points=rows,
                          cpu_oracle_s_per_slot=t_cpu / (len(snrs) * a.slots))
    json.dump(res, open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
