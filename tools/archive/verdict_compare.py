"""float32 fast mode against the float64 chain, verdict by verdict: the same slots (same device-generator
keys => same transport blocks, channel and noise) through PdschLink(decoder='f32') and PdschLink(decoder='f64'), per-code-block
CRC verdicts compared over >= 1e5 code blocks across the waterfall.  Writes profiles/r2_f32_vs_f64_verdicts.json.

    python tools/archive/verdict_compare.py [metric|cfg2] [blocks_per_point]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import neoradium_amd as nr
import bench


def cfg2_link(decoder):
    nr.random.setSeed(123)
    car = nr.Carrier(numRbs=106, spacing=15)
    p = nr.PDSCH(car.curBwp, numLayers=2, nID=car.cellId, modulation='64QAM')
    p.setDMRS(configType=1, additionalPos=1)
    ch = nr.CdlChannel(car.curBwp, 'C', delaySpread=300, carrierFreq=4e9, dopplerShift=5,
                       txAntenna=nr.AntennaPanel([1, 1], polarization="x"), rxAntenna=nr.AntennaPanel([1, 1], polarization="x"))
    return nr.PdschLink(p, ch, 666 / 1024, baseGraphNo=1, numIter=50, freqDomain=False, chanEst="LS", decoder=decoder)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'metric'
    per_point = int(sys.argv[2]) if len(sys.argv) > 2 else 21000
    if which == 'metric':
        l32, l64 = bench.build_link(nr, decoder='f32'), bench.build_link(nr, decoder='f64')
        snrs, batch = [28.0, 29.0, 30.0, 31.0, 32.0, 33.0, 35.0], 128
    else:
        l32, l64 = cfg2_link('f32'), cfg2_link('f64')
        snrs, batch = [19.0, 21.0, 23.0, 25.0, 27.0, 30.0], 512
    C = l64.cfg.C
    n_slots = -(-per_point // C)
    rows, tot_blocks, tot_diff = [], 0, 0
    for i, snr in enumerate(snrs):
        e32 = e64 = diff = blocks = 0
        only32 = only64 = 0
        done = 0
        while done < n_slots:
            nb = min(batch, n_slots - done)
            s0 = i * 1000000 + done
            a = l32.run(s0, nb, snr, seed=77, details="verdicts")[1]
            b = l64.run(s0, nb, snr, seed=77, details="verdicts")[1]
            va = torch.cat([d['cb_ok'].reshape(-1) for _, d in a]).cpu().numpy().astype(bool)
            vb = torch.cat([d['cb_ok'].reshape(-1) for _, d in b]).cpu().numpy().astype(bool)
            e32 += int((~va).sum()); e64 += int((~vb).sum()); diff += int((va != vb).sum()); blocks += va.size
            only32 += int((~va & vb).sum()); only64 += int((va & ~vb).sum())
            done += nb
        rows.append(dict(snr_db=snr, blocks=blocks, block_errors_f32=e32, block_errors_f64=e64, verdicts_differing=diff,
                         fails_only_in_f32=only32, fails_only_in_f64=only64))
        tot_blocks += blocks; tot_diff += diff
        print(rows[-1], flush=True)
    out = dict(config=which, workload=bench.WORKLOAD if which == 'metric' else "106 PRB @15 kHz, 64-QAM, 2 layers, 2x2 CDL-C 300 ns, BG1 R=666/1024 (16 CB, Zc 384), TD channel, LS + MMSE, 50 it",
               seed=77, points=rows, total_blocks=tot_blocks, total_verdicts_differing=tot_diff,
               disagreement_rate=tot_diff / max(tot_blocks, 1),
               note="same slots through both chains; float32 = float32 LLRs + ldpc_dec_fast_kernel, float64 = float64 LLRs + on-chip float64 decoder")
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', f'r2_f32_vs_f64_verdicts_{which}.json'), 'w'), indent=1)
    print(json.dumps({k: out[k] for k in ('config', 'total_blocks', 'total_verdicts_differing', 'disagreement_rate')}))


if __name__ == '__main__':
    main()
