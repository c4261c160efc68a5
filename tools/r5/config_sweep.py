"""Round-5 tool (GPU): throughput of the float64 chain OUTSIDE the metric configuration -- what the decoders that serve other lifting
sizes, base graphs and row counts deliver (VERDICT r4 missing #5: "perf-wise a one-configuration build").

For each link: slots/s of the throughput mode (device RNG, fixed iteration count), the decoder's share of the step, and the decoder's
edge-visit rate (code blocks x iterations x edges of the rows that are needed x Zc / decoder time) beside the metric kernel's.

    python tools/r5/config_sweep.py [--steps 3]
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



def link_of(num_rbs, spacing, mod, layers, panel, rate, bg, num_iter, chan=('C', 300, 5)):
    nr.random.setSeed(123)
    car = nr.Carrier(numRbs=num_rbs, spacing=spacing)
    bwp = car.curBwp
    p = nr.PDSCH(bwp, numLayers=layers, nID=car.cellId, modulation=mod)
    p.setDMRS(configType=1, additionalPos=1)
    ch = nr.CdlChannel(bwp, chan[0], delaySpread=chan[1], carrierFreq=4e9, dopplerShift=chan[2],
                       txAntenna=nr.AntennaPanel(panel, polarization='x'), rxAntenna=nr.AntennaPanel(panel, polarization='x'))
    return nr.PdschLink(p, ch, rate, baseGraphNo=bg, numIter=num_iter, decoder="f64")


CASES = [
    # name, (numRbs, spacing, mod, layers, panel, rate, bg, numIter), batch, snr
    ("metric: 273 PRB 64QAM 4x4 R .65 BG1", (273, 30, '64QAM', 4, [1, 2], 666 / 1024, 1, 50), 256, 31.0),
    ("cfg2: 106 PRB 64QAM 2x2 R .65 BG1", (106, 15, '64QAM', 2, [1, 1], 666 / 1024, 1, 50), 512, 25.0),
    ("BLER notebook: 51 PRB 16QAM 2x2 R .48 BG1", (51, 30, '16QAM', 2, [1, 1], 490 / 1024, 1, 20), 1024, 12.0),
    ("106 PRB 16QAM 2x2 R .37 BG1", (106, 15, '16QAM', 2, [1, 1], 378 / 1024, 1, 50), 512, 10.0),
    ("273 PRB QPSK 4x4 R .30 BG1", (273, 30, 'QPSK', 4, [1, 2], 308 / 1024, 1, 50), 256, 6.0),
    ("52 PRB 64QAM 2x2 R .75 BG1", (52, 15, '64QAM', 2, [1, 1], 772 / 1024, 1, 50), 1024, 28.0),
    ("25 PRB QPSK 2x2 R .30 BG2", (25, 15, 'QPSK', 2, [1, 1], 308 / 1024, 2, 50), 2048, 4.0),
    ("106 PRB 16QAM 2x2 R .60 BG2", (106, 15, '16QAM', 2, [1, 1], 616 / 1024, 2, 50), 512, 14.0),
]


def row_starts(bg):
    """cumulative edge counts per row of the base graph, from the library's own tables via the decoder's row query"""
    if bg == 1:
        return bench.BG1_ROW_START
    # BG2 (ldpc.py:46-654): edges per row
    deg = [8, 10, 8, 10, 4, 6, 6, 6, 4, 5, 5, 5, 4, 5, 5, 4, 5, 5, 4, 4, 4, 4, 3, 4, 4, 3, 5, 3, 4, 3, 5, 3, 4, 4, 4, 4, 4, 3, 4, 4, 4, 4]
    return [0] + list(np.cumsum(deg))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3)
    a = ap.parse_args()
    out = []
    for name, args, B, snr in CASES:
        link = link_of(*args)
        cfg = link.cfg
        rows = link.cw[0]['rows'] or (46 if cfg.bg == 1 else 42)
        dt, c, dec_ms = bench.timed_steps(link, ops, B, a.steps, 1, snr, 0, None, torch.cuda.synchronize)
        c = c.cpu().numpy()
        rs = row_starts(cfg.bg)
        ev = B * cfg.C * link.numIter * rs[min(rows, len(rs) - 1)] * cfg.Zc      # the rows the code rate needs (an instantiation may run more)
        r = dict(config=name, tbs=link.tbs, code_blocks=cfg.C, Zc=cfg.Zc, bg=cfg.bg, rows=rows, num_iter=link.numIter, batch=B, snr_db=snr,
                 slots_per_s=B * a.steps / dt, ms_per_step=1e3 * dt / a.steps, decoder_ms=dec_ms, decoder_share=dec_ms / (1e3 * dt / a.steps),
                 decoder_edge_visits_per_s=ev / (dec_ms * 1e-3), bler=float(c[0]) / max(1, int(c[1])))
        out.append(r)
        print(json.dumps(r), flush=True)
        del link
        torch.cuda.empty_cache()
    base = out[0]['decoder_edge_visits_per_s']
    for r in out:
        r['decoder_rate_vs_metric_kernel'] = r['decoder_edge_visits_per_s'] / base
    print(json.dumps(dict(configs=out)), flush=True)


if __name__ == '__main__':
    main()
