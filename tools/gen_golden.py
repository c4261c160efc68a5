#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ by running the REFERENCE (read-only /root/reference)
in the development container.  The reference never travels to the GPU box; these small data files do.

    python tools/gen_golden.py

Outputs (all data: inputs + the reference's outputs, nothing of its source):
  tests/golden/matlab_ldpc/*.mat, matlab_polar/*.mat   MATLAB 5G-Toolbox vectors the reference's own notebooks assert on
  tests/golden/coding.npz        CRC / segmentation / encode / rate-match / rate-recover / decode at several configs
  tests/golden/coding_lbrm.npz   the same with a limited buffer (nRef > 0): rate-matched bits, (C, Ncb) rate-recovered LLRs, HARQ buffers
  tests/golden/phy.npz           gold sequence, constellations, LLRs, equaliser, OFDM, FIR bank, LS estimate
  tests/golden/host.npz          Carrier / PDSCH / DMRS index + pilot tables, TBS values, SnrScheduler walks
  tests/golden/channels.npz      CDL / TDL per-slot gains + coefficient matrices for seeded channels
  tests/golden/chest.npz         DMRS LS estimates (linear / polar subcarrier interpolation) + noise estimates
  tests/golden/prg.npz           per-PRG precoders (group lists + matrices) and the precoded grid
  tests/golden/polar.npz         polar DCI/PBCH/UCI chains: bits, LLRs, SCL candidate lists and path costs
  tests/golden/csifeedback_multipanel.json  what the reference's Type-I multi-panel codebooks do (raise / wrong shapes)
  tests/golden/snr.npz           SnrCalculations.ipynb noise-level anchors, useRxPower=False noise, seed-chain anchors
  tests/golden/bler_notebook.npz PDSCH-BLER.ipynb table (Perfect CSI): per-block CRC verdicts at 5.8 / 5.6 / 5.4 dB, per-slot LAPACK precoders
  tests/golden/e2e_*.npz         whole PDSCH slots (inputs: seed-derived bits/noise; outputs: LLRs, bits, CRC)
"""
import os
import shutil
import sys

import numpy as np

sys.path.insert(0, '/root/reference')
os.chdir('/tmp')
import neoradium as nr                                   # noqa: E402
from neoradium.utils import goldSequence                 # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
PLAY = '/root/reference/Playground/CompareWithMatlab'


def copy_matlab():
    for sub, dst in (('LDPC', 'matlab_ldpc'), ('Polar', 'matlab_polar')):
        d = os.path.join(GOLD, dst)
        os.makedirs(d, exist_ok=True)
        src = os.path.join(PLAY, sub, 'MatlabFiles')
        for f in sorted(os.listdir(src)):
            if f.endswith('.mat'):
                shutil.copyfile(os.path.join(src, f), os.path.join(d, f))
    # PDSCH-waveform.ipynb (52 PRB @30 kHz, startRb 1, 2 layers, VRB interleaving bundle 2, 16-QAM): the four small vectors
    # as they are; of the 4 x 15360 waveform the complete first symbol (long CP, windowed head, wrapped tail of the last
    # symbol) and the complete last one, plus every 13th sample in between (the full file is 0.9 MB)
    import scipy.io
    d = os.path.join(GOLD, 'matlab_pdsch')
    os.makedirs(d, exist_ok=True)
    src = os.path.join(PLAY, 'PDSCH', 'MatlabFiles')
    for f in ('dmrsSymbols.mat', 'pdschBits.mat', 'pdschSymbols.mat', 'pdschGrid.mat'):
        shutil.copyfile(os.path.join(src, f), os.path.join(d, f))
    w = scipy.io.loadmat(os.path.join(src, 'txWaveform.mat'))['txWaveform'].T            # (4, 15360)
    n = w.shape[1]
    idx = np.unique(np.concatenate([np.arange(0, 1200), np.arange(n - 1200, n), np.arange(0, n, 13)]))
    np.savez_compressed(os.path.join(d, 'txWaveform_samples.npz'), idx=np.int32(idx), samples=w[:, idx], n=np.int64(n))


def matlab_cdl():
    """CompareWithMatlab/CDL/CDL-Matlab.ipynb: CDL-D, 8 Tx (2x2 cross-polarised panel, MATLAB element order, rotated) x 2 Rx,
    angle scaling, MATLAB's random phases / ray coupling, 70 dB stop band -- MATLAB's input waveform through applyToSignal.
    Stored: the first three OFDM symbols' worth of the 1 ms input (the filter is causal), MATLAB's output and the reference's."""
    import scipy.io
    carrier = nr.Carrier(startRb=0, numRbs=25, spacing=15)
    bwp = carrier.curBwp
    phi, coupling = nr.CdlChannel.getMatlabRandomInit('D', 123)
    d = 15 * 1000 / 3600 * 4e9 / 299792458
    ch = nr.CdlChannel(bwp, 'D', delaySpread=10, carrierFreq=4e9, dopplerShift=d, initialPhases=phi, rayCoupling=coupling,
                       txAntenna=nr.AntennaPanel([2, 2], polarization="x", matlabOrder=True),
                       rxAntenna=nr.AntennaPanel([1, 1], polarization="+", matlabOrder=True),
                       txOrientation=[10, 20, 30], rxOrientation=[180, 0, 0],
                       angleScaling=([130, 70, 80, 110], [5, 11, 3, 3]), stopBandAtten=70)
    src = os.path.join(PLAY, 'CDL', 'MatlabFiles')
    tx = scipy.io.loadmat(os.path.join(src, 'txWaveform.mat'))['txWaveform'].T          # (8, 30720)
    rx_m = scipy.io.loadmat(os.path.join(src, 'rxWaveform.mat'))['rxWaveform'].T        # (2, 30720)
    n = int(bwp.symbolLens[:3].sum())
    tin = np.zeros_like(tx)                      # applyToSignal wants a whole slot: the rest is zero (causal filter: the
    tin[:, :n] = tx[:, :n]                       # first n output samples do not depend on it)
    rx = ch.applyToSignal(nr.Waveform(tin)).waveform[:, :n]
    nmse = float((np.abs(rx - rx_m[:, :n]) ** 2).sum() / (np.abs(rx_m[:, :n]) ** 2).sum())
    np.savez_compressed(os.path.join(GOLD, 'matlab_cdl.npz'), tx=tx[:, :n], rx_matlab=rx_m[:, :n], rx_ref=rx, nmse_ref=nmse,
                        slot_len=np.int64(tx.shape[1]))
    print('matlab_cdl: n', n, 'NMSE(reference vs MATLAB)', nmse)


def coding():
    out = {}
    rng = np.random.default_rng(2024)
    cases = [(1, 10000, 22808, 1, 2, 'QPSK', 0), (2, 2408, 7800, 1, 2, 'QPSK', 0), (1, 30216, 63648, 2, 4, '16QAM', 0),
             (2, 3817, 12000, 1, 6, '64QAM', 2), (1, 800, 2400, 2, 4, '16QAM', 3), (2, 100, 600, 1, 2, 'QPSK', 1),
             (1, 8425, 26000, 1, 8, '256QAM', 1), (2, 641, 2000, 1, 2, 'QPSK', 0), (1, 1000, 2880, 1, 2, 'QPSK', 0)]
    out['cases'] = np.int64([c[:5] + (c[6],) for c in cases])
    for i, (bg, A, G, nl, qm, mod, rv) in enumerate(cases):
        tb = rng.integers(0, 2, A).astype(np.int8)
        enc = nr.LdpcEncoder(baseGraphNo=bg, modulation=mod, txLayers=nl)
        tbc = enc.appendCrc(tb, '24A')
        cbs = enc.doSegmentation(tbc)
        coded = enc.encode(cbs)
        rm = enc.rateMatch(coded, G, rv=rv)
        dec = nr.LdpcDecoder(bg, mod, nl)
        sigma = 0.8 if i % 2 == 0 else 1.05
        llr = (1 - 2.0 * rm) * 2 / sigma ** 2 + rng.normal(0, 2 / sigma, len(rm))

        class H:
            pass
        h = H()
        h.decBuffer, h.rv = None, rv
        rr = dec.recoverRate(llr, A, harq=h)
        bel = dec.decode(rr, numIter=6, onlyInfoBits=False, outputBelief=True)
        merged, crc = dec.checkCrcAndMerge(np.int8(bel[:, :enc.codeBlockSize] < 0))
        p = f'c{i}_'
        out[p + 'tb'], out[p + 'params'] = tb, np.int64([enc.numCodeBlocks, enc.liftingSize, enc.setIndex,
                                                         enc.codeBlockSize, enc.numFillerBits])
        out[p + 'cbs'], out[p + 'coded'], out[p + 'rm'] = np.packbits(cbs.astype(np.uint8)), np.packbits(coded.astype(np.uint8)), np.packbits(rm.astype(np.uint8))
        out[p + 'llr'] = llr
        out[p + 'rr_sum'] = np.float64([np.where(rr > 1e19, 0, rr).sum(), (rr > 1e19).sum()])
        # all columns for the small cases, information columns only for the large ones (fixture size)
        out[p + 'belief'] = bel if bel.size <= 20000 else bel[:, :enc.codeBlockSize]
        out[p + 'merged'], out[p + 'crc'] = np.packbits(merged.astype(np.uint8)), np.asarray(crc, dtype=bool)
    # HARQ-IR soft combining: 4 redundancy versions into one buffer
    bg, A, G, nl, qm, mod = 1, 10000, 20900, 1, 4, '16QAM'
    tb = rng.integers(0, 2, A).astype(np.int8)
    enc = nr.LdpcEncoder(baseGraphNo=bg, modulation=mod, txLayers=nl)
    coded = enc.encode(enc.doSegmentation(enc.appendCrc(tb, '24A')))
    dec = nr.LdpcDecoder(bg, mod, nl)

    class H:
        pass
    h = H()
    h.decBuffer = None
    out['harq_tb'] = tb
    for t, rv in enumerate((0, 2, 3, 1)):
        rm = enc.rateMatch(coded, G, rv=rv)
        llr = (1 - 2.0 * rm) + rng.normal(0, 1.2, len(rm))
        h.rv = rv
        rr = dec.recoverRate(llr, A, harq=h)
        out[f'harq_llr{t}'] = llr
        out[f'harq_buf{t}'] = h.decBuffer.copy()
    # CRC known answers for every polynomial
    msg = rng.integers(0, 2, (3, 517)).astype(np.int8)
    out['crc_msg'] = msg
    for poly in ('6', '11', '16', '24A', '24B', '24C'):
        out['crc_' + poly] = enc.getCrc(msg, poly)
    # segmentation anchors B -> (C, Zc, iLS, K)   
    anchors = []
    for bg, B in ((1, 2432), (1, 30240), (1, 129152), (1, 606528), (1, 8449), (2, 2432), (2, 3841), (2, 641), (2, 192), (2, 561), (2, 3840)):
        e = nr.LdpcEncoder(baseGraphNo=bg)
        e.initialize(B)
        anchors.append((bg, B, e.numCodeBlocks, e.liftingSize, e.setIndex, e.codeBlockSize))
    out['seg_anchors'] = np.int64(anchors)
    np.savez_compressed(os.path.join(GOLD, 'coding.npz'), **out)


def coding_lbrm():
    """Limited-buffer rate matching (nRef > 0; ldpc.py:1093-1159 rateMatch, :1347-1418 recoverRate): the reference's rate-matched bits
    and its (C, Ncb) rate-recovered LLRs -- with nRef < N its own decode() then stops on that shape (ldpc.py:1538), so the
    fixture ends at recoverRate -- plus the HARQ buffers of rv 0 -> 2 soft-combined under LBRM.  VERDICT r5 weak #1."""
    out = {}
    rng = np.random.default_rng(2026)
    cases = [(1, 10000, 22808, 1, 2, 'QPSK', 0, 9000), (1, 10000, 22808, 1, 2, 'QPSK', 2, 9000), (2, 3000, 9000, 1, 2, 'QPSK', 3, 5000),
             (1, 12000, 26000, 2, 4, '16QAM', 1, 12000), (2, 3817, 12000, 1, 6, '64QAM', 0, 4000), (1, 8425, 30000, 1, 8, '256QAM', 3, 25000),
             (2, 641, 4000, 1, 2, 'QPSK', 2, 1500)]
    out['cases'] = np.int64([(c[0], c[1], c[2], c[3], c[4], c[6], c[7]) for c in cases])

    class H:
        pass
    for i, (bg, A, G, nl, qm, mod, rv, nref) in enumerate(cases):
        tb = rng.integers(0, 2, A).astype(np.int8)
        enc = nr.LdpcEncoder(baseGraphNo=bg, modulation=mod, txLayers=nl, nRef=nref)
        coded = enc.encode(enc.doSegmentation(enc.appendCrc(tb, '24A')))
        rm = enc.rateMatch(coded, G, rv=rv)
        dec = nr.LdpcDecoder(bg, mod, nl, nRef=nref)
        llr = np.float64(np.float32((1 - 2.0 * rm) * 2.5 + rng.normal(0, 1.6, len(rm))))     # float32-representable: the fixture stores float32
        h = H()
        h.decBuffer, h.rv = None, rv
        rr = dec.recoverRate(llr, A, harq=h)
        p = f'c{i}_'
        out[p + 'tb'], out[p + 'llr'] = tb, np.float32(llr)
        out[p + 'params'] = np.int64([enc.numCodeBlocks, enc.liftingSize, enc.codeBlockSize, enc.numFillerBits, rr.shape[1]])
        out[p + 'coded'], out[p + 'rm'] = np.packbits(coded.astype(np.uint8)), np.packbits(rm.astype(np.uint8))
        out[p + 'rr'] = np.where(rr > 1e19, np.inf, rr)             # fillers as inf: LARGE_LLR is the library's business
    # two transmissions into one limited buffer
    bg, A, G, nl, qm, mod, nref = 1, 10000, 20900, 1, 4, '16QAM', 14000
    tb = rng.integers(0, 2, A).astype(np.int8)
    enc = nr.LdpcEncoder(baseGraphNo=bg, modulation=mod, txLayers=nl, nRef=nref)
    coded = enc.encode(enc.doSegmentation(enc.appendCrc(tb, '24A')))
    dec = nr.LdpcDecoder(bg, mod, nl, nRef=nref)
    h = H()
    h.decBuffer = None
    out['harq_tb'], out['harq_params'] = tb, np.int64([bg, A, G, nl, qm, nref])
    for t, rv in enumerate((0, 2, 3)):
        rm = enc.rateMatch(coded, G, rv=rv)
        llr = np.float64(np.float32((1 - 2.0 * rm) + rng.normal(0, 1.2, len(rm))))
        h.rv = rv
        dec.recoverRate(llr, A, harq=h)
        out[f'harq_llr{t}'] = np.float32(llr)
    out['harq_buf'] = h.decBuffer.copy()                           # after the three transmissions
    np.savez_compressed(os.path.join(GOLD, 'coding_lbrm.npz'), **out)


def phy():
    out = {}
    rng = np.random.default_rng(7)
    for ci in (32769, 1, 123456789):
        out[f'gold_{ci}'] = np.int8(goldSequence(ci, 2000))
    for mod in ('BPSK', 'QPSK', '16QAM', '64QAM', '256QAM', '1024QAM'):
        m = nr.Modem(mod)
        out['const_' + mod] = m.constellation
        b = rng.integers(0, 2, m.qm * 64).astype(np.int8)
        y = m.modulate(b) + 0.12 * (rng.normal(size=64) + 1j * rng.normal(size=64))
        out['bits_' + mod], out['rx_' + mod] = b, y
        out['llr_' + mod] = m.getLLRsFromSymbols(y, 0.03)
        if m.qm <= 6:
            out['llrx_' + mod] = m.getLLRsFromSymbols(y, 0.3, useMax=False)
    np.savez_compressed(os.path.join(GOLD, 'phy.npz'), **out)


def host():
    out = {}
    cfgs = [dict(numRbs=25, spacing=15, layers=1, mod='QPSK', dm=dict(configType=1, additionalPos=1), pk={}),
            dict(numRbs=51, spacing=30, layers=2, mod='16QAM', dm=dict(configType=2, additionalPos=2), pk={}),
            dict(numRbs=52, spacing=30, layers=2, mod='16QAM', dm=dict(configType=1, additionalPos=1, symbols=2),
                 pk=dict(interleavingBundleSize=2), startRb=1),
            dict(numRbs=30, spacing=15, layers=4, mod='256QAM', dm=dict(configType=1, additionalPos=3), pk={}),
            dict(numRbs=24, spacing=60, layers=3, mod='64QAM', dm=dict(configType=2, additionalPos=1, otherCdmGroups=[2]),
                 pk=dict(symStart=2, symLen=10, mappingType='B', prbSet=list(range(4, 20))))]
    import json
    out['cfgs'] = np.array(json.dumps(cfgs))
    for i, c in enumerate(cfgs):
        kw = dict(numRbs=c['numRbs'], spacing=c['spacing'])
        if 'startRb' in c:
            kw['startRb'] = c['startRb']
        car = nr.Carrier(**kw)
        bwp = car.curBwp
        p = nr.PDSCH(bwp, numLayers=c['layers'], modulation=c['mod'], **c['pk'])
        p.setDMRS(**c['dm'])
        out[f'h{i}_numerology'] = np.int64([bwp.nFFT] + bwp.symbolLens.tolist())
        for slot in (0, 7):
            car.slotNo = slot
            g = p.getGrid()
            idx = np.nonzero(g.reTypeIds == g.retNameToId['DMRS'])
            out[f'h{i}_s{slot}_types'] = g.reTypeIds
            out[f'h{i}_s{slot}_dmrs'] = g.grid[idx]
            out[f'h{i}_s{slot}_data'] = np.int32(p.dataIndices)
            out[f'h{i}_s{slot}_lm'] = np.int32(p.getLayerMapIndexes(p.dataIndices)[0])
        out[f'h{i}_tbs'] = np.int64([p.getTxBlockSize(r)[0] for r in (0.2, 0.3, 0.5, 666 / 1024, 0.75, 0.92)])
        out[f'h{i}_bits'] = np.int64(p.getBitSizes(g))
        out[f'h{i}_dataREs'] = np.int64(p.dmrs.dataREs)
    np.savez_compressed(os.path.join(GOLD, 'host.npz'), **out)


def channels():
    out = {}
    specs = [('cdl', 'C', dict(delaySpread=300, carrierFreq=4e9, dopplerShift=5), ([1, 2], [1, 2])),
             ('cdl', 'D', dict(delaySpread=100, dopplerShift=50, ueDirAZ=[30, 80]), ([1, 2], [1, 1])),
             ('cdl', 'A', dict(delaySpread=30, dopplerShift=100, angleScaling=([120, 200, 90, 95], [10, 30, 5, 8])), ([2, 2], [1, 1])),
             ('tdl', 'A', dict(delaySpread=30, dopplerShift=5), None),
             ('tdl', 'C', dict(delaySpread=300, dopplerShift=100, txAntennaCount=2, rxAntennaCount=2, mimoCorrelation='Medium'), None),
             ('tdl', 'D', dict(delaySpread=100, dopplerShift=30, txAntennaCount=4, rxAntennaCount=2, mimoCorrelation='High'), None)]
    for i, (kind, prof, kw, ant) in enumerate(specs):
        nr.random.setSeed(100 + i)
        car = nr.Carrier(numRbs=25, spacing=15)
        if kind == 'cdl':
            ch = nr.CdlChannel(car.curBwp, prof, txAntenna=nr.AntennaPanel(ant[0], polarization='x'),
                               rxAntenna=nr.AntennaPanel(ant[1], polarization='x'), **kw)
        else:
            ch = nr.TdlChannel(car.curBwp, prof, **kw)
        ch.goNext()
        ch.goNext()                                    # slot 2: non-trivial absolute time
        ch.prepareForNextSlot()
        out[f'ch{i}_samples'] = ch.chanGainSamples
        out[f'ch{i}_gains1'] = ch.chanGains1
        out[f'ch{i}_coeff'] = ch.coeffMatrix
        out[f'ch{i}_misc'] = np.int64([ch.chanOffset, ch.getMaxDelay()])
        H = ch.getChannelMatrix()
        out[f'ch{i}_H'] = H[::6, ::25]                 # sub-sampled channel matrix
    np.savez_compressed(os.path.join(GOLD, 'channels.npz'), **out)


def ptrs():
    """PTRS configuration + insertion (dmrs.py:554-797) on three PDSCHs: grid values at the PTRS REs, their positions, the bit
    sizes left for data, the PTRS symbol set -- for slots 0 and 7."""
    out = {}
    cfgs = [dict(numRbs=25, spacing=15, layers=2, mod='16QAM', dm=dict(configType=1, additionalPos=1), pk={},
                 pt=dict(timeDensity=2, freqDensity=2, reOffset='01')),
            dict(numRbs=51, spacing=30, layers=3, mod='64QAM', dm=dict(configType=2, additionalPos=2),
                 pk=dict(interleavingBundleSize=2, rnti=7), pt=dict(mcsi=(5, 10, 20), iMCS=12, nRBi=(10, 40), epreRatio=1, reOffset=2)),
            dict(numRbs=24, spacing=60, layers=1, mod='QPSK', dm=dict(configType=1, additionalPos=0),
                 pk=dict(symStart=2, symLen=9, mappingType='B', prbSet=list(range(3, 20)), rnti=3), pt=dict(timeDensity=1, freqDensity=4))]
    import json
    out['cfgs'] = np.array(json.dumps(cfgs))
    for i, c in enumerate(cfgs):
        car = nr.Carrier(numRbs=c['numRbs'], spacing=c['spacing'])
        p = nr.PDSCH(car.curBwp, numLayers=c['layers'], modulation=c['mod'], **c['pk'])
        p.setDMRS(**c['dm'])
        p.setPTRS(**c['pt'])
        out[f'p{i}_dens'] = np.int64([p.dmrs.ptrs.timeDensity, p.dmrs.ptrs.freqDensity])
        out[f'p{i}_syms'] = np.int64(p.dmrs.ptrs.symSet)
        for slot in (0, 7):
            car.slotNo = slot
            g = p.getGrid()
            idx = np.nonzero(g.reTypeIds == g.retNameToId['PTRS'])
            out[f'p{i}_s{slot}_idx'] = np.int32(np.stack(idx))
            out[f'p{i}_s{slot}_val'] = g.grid[idx]
            out[f'p{i}_s{slot}_bits'] = np.int64(p.getBitSizes(g))
    np.savez_compressed(os.path.join(GOLD, 'ptrs.npz'), **out)


def csirs():
    """CSI-RS (csirs.py) on ten configurations that cover the rows of TS 38.211 Table 7.4.1.5.3-1 with every CDM type, both
    densities, two-symbol rows, ZP + NZP sets with periods/offsets: populated values and RE types, the reserved map;
    then CSI-RS based channel estimation (grid.py:740-873) through every interpolation kind (every fifth subcarrier
    of the estimate is stored), and the timing estimate
    (grid.py:592-622)."""
    import json
    out = {}
    cfgs = [dict(rb=24, sp=15, slot=0, kw=dict(numPorts=1)),
            dict(rb=24, sp=15, slot=4, kw=dict(numPorts=1, density=3, freqMap='0100', symbols=[6], powerDb=3)),
            dict(rb=25, sp=30, slot=8, kw=dict(numPorts=2, density=0.5, scramblingID=7, symbols=[3])),
            dict(rb=24, sp=15, slot=0, kw=dict(numPorts=4, freqMap='010')),
            dict(rb=24, sp=15, slot=0, kw=dict(numPorts=4, freqMap='000100', symbols=[9])),
            dict(rb=24, sp=30, slot=12, kw=dict(numPorts=8, cdmSize=4, symbols=[4], scramblingID=100)),
            dict(rb=26, sp=15, slot=0, kw=dict(numPorts=16, cdmSize=4, density=0.5, symbols=[7])),
            dict(rb=24, sp=15, slot=0, kw=dict(numPorts=32, cdmSize=8, symbols=[5])),
            dict(rb=24, sp=15, slot=0, kw=dict(numPorts=24, cdmSize=2, symbols=[2, 8])),
            dict(rb=52, sp=30, slot=5, kw=dict(numPorts=8, cdmSize=2, freqMap='001111', symbols=[10], startRb=4, numRbs=40,
                                                period=5, offset=0, powerDb=-3, scramblingID=33))]
    out['cfgs'] = np.array(json.dumps(cfgs))
    for i, c in enumerate(cfgs):
        car = nr.Carrier(numRbs=c['rb'], spacing=c['sp'])
        car.slotNo = c['slot']
        bwp = car.curBwp
        cc = nr.CsiRsConfig(csiType='NZP', bwp=bwp, **c['kw'])
        r = cc.csiRsSetList[0].csiRsList[0]
        out[f'c{i}_row'] = np.int64([r.row] + list(r.ks) + list(r.ls))
        g = bwp.createGrid(cc.numPorts)
        cc.populateGrid(g)
        idx = np.nonzero(g.reTypeIds == g.retNameToId['CSIRS_NZP'])
        out[f'c{i}_idx'] = np.int32(np.stack(idx))
        out[f'c{i}_val'] = g.grid[idx]
        out[f'c{i}_untouched'] = np.int64([(g.reTypeIds == g.retNameToId['UNASSIGNED']).sum()])
        g2 = bwp.createGrid(3)
        cc.reserveGridResources(g2)
        out[f'c{i}_res'] = np.int32(np.stack(np.nonzero(g2.reTypeIds == g2.retNameToId['CSIRS_NZP'])))
    # a ZP set and an NZP set with different slot behaviour: which slots of 0..9 carry which
    car = nr.Carrier(numRbs=24, spacing=15)
    bwp = car.curBwp
    zp = nr.CsiRsSet("ZP", bwp, csiRsList=[nr.CsiRs(offset=1, symbols=[1], numPorts=1, freqMap="000001000000", density=0.5)],
                     resourceType='semiPersistent', period=10)
    nz = nr.CsiRsSet("NZP", bwp, csiRsList=[nr.CsiRs(offset=0, symbols=[1], numPorts=1, freqMap="0010", density=3),
                                             nr.CsiRs(offset=3, symbols=[3], numPorts=1, freqMap="000000001000", density=1)],
                     resourceType='periodic', period=5)
    cc = nr.CsiRsConfig([zp, nz])
    counts = []
    for slot in range(10):
        car.slotNo = slot
        g = bwp.createGrid(1)
        cc.populateGrid(g)
        counts.append([(g.reTypeIds == g.retNameToId[t]).sum() for t in ('CSIRS_ZP', 'CSIRS_NZP')])
        if slot in (0, 1, 3):
            out[f'mix_s{slot}_types'] = g.reTypeIds
            out[f'mix_s{slot}_grid'] = g.grid
    out['mix_counts'] = np.int64(counts)

    # estimation: a smooth channel (a few delayed taps, slow time variation) applied to the CSI-RS grid, noise from the seed
    rng = np.random.default_rng(2024)
    est = [dict(rb=24, sp=15, nr=2, kw=dict(numPorts=4, freqMap='000100', symbols=[9])),          # ports on two symbols
           dict(rb=25, sp=30, nr=3, kw=dict(numPorts=2, density=0.5, symbols=[3])),
           dict(rb=24, sp=15, nr=2, kw=dict(numPorts=8, cdmSize=4, symbols=[4])),
           dict(rb=24, sp=15, nr=1, kw=dict(numPorts=24, cdmSize=2, symbols=[2, 8])),               # four pilot symbols
           dict(rb=24, sp=15, nr=1, kw=dict(numPorts=1, density=3, freqMap='0100', symbols=[6]))]
    out['est_cfgs'] = np.array(json.dumps(est))
    for i, c in enumerate(est):
        car = nr.Carrier(numRbs=c['rb'], spacing=c['sp'])
        bwp = car.curBwp
        cc = nr.CsiRsConfig(csiType='NZP', bwp=bwp, **c['kw'])
        nt = cc.numPorts
        tx = bwp.createGrid(nt)
        cc.populateGrid(tx)
        L, K = tx.shape[1:]
        taps = (rng.standard_normal((4, c['nr'], nt)) + 1j * rng.standard_normal((4, c['nr'], nt))) * np.float64([1, .6, .3, .15])[:, None, None]
        dl = np.float64([0, 1.3, 2.9, 5.2])
        dop = rng.uniform(-.02, .02, (4,))
        k = np.arange(K)[None, :, None]
        l = np.arange(L)[:, None, None]
        h = (taps[None, None] * np.exp(-2j * np.pi * (k * dl / 256 - l * dop))[..., None, None]).sum(2)       # (L,K,nr,nt)
        rx = tx.applyChannel(h)
        noise = (rng.standard_normal(rx.shape) + 1j * rng.standard_normal(rx.shape)) * 0.02
        rx.grid = rx.grid + noise
        out[f'e{i}_rx'] = rx.grid
        he, nv = rx.estimateChannelLS(cc)
        out[f'e{i}_lin'], out[f'e{i}_lin_nv'] = he[:, ::5], np.float64([nv])
        he, nv = rx.estimateChannelLS(cc, polarInt=True)
        out[f'e{i}_pol'], out[f'e{i}_pol_nv'] = he[:, ::5], np.float64([nv])
        if i in (0, 1, 3):
            for kern in ('nearest', 'quadratic', 'thin_plate_spline', 'multiquadric'):
                he, nv = rx.estimateChannelLS(cc, kernel=kern)
                out[f'e{i}_{kern}'], out[f'e{i}_{kern}_nv'] = he[:, ::5], np.float64([nv])
            he, nv = rx.estimateChannelLS(cc, meanCdm=False)
            out[f'e{i}_nomean'], out[f'e{i}_nomean_nv'] = he[:, ::5], np.float64([nv])
        if i == 3:
            he, nv, hps = rx.estimateChannelLsEx(cc)                       # the Ex defaults: polar + 2-D thin-plate spline
            out[f'e{i}_ex'], out[f'e{i}_ex_nv'] = he[:, ::5], np.float64([nv])
            he, nv, hps = rx.estimateChannelLsEx(cc, polarInt=False, int2d=True, kernel='thin_plate_spline', neighbors=9, smoothing=0.1)
            out[f'e{i}_ex2'], out[f'e{i}_ex2_nv'] = he[:, ::5], np.float64([nv])
    # DMRS pilots on four symbols (additionalPos 3): interpolation along the symbols with every kind, and the 2-D RBF path
    car = nr.Carrier(numRbs=24, spacing=15)
    bwp = car.curBwp
    pd = nr.PDSCH(bwp, numLayers=2, modulation='16QAM')
    pd.setDMRS(configType=1, additionalPos=3)
    tx = pd.getGrid()
    bits = rng.integers(0, 2, pd.getBitSizes(tx)[0], dtype=np.int8)
    pd.populateGrid(tx, bits)
    L, K = tx.shape[1:]
    taps = (rng.standard_normal((4, 2, 2)) + 1j * rng.standard_normal((4, 2, 2))) * np.float64([1, .6, .3, .15])[:, None, None]
    k = np.arange(K)[None, :, None]
    l = np.arange(L)[:, None, None]
    h = (taps[None, None] * np.exp(-2j * np.pi * (k * np.float64([0, 1.3, 2.9, 5.2]) / 256 - l * np.float64([.01, -.02, .015, 0])))[..., None, None]).sum(2)
    rx = tx.applyChannel(h)
    rx.grid = rx.grid + (rng.standard_normal(rx.shape) + 1j * rng.standard_normal(rx.shape)) * 0.02
    out['dm_rx'] = rx.grid
    out['dm_bits'] = bits
    for kern in ('linear', 'nearest', 'quadratic', 'thin_plate_spline', 'multiquadric'):
        he, nv = rx.estimateChannelLS(pd.dmrs, kernel=kern, polarInt=(kern == 'quadratic'))
        out[f'dm_{kern}'], out[f'dm_{kern}_nv'] = he[:, ::5], np.float64([nv])
    he, nv, hps = rx.estimateChannelLsEx(pd.dmrs)
    out['dm_ex'], out['dm_ex_nv'], out['dm_ex_hk0'] = he[:, ::5], np.float64([nv]), hps[0][:, ::5]
    he, nv, hps = rx.estimateChannelLsEx(pd.dmrs, polarInt=False, neighbors=14, smoothing=0.05)
    out['dm_ex2'], out['dm_ex2_nv'] = he[:, ::5], np.float64([nv])
    he, nv, hps = rx.estimateChannelLsEx(pd.dmrs, polarInt=False, kernel='linear', neighbors=12, degree=1)
    out['dm_ex3'], out['dm_ex3_nv'] = he[:, ::5], np.float64([nv])
    # timing estimate: the CSI-RS grid's waveform against a delayed, noisy copy through a 2x1 mix
    car = nr.Carrier(numRbs=24, spacing=15)
    bwp = car.curBwp
    cc = nr.CsiRsConfig(csiType='NZP', bwp=bwp, numPorts=2, symbols=[3])
    tx = bwp.createGrid(2)
    cc.populateGrid(tx)
    w = tx.ofdmModulate(windowing="NONE").waveform
    for d in (0, 7, 23):
        mix = np.complex128([[0.8, 0.3j], [-0.2, 0.9]])
        rxw = np.concatenate([np.zeros((2, d), complex), (mix @ w)[:, :w.shape[1] - d]], 1)
        rxw = rxw + (rng.standard_normal(rxw.shape) + 1j * rng.standard_normal(rxw.shape)) * 1e-3
        out[f't{d}_rx'] = rxw[:, ::1].astype(np.complex64)
        out[f't{d}_off'] = np.int64([tx.estimateTimingOffset(nr.Waveform(rxw.astype(np.complex64).astype(np.complex128)))])
    np.savez_compressed(os.path.join(GOLD, 'csirs.npz'), **out)


def csi_channel(seed, L, K, nr, nt):
    """Seeded smooth channel used by the csifeedback fixtures and their tests: four delayed taps with slow phase drift."""
    rng = np.random.default_rng(seed)
    taps = (rng.standard_normal((4, nr, nt)) + 1j * rng.standard_normal((4, nr, nt))) * np.float64([1, .7, .4, .2])[:, None, None]
    k = np.arange(K)[None, :, None]
    l = np.arange(L)[:, None, None]
    ph = np.exp(-2j * np.pi * (k * np.float64([0, 1.7, 3.1, 6.4]) / 512 - l * np.float64([.004, -.006, .002, 0])))
    return (taps[None, None] * ph[..., None, None]).sum(2) / 2


def csifeedback():
    """CSI report (csifeedback.py): Type-I single-panel codebooks (indices + precoders), the per-rank PMI search and the rank
    choice on seeded channels, sub-band layouts -- for the configurations the reference can run (one-row panels)."""
    import json
    out = {}
    cfgs = [dict(rb=24, ports=4, cdm=2, kw=dict(freqMap='000100'), rep=dict(n1=2, n2=1), nr=4, nv=0.01, ranks=[1, 2, 3, 4]),
            dict(rb=24, ports=8, cdm=2, kw={}, rep=dict(n1=4, n2=1), nr=4, nv=0.02, ranks=[1, 2, 3, 4]),
            dict(rb=24, ports=8, cdm=4, kw={}, rep=dict(n1=4, n2=1, codebookMode=2, prgSize=2, cbRiRestriction='00011111'), nr=8, nv=0.01, ranks=[1, 2, 3, 4, 5]),
            dict(rb=52, ports=16, cdm=4, kw={}, rep=dict(n1=8, n2=1, subbandSize=8), nr=4, nv=0.005, ranks=[1, 2, 3, 4]),
            dict(rb=24, ports=32, cdm=8, kw={}, rep=dict(n1=16, n2=1, prgSize=0, cbRiRestriction='00000011'), nr=2, nv=0.05, ranks=[1, 2]),
            dict(rb=20, ports=12, cdm=2, kw={}, rep=dict(n1=6, n2=1), nr=4, nv=0.01, ranks=[1, 2, 3, 4]),
            dict(rb=24, ports=2, cdm=2, kw={}, rep=dict(n1=1, n2=1, cbRiRestriction='00000001'), nr=2, nv=0.01, ranks=[1])]
    out['cfgs'] = np.array(json.dumps(cfgs))
    for i, c in enumerate(cfgs):
        car = nr.Carrier(numRbs=c['rb'], spacing=15)
        bwp = car.curBwp
        cc = nr.CsiRsConfig(csiType='NZP', bwp=bwp, numPorts=c['ports'], cdmSize=c['cdm'], **c['kw'])
        rep = nr.CsiReport(cc, **c['rep'])
        h = csi_channel(100 + i, 14, 12 * c['rb'], c['nr'], c['ports'])
        for rank in c['ranks']:
            idx, cb = rep.getCodebook(rank)
            out[f'r{i}_{rank}_idx'] = np.int32([list(a) + [b] for a, b in idx])
            out[f'r{i}_{rank}_cb'] = cb[::max(1, len(cb) // 24)]                  # a spread of entries (all when <= 24)
            pmi, ws, sb = rep.bestPmiForRank(h, rank, c['nv'])
            out[f'r{i}_{rank}_pmi'] = np.int32(list(pmi[0]) + list(pmi[1]))
            out[f'r{i}_{rank}_w'] = np.array([np.asarray(w).reshape(c['ports'], rank) for w in ws])
            out[f'r{i}_{rank}_sinr'] = np.concatenate([np.asarray(v) for v in sb])
        rank, pmi, sb = rep.getBestRank(h, c['nv'])
        out[f'r{i}_best'] = np.int32([rank] + list(pmi[0]) + list(pmi[1]))
        out[f'r{i}_subbands'] = np.int32(list(rep.subbands(4)) + [-1] + list(rep.subbands(8)))
        out[f'r{i}_cqi2pmi'] = np.array(json.dumps([rep.getCqiToPmiIdxes(0), rep.getCqiToPmiIdxes(2), rep.getCqiToPmiIdxes(4)]))
    car = nr.Carrier(startRb=3, numRbs=50, spacing=15)
    cc = nr.CsiRsConfig(csiType='NZP', bwp=car.curBwp, numPorts=8)
    rep = nr.CsiReport(cc, n1=4, n2=1, subbandSizeCqi=8, subbandSizePmi=4)
    out['off_subbands'] = np.int32(list(rep.subbands(4)) + [-1] + list(rep.subbands(8)))
    out['off_cqi2pmi'] = np.array(json.dumps([rep.getCqiToPmiIdxes(4), rep.getCqiToPmiIdxes(2)]))
    out['cqi_tables'] = np.array(json.dumps(nr.csifeedback.cqiTables))
    np.savez_compressed(os.path.join(GOLD, 'csifeedback.npz'), **out)


def csifeedback_multipanel():
    """Type-I MULTI-panel codebooks (csifeedback.py:566-577, 1040-1327) in the reference itself, every Ng-N1-N2 combination of TS 38.214
    Table 5.2.2.2.2-1 x codebook mode x 1..4 layers: what getCodebook does.  In v0.4.0 none of them yields a (ports x layers) codebook:
    mode 1 and every one-layer case raise, mode 2 with >= 2 layers returns arrays of another shape.  The fixture records, per
    configuration, the exception (type, start of the message, line of csifeedback.py) or the shape returned -- data that shows why
    neoradium_amd raises NotImplementedError there instead of restating it."""
    import json
    import traceback
    car = nr.Carrier(numRbs=24, spacing=15)
    rows = []
    for (ng, n1, n2) in ((2, 2, 1), (2, 4, 1), (4, 2, 1), (2, 2, 2), (2, 8, 1), (4, 4, 1), (2, 4, 2), (4, 2, 2)):
        ports = 2 * ng * n1 * n2
        for mode in (1, 2):
            if ng == 4 and mode == 2:
                continue                                      # (TS 38.214 5.2.2.2.2: mode 2 is not defined for Ng = 4; the constructor refuses it)
            for nl in (1, 2, 3, 4):
                cc = nr.CsiRsConfig(csiType='NZP', bwp=car.curBwp, numPorts=ports, cdmSize=(2 if ports <= 12 else (4 if ports <= 16 else 8)))
                rep = nr.CsiReport(cc, codebookType='Type1MP', ng=ng, n1=n1, n2=n2, codebookMode=mode)
                row = dict(ng=ng, n1=n1, n2=n2, ports=ports, mode=mode, layers=nl)
                try:
                    idx, cb = rep.getCodebook(nl)
                    row.update(outcome='returns', shape=list(np.asarray(cb).shape), entries=len(idx),
                               is_ports_by_layers=bool(np.asarray(cb).ndim == 3 and np.asarray(cb).shape[1:] == (ports, nl)))
                except Exception as e:
                    tb = traceback.extract_tb(e.__traceback__)[-1]
                    row.update(outcome='raises', exception=type(e).__name__, message=str(e)[:60], line=int(tb.lineno),
                               file=os.path.basename(tb.filename))
                rows.append(row)
    usable = [r for r in rows if r.get('is_ports_by_layers')]
    print(len(rows), 'multi-panel configurations,', len(usable), 'yield a ports x layers codebook')
    json.dump(dict(reference='InterDigitalInc/NeoRadium v0.4.0 csifeedback.py getCodebook(codebookType="Type1MP")', rows=rows),
              open(os.path.join(GOLD, 'csifeedback_multipanel.json'), 'w'), indent=1)


def ofdm_options():
    """The non-default kwargs of Grid.ofdmModulate / Waveform.ofdmDemodulate: two slots in one call, carrier up-conversion
    (f0 > 0), an FFT window that starts 30 % / 80 % into the CP.  Input grid from the seed; outputs stored."""
    out = {}
    car = nr.Carrier(numRbs=24, spacing=30)
    bwp = car.curBwp
    rng = np.random.default_rng(99)
    g = nr.Grid(bwp, numPlanes=2, numSlots=2)
    vals = rng.standard_normal(g.shape) + 1j * rng.standard_normal(g.shape)
    g.grid = vals
    for name, f0 in (('base', 0), ('up', 3.5e9)):
        w = g.ofdmModulate(f0=f0)
        out[f'w2_{name}'] = w.waveform[:, ::9]                     # every ninth sample of the two-slot waveform
        out[f'w2_{name}_shape'] = np.int64(w.waveform.shape)
        out[f'w2_{name}_head'] = w.waveform[:, :600]
        for ratio in (0.3, 0.8):
            rx = w.ofdmDemodulate(bwp, f0=f0, cpOffsetRatio=ratio)  # first slot
            out[f'rx_{name}_{int(ratio * 10)}'] = rx.grid
    np.savez_compressed(os.path.join(GOLD, 'ofdm_options.npz'), **out)


def channels_xiao():
    """TDL with the statistical sum-of-sinusoids model (sosType='Xiao'): new random angles / phases are drawn for every slot,
    so the fixture follows the generator: construction (first slot prepared), two goNext, the slot prepared again."""
    out = {}
    specs = [('B', dict(delaySpread=100, dopplerShift=70, sosType='Xiao')),
             ('D', dict(delaySpread=30, dopplerShift=20, sosType='Xiao', txAntennaCount=2, rxAntennaCount=2, mimoCorrelation='Medium'))]
    for i, (prof, kw) in enumerate(specs):
        nr.random.setSeed(300 + i)
        car = nr.Carrier(numRbs=25, spacing=15)
        ch = nr.TdlChannel(car.curBwp, prof, **kw)
        out[f'x{i}_samples0'] = ch.chanGainSamples
        out[f'x{i}_gains0'] = ch.chanGains
        ch.goNext()
        H = ch.getChannelMatrix()                      # prepares slot 1: the second pair of draws
        out[f'x{i}_samples1'] = ch.chanGainSamples
        out[f'x{i}_gains1'] = ch.chanGains
        out[f'x{i}_H'] = H[::6, ::25]
    np.savez_compressed(os.path.join(GOLD, 'channels_xiao.npz'), **out)


def chest():
    """Grid.estimateChannelLS with both subcarrier interpolators (complex-linear and polar-linear) and its noise
    side output, on noisy received grids of three small links (inputs: the received grid and the DMRS tables)."""
    out = {}
    cases = [  # name, seed, numRbs, spacing, layers, rx panel, tx panel, dmrs kwargs, snr dB
        ('a', 11, 25, 15, 2, [1, 1], [1, 1], dict(configType=1, additionalPos=1), 10.0),
        ('b', 12, 24, 30, 4, [1, 2], [1, 2], dict(configType=1, additionalPos=1), 32.0),
        ('c', 13, 25, 15, 1, [1, 1], [1, 1], dict(configType=1, additionalPos=0), 4.0),
        ('d', 14, 24, 30, 3, [1, 2], [1, 2], dict(configType=1, additionalPos=2, symbols=1), 14.0),
    ]
    out['names'] = np.array([c[0] for c in cases])
    for name, seed, numRbs, spacing, layers, rxp, txp, dm, snr in cases:
        nr.random.setSeed(seed)
        car = nr.Carrier(numRbs=numRbs, spacing=spacing)
        bwp = car.curBwp
        p = nr.PDSCH(bwp, numLayers=layers, nID=car.cellId, modulation='16QAM')
        p.setDMRS(**dm)
        ch = nr.CdlChannel(bwp, 'C', delaySpread=300, carrierFreq=4e9, dopplerShift=5,
                           txAntenna=nr.AntennaPanel(txp, polarization='x'), rxAntenna=nr.AntennaPanel(rxp, polarization='x'))
        g = p.getGrid()
        nb = p.getBitSizes(g)
        p.populateGrid(g, nr.random.bits(nb[0]))
        H = ch.getChannelMatrix()
        F = p.getPrecodingMatrix(H)
        rx = g.precode(F).applyChannel(H).addNoise(snrDb=snr, useRxPower=True)
        h_lin, nv_lin = rx.estimateChannelLS(p.dmrs, polarInt=False, kernel='linear')
        h_pol, nv_pol = rx.estimateChannelLS(p.dmrs, polarInt=True, kernel='linear')
        # the DMRS tables in the layout of the oracle / kernels: pilots (P, nDmrsSym, nK), subcarriers (P, nK), symbols
        rs = bwp.createGrid(len(p.portSet))
        p.dmrs.populateGrid(rs)
        idx = rs.getReIndexes("DMRS")
        pil, pks, ds = [], [], None
        for port in range(len(p.portSet)):
            pl, pk = idx[1][idx[0] == port], idx[2][idx[0] == port]
            ls = np.unique(pl)
            ks = pk[pl == ls[0]]
            pil.append(rs.grid[port, ls, :][:, ks])
            pks.append(ks)
            ds = ls
        out[name + '_cfg'] = np.array(repr(dict(seed=seed, numRbs=numRbs, spacing=spacing, layers=layers, rx=rxp, tx=txp,
                                                dmrs=dm, snr=snr)))
        out[name + '_rx'] = rx.grid
        out[name + '_true_noise_var'] = np.float64(rx.noiseVar)
        out[name + '_pilots'] = np.stack(pil)
        out[name + '_port_ks'] = np.int32(np.stack(pks))
        out[name + '_dmrs_syms'] = np.int32(ds)
        out[name + '_geom'] = np.int64([p.dmrs.symbols, 4 if p.dmrs.enhanced else 2, bwp.nFFT,
                                        min(bwp.symbolLens) - bwp.nFFT, spacing])
        out[name + '_h_lin'] = h_lin[::3]
        out[name + '_h_pol'] = h_pol[::3]
        out[name + '_nv'] = np.float64([nv_lin, nv_pol])
        print('chest', name, 'true nv', rx.noiseVar, 'est', nv_lin, nv_pol)
    np.savez_compressed(os.path.join(GOLD, 'chest.npz'), **out)


def prg():
    """PRG precoding (prgSize 2 / 4 and wideband on a partial allocation): PDSCH.getPrecodingMatrix groups + matrices and
    Grid.precode with the per-group list (inputs: channel matrix and populated grid)."""
    out = {}
    cases = [('p2', 21, 13, 15, 2, 2, None), ('p4', 22, 13, 15, 2, 4, None), ('w_part', 23, 13, 15, 1, 0, list(range(3, 11)))]
    out['names'] = np.array([c[0] for c in cases])
    for name, seed, numRbs, spacing, layers, prg, prbs in cases:
        nr.random.setSeed(seed)
        car = nr.Carrier(numRbs=numRbs, spacing=spacing)
        bwp = car.curBwp
        kw = dict(numLayers=layers, nID=car.cellId, modulation='16QAM', prgSize=prg)
        if prbs is not None:
            kw['prbSet'] = prbs
        p = nr.PDSCH(bwp, **kw)
        p.setDMRS(configType=1, additionalPos=1)
        ch = nr.CdlChannel(bwp, 'C', delaySpread=300, carrierFreq=4e9, dopplerShift=5,
                           txAntenna=nr.AntennaPanel([1, 2], polarization='x'), rxAntenna=nr.AntennaPanel([1, 1], polarization='x'))
        g = p.getGrid()
        p.populateGrid(g, nr.random.bits(p.getBitSizes(g)[0]))
        H = ch.getChannelMatrix()
        F = p.getPrecodingMatrix(H)
        assert isinstance(F, list)
        pg = g.precode(F)
        out[name + '_cfg'] = np.array(repr(dict(seed=seed, numRbs=numRbs, spacing=spacing, layers=layers, prgSize=prg, prbSet=prbs)))
        out[name + '_H'] = H
        out[name + '_grid'] = g.grid
        out[name + '_n_groups'] = np.int64(len(F))
        for i, (rbs, f) in enumerate(F):
            out[f'{name}_rbs{i}'] = np.int64(rbs)
            out[f'{name}_f{i}'] = f
        out[name + '_precoded'] = pg.grid
        print('prg', name, 'groups', [list(map(int, r)) for r, _ in F])
    np.savez_compressed(os.path.join(GOLD, 'prg.npz'), **out)


def e2e():
    """Whole slots.  The random stream (bits -> channel construction -> noise) is reproducible from the seed with
    NumPy's PCG64, so only outputs are stored."""
    def one(name, seed, numRbs, spacing, mod, layers, rate, bg, chan, dm, numIter, snr, freqDomain, perfect, slot0):
        nr.random.setSeed(seed)
        car = nr.Carrier(numRbs=numRbs, spacing=spacing)
        bwp = car.curBwp
        p = nr.PDSCH(bwp, numLayers=layers, nID=car.cellId, modulation=mod)
        p.setDMRS(**dm)
        if chan[0] == 'cdl':
            ch = nr.CdlChannel(bwp, chan[1], delaySpread=chan[2], carrierFreq=4e9, dopplerShift=chan[3],
                               txAntenna=nr.AntennaPanel(chan[4], polarization='x'),
                               rxAntenna=nr.AntennaPanel(chan[5], polarization='x'))
        else:
            ch = nr.TdlChannel(bwp, chan[1], delaySpread=chan[2], dopplerShift=chan[3], txAntennaCount=chan[4], rxAntennaCount=chan[5])
        for _ in range(slot0):
            ch.goNext()
        enc = nr.LdpcEncoder(baseGraphNo=bg, modulation=mod, txLayers=layers, targetRate=rate)
        dec = enc.getDecoder()
        g = p.getGrid()
        tbs = p.getTxBlockSize(rate)
        tb = nr.random.bits(tbs[0])
        nb = p.getBitSizes(g)
        rm = enc.getRateMatchedCodeBlocks(tb, nb[0])
        p.populateGrid(g, rm)
        idx = p.getReIndexes(g, "PDSCH")
        H = ch.getChannelMatrix()
        F = p.getPrecodingMatrix(H)
        pg = g.precode(F)
        if freqDomain:
            rx = pg.applyChannel(H).addNoise(snrDb=snr, useRxPower=True)
        else:
            w = pg.ofdmModulate().pad(ch.getMaxDelay())
            r = ch.applyToSignal(w).addNoise(snrDb=snr, bwp=bwp, useRxPower=True)
            rx = r.sync(ch.getTimingOffset()).ofdmDemodulate(bwp)
        hest = (H @ F[None, ...]) if perfect else rx.estimateChannelLS(p.dmrs, polarInt=False, kernel='linear')[0]
        eq, sc = rx.equalize(hest)
        llr = p.getLLRsFromGrid(eq, idx, sc)[0]
        rr = dec.recoverRate(llr, tbs[0])
        db = dec.decode(rr, numIter=numIter)
        out_tb, crc = dec.checkCrcAndMerge(db)
        np.savez_compressed(os.path.join(GOLD, f'e2e_{name}.npz'),
                            cfg=np.array(repr(dict(seed=seed, numRbs=numRbs, spacing=spacing, mod=mod, layers=layers,
                                                   rate=rate, bg=bg, chan=chan, dm=dm, numIter=numIter, snr=snr,
                                                   freqDomain=freqDomain, perfect=perfect, slot0=slot0))),
                            tbs=np.int64(tbs), G=np.int64(nb), tb=np.packbits(tb.astype(np.uint8)), F=F,
                            noise_var=np.float64(rx.noiseVar), llr=llr, eq_sample=eq.grid[:, ::3, ::7],
                            hest_sample=hest[::3, ::7], decoded=np.packbits(np.uint8(out_tb)), crc=np.asarray(crc, bool),
                            t_off=np.int64(ch.getTimingOffset()), max_delay=np.int64(ch.getMaxDelay()))
        print(name, 'TBS', tbs, 'G', nb, 'crc', np.asarray(crc), 'bit errors', int(np.abs(out_tb[:-24] - tb).sum()))

    e2e.one = one
    dm1 = dict(configType=1, additionalPos=1)
    # BASELINE cfg1: 25 PRB @15 kHz, QPSK, 1 layer, TDL-A SISO, BG2 R=0.3, 5 it, time domain, LS
    one('cfg1_tdl_siso', 123, 25, 15, 'QPSK', 1, 0.3, 2, ('tdl', 'A', 30, 5, 1, 1), dm1, 5, 4.0, False, False, 0)
    # MIMO CDL-C 4x2, 2 layers 16QAM, TD + LS, slot 3
    one('cdl_mimo_td_ls', 321, 25, 15, '16QAM', 2, 490 / 1024, 1, ('cdl', 'C', 300, 5, [1, 2], [1, 1]), dm1, 10, 24.0, False, False, 3)
    # same channel family, frequency domain + perfect CSI (the BLER notebook's default path), 4x4 4 layers 64QAM
    one('cdl_mimo_fd_perfect', 55, 24, 30, '64QAM', 4, 0.5, 1, ('cdl', 'C', 100, 5, [1, 2], [1, 2]), dm1, 10, 26.0, True, True, 1)
    # a failing slot (low SNR) -- CRC failures must match too
    one('cdl_fail_td_ls', 77, 25, 15, '16QAM', 2, 0.6, 1, ('cdl', 'D', 100, 30, [1, 2], [1, 1]), dm1, 8, 6.0, False, False, 0)
    e2e_baseline_configs()


def e2e_baseline_configs():
    """BASELINE.json cfg2 at the size the reference can build it (106 PRB needs 15 kHz) and cfg3's link (4 layers, 256-QAM,
    4x4 CDL-D, R = 0.75) at 24 PRB @30 kHz: with the DMRS-LS estimate every code block fails at ANY SNR in the reference
    itself (CDL-D's LOS + 300 ns spread against a linear interpolation between DMRS subcarriers at 256-QAM), with perfect
    CSI (frequency-domain channel, the BLER notebook's default path) every block passes -- both behaviours are fixtures."""
    one = e2e.one
    dm1 = dict(configType=1, additionalPos=1)
    one('cfg2_cdl_c_2x2', 2024, 106, 15, '64QAM', 2, 666 / 1024, 1, ('cdl', 'C', 300, 5, [1, 1], [1, 1]), dm1, 20, 22.0, False, False, 1)
    one('cfg3_cdl_d_4x4_ls', 31, 24, 30, '256QAM', 4, 0.75, 1, ('cdl', 'D', 300, 5, [1, 2], [1, 2]), dm1, 20, 45.0, False, False, 1)
    one('cfg3_cdl_d_4x4_perfect', 31, 24, 30, '256QAM', 4, 0.75, 1, ('cdl', 'D', 300, 5, [1, 2], [1, 2]), dm1, 20, 45.0, True, True, 1)


def e2e_2cw():
    """A two-codeword slot (6 layers = 3 + 3, different modulation and rate per codeword, 8x8 CDL-C, double-symbol DMRS for
    6 ports): frequency-domain channel + noise, DMRS-LS estimate, MMSE, per-codeword LLRs / decode / CRC."""
    seed, numRbs, spacing, layers = 31, 24, 30, 6
    mods, rates, numIter, snr = ['16QAM', '64QAM'], [0.45, 0.55], 10, 28.0
    dm = dict(configType=1, additionalPos=1, symbols=2)
    nr.random.setSeed(seed)
    car = nr.Carrier(numRbs=numRbs, spacing=spacing)
    bwp = car.curBwp
    p = nr.PDSCH(bwp, numLayers=layers, nID=car.cellId, modulation=mods)
    p.setDMRS(**dm)
    ch = nr.CdlChannel(bwp, 'C', delaySpread=100, carrierFreq=4e9, dopplerShift=5,
                       txAntenna=nr.AntennaPanel([2, 2], polarization='x'), rxAntenna=nr.AntennaPanel([2, 2], polarization='x'))
    g = p.getGrid()
    tbs = p.getTxBlockSize(rates)
    nb = p.getBitSizes(g)
    cw_layers = [layers // 2, layers - layers // 2]
    encs = [nr.LdpcEncoder(baseGraphNo=1, modulation=mods[c], txLayers=cw_layers[c], targetRate=rates[c]) for c in range(2)]
    tbl = [nr.random.bits(tbs[c]) for c in range(2)]
    rm = [encs[c].getRateMatchedCodeBlocks(tbl[c], nb[c]) for c in range(2)]
    p.populateGrid(g, rm)
    idx = p.getReIndexes(g, "PDSCH")
    H = ch.getChannelMatrix()
    F = p.getPrecodingMatrix(H)
    rx = g.precode(F).applyChannel(H).addNoise(snrDb=snr, useRxPower=True)
    hest = rx.estimateChannelLS(p.dmrs, polarInt=False, kernel='linear')[0]
    eq, sc = rx.equalize(hest)
    llrs = p.getLLRsFromGrid(eq, idx, sc)
    out = dict(cfg=np.array(repr(dict(seed=seed, numRbs=numRbs, spacing=spacing, layers=layers, mods=mods, rates=rates, dm=dm,
                                      numIter=numIter, snr=snr))),
               tbs=np.int64(tbs), G=np.int64(nb), F=F, noise_var=np.float64(rx.noiseVar), eq_sample=eq.grid[:, ::3, ::5],
               hest_sample=hest[::3, ::5])
    for c in range(2):
        dec = encs[c].getDecoder()
        db = dec.decode(dec.recoverRate(llrs[c], tbs[c]), numIter=numIter)
        tb_out, crc = dec.checkCrcAndMerge(db)
        out[f'tb{c}'] = np.packbits(tbl[c].astype(np.uint8))
        out[f'llr{c}'] = llrs[c]
        out[f'decoded{c}'] = np.packbits(np.uint8(tb_out))
        out[f'crc{c}'] = np.asarray(crc, bool)
        print('2cw', c, 'TBS', tbs[c], 'G', nb[c], 'crc', np.asarray(crc), 'bit errors', int(np.abs(tb_out[:-24] - tbl[c]).sum()))
    np.savez_compressed(os.path.join(GOLD, 'e2e_2cw.npz'), **out)


def harq_loop():
    """Playground/HARQ/Harq.ipynb cell 7 with both random streams seeded (60 transmissions, Eb/No 0.5 dB)."""
    from neoradium.utils import toLinear
    mod, rate = "16QAM", 490 / 1024
    enc = nr.LdpcEncoder(baseGraphNo=1, modulation=mod, txLayers=1, targetRate=rate)
    harq = nr.HarqEntity(enc, "IR", 16)
    snr = toLinear(0.5 + 10 * np.log10(enc.qm * rate))
    std = np.sqrt(1 / snr)
    nr.random.setSeed(123)
    rangen = nr.random.getGenerator(456)
    modem = nr.Modem(mod)
    errs = []
    for t in range(60):
        tbs = [nr.random.bits(10000) if harq.needNewData[0] else None]
        rm = harq.getRateMatchedCodeBlocks(tbs)
        y = modem.modulate(rm[0])
        y = y + rangen.awgn(y.shape, std)
        _, be = harq.decodeLLRs([modem.getLLRsFromSymbols(y, std ** 2)], [10000])
        errs.append(be[0])
        harq.goNext()
    np.savez_compressed(os.path.join(GOLD, 'harq_loop.npz'), noise_std=np.float64(std), block_errors=np.int64(errs),
                        txBlocks=harq.txBlocks, rxBlocks=harq.rxBlocks, txBits=harq.txBits, rxBits=harq.rxBits,
                        timeouts=np.int64(harq.numTimeouts), throughput=np.float64(harq.throughput))
    print('harq loop: txBlocks', harq.txBlocks, 'rxBlocks', harq.rxBlocks, 'throughput %.2f' % harq.throughput)


def snr_walks():
    out = {}
    rng = np.random.default_rng(5)
    walks = []
    for t in range(40):
        mid, width, snr0, step = rng.uniform(-5, 20), rng.uniform(0.2, 2.5), int(rng.integers(-5, 20)), float(rng.choice([0.2, 0.5, 1.0]))
        s = nr.SnrScheduler(snr0, step)
        seq = []
        for snr in s:
            v = 100 / (1 + np.exp(4 * (snr - mid) / width))
            v = 100.0 if v > 99.7 else (0.0 if v < 0.3 else float(np.round(v, 1)))
            seq.append((snr, v))
            s.setData(v)
        walks.append((mid, width, snr0, step, len(seq)))
        out[f'walk{t}'] = np.float64(seq)
    out['walk_params'] = np.float64(walks)
    np.savez_compressed(os.path.join(GOLD, 'snr_walks.npz'), **out)


def polar():
    """Polar DCI / PBCH / UCI chains (BASELINE cfg4).  Per case: payload, segmented blocks, coded and rate-matched
    bits, noisy LLRs, rate-recovered LLRs, the decoder's output and the SCL list (message bits + path costs)."""
    from neoradium.polar import PolarEncoder, PolarDecoder, SclDecoder
    rng = np.random.default_rng(2024)
    cases = [('dci', 30, 120), ('dci', 40, 216), ('dci', 60, 432), ('dci', 100, 200), ('dci', 12, 108),
             ('dci', 70, 108), ('dci', 140, 432), ('pbch', 32, 432), ('uci', 12, 60), ('uci', 19, 100),
             ('uci', 20, 80), ('uci', 64, 150), ('uci', 200, 600), ('uci', 400, 1200), ('uci', 401, 1300)]
    out = {'cases': np.array(['%s,%d,%d' % c for c in cases])}
    for ci, (typ, A, E) in enumerate(cases):
        enc, dec = PolarEncoder(A, E, typ), PolarDecoder(A, E, typ, sclListSize=8)
        tb = rng.integers(0, 2, A).astype(np.int8)
        cbs = enc.doSegmentation(tb)
        coded = enc.encode(cbs)
        rm = enc.rateMatch(coded)
        p = f'c{ci}_'
        out.update({p + 'tb': tb, p + 'cbs': cbs, p + 'coded': coded, p + 'rm': rm,
                    p + 'params': np.int64([enc.codeBlockSize, enc.polarCodeSize, enc.rateMatchedBlockLen, enc.nPC]),
                    p + 'msgBits': np.int32(enc.msgBits), p + 'frozenBits': np.int32(enc.frozenBits),
                    p + 'pcBits': np.int32(enc.pcBits)})
        for si, snr in enumerate((-1.0, 3.0)):
            sig = 10 ** (-snr / 20)
            llr = 2 * (1 - 2.0 * rm + sig * rng.standard_normal(rm.shape)) / sig ** 2
            rr = dec.recoverRate(llr)
            bits, nerr = dec.decode(rr)
            q = p + f's{si}_'
            out.update({q + 'llr': llr, q + 'rr': rr, q + 'bits': bits, q + 'nerr': np.int64(nerr)})
            cands, costs = [], []
            for row in np.clip(rr, -20, 20):
                sd = SclDecoder(dec.frozenBits, 8)
                u = sd.decode(row)
                m = u[:, dec.msgBits]
                if dec.iIL:
                    m = m[:, dec.inInterleaveIndexes]
                cands.append(m)
                costs.append(sd.pathCosts.copy())
            out.update({q + 'cands': np.int8(cands), q + 'costs': np.float64(costs)})
    np.savez_compressed(os.path.join(GOLD, 'polar.npz'), **out)


def snr_anchors():
    """Playground/Others/SnrCalculations.ipynb cells 1-3 (52 PRB @30 kHz, 16QAM, 1 layer, seed 123, 0 dB): the published
    Waveform.getNoiseStd / Grid.getNoiseStd values, the seed-chain anchors of SURVEY 8c (bits sha, rate-matched sum, DMRS sha), and
    addNoise(snrDb=..., useRxPower=False) -- the MATLAB convention of waveform.py:142 / grid.py:1046 -- on the grid and the waveform."""
    import hashlib
    from neoradium.utils import toLinear
    snr = toLinear(0)
    carrier = nr.Carrier(numRbs=52, spacing=30)
    bwp = carrier.curBwp
    n_r, n_t = 2, 2
    pdsch = nr.PDSCH(bwp, interleavingBundleSize=0, numLayers=1, modulation='16QAM', nID=carrier.cellId)
    pdsch.setDMRS(prgSize=0, configType=2, additionalPos=2)
    enc = nr.LdpcEncoder(baseGraphNo=1, modulation=pdsch.modems[0].modulation, txLayers=pdsch.numLayers, targetRate=490 / 1024)
    nr.random.setSeed(123)
    grid = pdsch.getGrid()
    tbs = pdsch.getTxBlockSize(490 / 1024)
    tb = nr.random.bits(tbs[0])
    nb = pdsch.getBitSizes(grid)
    rm = enc.getRateMatchedCodeBlocks(tb, nb[0])
    dmrs_vals = grid.grid[grid.reTypeIds == grid.retNameToId["DMRS"]]
    pdsch.populateGrid(grid, rm)
    precoder = np.ones((n_t, pdsch.numLayers)) / np.sqrt(pdsch.numLayers)
    tx = grid.precode(precoder).ofdmModulate()
    rxw = nr.Waveform(tx.waveform / np.sqrt(n_r))
    rxg = rxw.ofdmDemodulate(bwp)
    std_t, std_f = rxw.getNoiseStd(snr, bwp), rxg.getNoiseStd(snr)
    print('getNoiseStd time / freq', repr(std_t), repr(std_f))
    # useRxPower=False on both containers (the generator continues from where the transport block left it)
    ng = rxg.addNoise(snrDb=3.0, useRxPower=False)
    nw = rxw.addNoise(snrDb=3.0, bwp=bwp, useRxPower=False)
    nw2 = rxw.addNoise(snrDb=3.0, nFFT=bwp.nFFT)              # (default useRxPower, FFT size given directly)
    np.savez_compressed(os.path.join(GOLD, 'snr.npz'), tbs=np.int64(tbs), G=np.int64(nb), tb_sha=np.array(hashlib.sha256(np.uint8(tb).tobytes()).hexdigest()[:16]),
                        rm_sum=np.int64(rm.sum()), dmrs_sha=np.array(hashlib.sha256(np.complex128(dmrs_vals).tobytes()).hexdigest()[:16]),
                        dmrs_sample=np.complex128(dmrs_vals[:64]), noise_std_time=np.float64(std_t), noise_std_freq=np.float64(std_f),
                        grid_noise_var=np.float64(ng.noiseVar), grid_noisy_sample=ng.grid[:, ::3, ::41],
                        wave_noise_var=np.float64(nw.noiseVar), wave_noisy_sample=nw.waveform[:, ::257],
                        wave2_noise_var=np.float64(nw2.noiseVar), wave2_noisy_sample=nw2.waveform[:, ::257])
    print('tb sha', hashlib.sha256(np.uint8(tb).tobytes()).hexdigest()[:16], 'rm sum', int(rm.sum()), 'dmrs sha',
          hashlib.sha256(np.complex128(dmrs_vals).tobytes()).hexdigest()[:16], 'noiseVar', ng.noiseVar, nw.noiseVar)


def bler_notebook(snrs=(5.8, 5.6, 5.4), num_slots=200):
    """Playground/PDSCH/PDSCH-BLER.ipynb code cell 2, "Perfect" channel estimation (51 PRB @30 kHz, 16QAM, 2 layers, CDL-C 300 ns 16x4,
    BG1 R = 490/1024, 20 iterations, frequency domain, seed 123 per SNR point): the reference itself, slot by slot.  Stored per SNR
    point: the CRC verdicts of every code block and the bit errors of every slot (published totals: 5.8 dB 2 / 800 blocks,
    5.6 dB 124 / 800, 5.4 dB 544 / 800); once (the channel, the transport blocks and the standard-normal draws are the same at every
    point): the LAPACK precoder of every slot -- LAPACK fixes a singular vector only up to a unit phase, so the precoder is DATA for
    whoever replays the table."""
    import time
    carrier = nr.Carrier(numRbs=51, spacing=30)
    bwp = carrier.curBwp
    pdsch = nr.PDSCH(bwp, interleavingBundleSize=0, numLayers=2, nID=carrier.cellId, modulation="16QAM")
    pdsch.setDMRS(prgSize=0, configType=2, additionalPos=2)
    rate = 490 / 1024
    enc = nr.LdpcEncoder(baseGraphNo=1, modulation=pdsch.modems[0].modulation, txLayers=pdsch.numLayers, targetRate=rate)
    dec = enc.getDecoder()
    out = dict(snrs=np.float64(snrs), num_slots=np.int64(num_slots))
    F_all = None
    for snr in snrs:
        nr.random.setSeed(123)
        t0 = time.time()
        carrier.slotNo = 0
        ch = nr.CdlChannel(bwp, 'C', delaySpread=300, carrierFreq=4e9, dopplerShift=5,
                           txAntenna=nr.AntennaPanel([2, 4], polarization="x"), rxAntenna=nr.AntennaPanel([1, 2], polarization="x"))
        crcs, biterrs, Fs, nvs = [], [], [], []
        for s in range(num_slots):
            grid = pdsch.getGrid()
            tbs = pdsch.getTxBlockSize(rate)
            tb = nr.random.bits(tbs[0])
            nb = pdsch.getBitSizes(grid)
            rm = enc.getRateMatchedCodeBlocks(tb, nb[0])
            if s == 0 and F_all is None:       # the seed-chain anchors of SURVEY 8c (bits(30216) sha, rate-matched sum, DMRS sha)
                import hashlib
                dm = grid.grid[grid.reTypeIds == grid.retNameToId["DMRS"]]
                out['tb_sha'] = np.array(hashlib.sha256(np.uint8(tb).tobytes()).hexdigest()[:16])
                out['rm_sum'], out['G'] = np.int64(rm.sum()), np.int64(nb)
                out['dmrs_sha'] = np.array(hashlib.sha256(np.complex128(dm).tobytes()).hexdigest()[:16])
                out['dmrs_sample'] = np.complex128(dm[:64])
                print('anchors: tb sha', out['tb_sha'], 'rm sum', int(rm.sum()), 'G', nb, 'dmrs sha', out['dmrs_sha'],
                      '| int8 tb sha', hashlib.sha256(np.int8(tb).tobytes()).hexdigest()[:16],
                      '| pilots sha', hashlib.sha256(np.complex128(pdsch.dmrs.getPilots()[0] if hasattr(pdsch.dmrs, 'getPilots') else dm).tobytes()).hexdigest()[:16], flush=True)
            pdsch.populateGrid(grid, rm)
            idx = pdsch.getReIndexes(grid, "PDSCH")
            H = ch.getChannelMatrix()
            F = pdsch.getPrecodingMatrix(H)
            rx = grid.precode(F).applyChannel(H).addNoise(snrDb=snr, useRxPower=True)
            eq, sc = rx.equalize(H @ F[None, ...])
            llr = pdsch.getLLRsFromGrid(eq, idx, sc)
            db = dec.decode(dec.recoverRate(llr[0], tbs[0]), numIter=20)
            o, crc = dec.checkCrcAndMerge(db)
            crcs.append(np.asarray(crc, bool))
            biterrs.append(int(np.abs(o[:-24] - tb).sum()))
            Fs.append(F)
            nvs.append(rx.noiseVar)
            ch.goNext()
        crcs = np.stack(crcs)
        print(f'{snr} dB: block errors {int((~crcs).sum())} / {crcs.size}, bit errors {sum(biterrs)}, {time.time() - t0:.1f} s', flush=True)
        key = ('%.1f' % snr).replace('.', '_')
        out['crc_' + key] = crcs
        out['biterr_' + key] = np.int64(biterrs)
        out['noise_var_' + key] = np.float64(nvs)
        if F_all is None:
            F_all = np.stack(Fs)
            out['F'] = F_all
            out['tbs'] = np.int64(tbs)
        else:
            assert np.array_equal(F_all, np.stack(Fs))
    np.savez_compressed(os.path.join(GOLD, 'bler_notebook.npz'), **out)


if __name__ == '__main__':
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 1:                                 # regenerate selected fixture files only
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    copy_matlab()
    matlab_cdl()
    coding()
    coding_lbrm()
    phy()
    host()
    ptrs()
    csirs()
    csifeedback()
    csifeedback_multipanel()
    ofdm_options()
    channels()
    channels_xiao()
    snr_walks()
    harq_loop()
    snr_anchors()
    bler_notebook()
    polar()
    chest()
    prg()
    e2e_2cw()
    e2e()
    print('fixtures written to', GOLD)
