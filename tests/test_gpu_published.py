"""GPU: the numbers the reference PUBLISHES in its stored notebook outputs, turned into assertions on the class surface and
on the batched engine: the PDSCH-BLER notebook's table, SnrCalculations.ipynb's noise levels, the seed-chain anchors, the
MATLAB-convention noise of `addNoise(useRxPower=False)`, CRC polynomial '16'."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def test_snr_calculations_notebook_and_matlab_convention_noise(dev):
    """Playground/Others/SnrCalculations.ipynb cells 1-3 at seed 123, 52 PRB @30 kHz, 0 dB on the class surface: Waveform.getNoiseStd
    (waveform.py:119-142) and Grid.getNoiseStd (grid.py:1040-1046) equal the values the notebook prints; then
    addNoise(snrDb=3, useRxPower=False) -- sigma^2 = 1 / (snr Nr) on the grid (grid.py:1182-1185), 1 / (snr Nr nFFT) on the waveform
    (waveform.py:279-292) -- reproduces the reference's noisy samples from the same generator state."""
    import neoradium_amd as nr
    g = np.load(os.path.join(GOLD, 'snr.npz'))
    snr = nr.utils.toLinear(0)
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
    assert tbs == g['tbs'].tolist() and nb == g['G'].tolist() and _sha(np.uint8(tb)) == str(g['tb_sha'])
    rm = enc.getRateMatchedCodeBlocks(tb, nb[0])
    assert int(rm.sum()) == int(g['rm_sum'])
    dm = grid.grid[grid.reTypeIds == grid.retNameToId["DMRS"]]
    assert np.abs(dm[:64] - g['dmrs_sample']).max() < 1e-15 and _sha(np.complex128(dm)) == str(g['dmrs_sha'])
    pdsch.populateGrid(grid, rm)
    precoder = np.ones((n_t, pdsch.numLayers)) / np.sqrt(pdsch.numLayers)
    tx = grid.precode(precoder).ofdmModulate()
    rxw = nr.Waveform(tx.waveform / np.sqrt(n_r))
    rxg = rxw.ofdmDemodulate(bwp)
    std_t, std_f = rxw.getNoiseStd(snr, bwp), rxg.getNoiseStd(snr)
    # the notebook's own printed values (cells 2 and 3)
    assert abs(std_t - 0.0220441589451537) <= 1e-12 * 0.0220441589451537
    assert abs(std_f - 0.705375297931587) <= 1e-12 * 0.705375297931587
    assert abs(std_t - float(g['noise_std_time'])) <= 1e-12 * std_t and abs(std_f - float(g['noise_std_freq'])) <= 1e-12 * std_f
    # useRxPower=False (the default of both addNoise methods): the MATLAB convention
    ng = rxg.addNoise(snrDb=3.0, useRxPower=False)
    nw = rxw.addNoise(snrDb=3.0, bwp=bwp, useRxPower=False)
    nw2 = rxw.addNoise(snrDb=3.0, nFFT=bwp.nFFT)
    s3 = nr.utils.toLinear(3.0)
    assert abs(ng.noiseVar - 1 / (s3 * n_r)) < 1e-15 and abs(ng.noiseVar - float(g['grid_noise_var'])) <= 1e-15
    assert abs(nw.noiseVar - 1 / (s3 * n_r * bwp.nFFT)) < 1e-18 and abs(nw.noiseVar - float(g['wave_noise_var'])) <= 1e-18
    assert abs(nw2.noiseVar - float(g['wave2_noise_var'])) <= 1e-18
    for got, ref in ((ng.grid[:, ::3, ::41], g['grid_noisy_sample']), (nw.waveform[:, ::257], g['wave_noisy_sample']),
                     (nw2.waveform[:, ::257], g['wave2_noisy_sample'])):
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()


def test_seed_chain_anchors_of_the_bler_notebook(dev):
    """random.setSeed(123); bits(30216) -> sha 86601f2723df4cb3; its rate-matched 63648 bits (BG1, 16QAM, 2 layers, R =
    490/1024) sum to 31786; the DMRS values of the notebook's grid (51 PRB @30 kHz, 2 layers, configType 2, additionalPos 2) -> sha
    0e2af13bb0ad828f.  The LDPC chain runs on the GPU through the class surface."""
    import neoradium_amd as nr
    nr.random.setSeed(123)
    tb = nr.random.bits(30216)
    assert _sha(np.uint8(tb)) == '86601f2723df4cb3'
    enc = nr.LdpcEncoder(baseGraphNo=1, modulation='16QAM', txLayers=2, targetRate=490 / 1024)
    rm = enc.getRateMatchedCodeBlocks(tb, 63648)
    assert rm.shape == (63648,) and int(rm.sum()) == 31786
    carrier = nr.Carrier(numRbs=51, spacing=30)
    pdsch = nr.PDSCH(carrier.curBwp, interleavingBundleSize=0, numLayers=2, nID=carrier.cellId, modulation="16QAM")
    pdsch.setDMRS(prgSize=0, configType=2, additionalPos=2)
    grid = pdsch.getGrid()
    assert pdsch.getTxBlockSize(490 / 1024)[0] == 30216 and pdsch.getBitSizes(grid)[0] == 63648
    assert _sha(np.complex128(grid.grid[grid.reTypeIds == grid.retNameToId["DMRS"]])) == '0e2af13bb0ad828f'


@pytest.mark.parametrize("poly", ['16', '11', '6', '24A', '24B', '24C'])
def test_every_crc_polynomial_on_the_gpu(dev, poly):
    """chancodebase.py:37-128 getCrc / appendCrc / checkCrc for EVERY polynomial of TS 38.212 5.1 -- '16' (transport blocks of at most
    3824 bits, ldpc.py:981-1005) included -- on the device kernel against the oracle's bit-serial division, and through the class
    surface: a transport block short enough for CRC16 segments, decodes and checks."""
    import torch
    import neoradium_amd as nr
    from neoradium_amd import ops
    from oracle import coding as oc
    rng = np.random.default_rng(len(poly) * 7 + 1)
    for n in (1, 17, 100, 3824, 8448):
        bits = rng.integers(0, 2, (3, n)).astype(np.uint8)
        got = ops.crc(torch.from_numpy(bits).to(dev), poly).cpu().numpy()
        for i in range(3):
            assert np.array_equal(got[i], oc.crc_bits(bits[i], poly)), (poly, n)
    if poly == '16':
        enc = nr.LdpcEncoder(baseGraphNo=2, modulation='QPSK', txLayers=1, targetRate=0.4)
        tb = rng.integers(0, 2, 1000).astype(np.int8)
        with_crc = enc.appendCrc(tb, '16')
        assert len(with_crc) == 1016 and enc.checkCrc(with_crc, '16') and np.array_equal(np.asarray(with_crc[-16:]), oc.crc_bits(np.uint8(tb), '16'))
        flipped = with_crc.copy()
        flipped[5] ^= 1
        assert not enc.checkCrc(flipped, '16')


def test_pdsch_bler_notebook_table(dev):
    """Playground/PDSCH/PDSCH-BLER.ipynb code cell 2, "Perfect" channel estimation, as stored in the notebook: 800 code blocks per SNR
    point (200 slots, seed 123 per point), block errors 5.8 dB 2, 5.6 dB 124, 5.4 dB 544.  The batched engine replays the table in
    parity mode: class-surface construction (the CDL channel draws its phases from the generator first), then per slot the host
    PCG64(123) stream in the reference's draw order (transport block, then the noise's standard normals), the reference's LAPACK
    precoders as data (tests/golden/bler_notebook.npz, tools/gen_golden.py bler_notebook).  Asserted: the published block-error
    counts within +-3, the reference's per-block verdicts (>= 99 % of 800 identical at the waterfall point), and -- 5.6 dB -- the CPU
    oracle on the same inputs: every verdict but at most two (float64 last bits on blocks that never converge)."""
    import torch
    import neoradium_amd as nr
    from neoradium_amd._dev import D
    from oracle import link as olink
    g = np.load(os.path.join(GOLD, 'bler_notebook.npz'))
    n = int(g['num_slots'])
    published = {5.8: 2, 5.6: 124, 5.4: 544}
    carrier = nr.Carrier(numRbs=51, spacing=30)
    bwp = carrier.curBwp
    pdsch = nr.PDSCH(bwp, interleavingBundleSize=0, numLayers=2, nID=carrier.cellId, modulation="16QAM")
    pdsch.setDMRS(prgSize=0, configType=2, additionalPos=2)
    nr.random.setSeed(123)
    ch = nr.CdlChannel(bwp, 'C', delaySpread=300, carrierFreq=4e9, dopplerShift=5,
                       txAntenna=nr.AntennaPanel([2, 4], polarization="x"), rxAntenna=nr.AntennaPanel([1, 2], polarization="x"))
    link = nr.PdschLink(pdsch, ch, 490 / 1024, baseGraphNo=1, numIter=20, freqDomain=True, chanEst="Perfect", decoder="f64")
    assert link.tbs == int(g['tbs'][0]) == 30216 and (link.nr, link.nt, link.nl) == (4, 16, 2)
    tb = np.empty((n, link.tbs), dtype=np.uint8)
    zc = np.empty((n, link.nr, link.L, link.K), dtype=np.complex128)
    for s in range(n):                                     # the notebook's draw order, slot by slot
        tb[s] = nr.random.bits(link.tbs)
        z = nr.random.normal(0, 1, (link.nr, link.L, link.K, 2))
        zc[s] = z[..., 0] + 1j * z[..., 1]
    assert _sha(tb[0]) == str(g['tb_sha'])
    F = g['F']
    tb_d, zc_d, F_d = torch.from_numpy(tb), D(zc), D(F)
    for snr in g['snrs']:
        key = ('%.1f' % snr).replace('.', '_')
        ref_crc = g['crc_' + key]
        assert int((~ref_crc).sum()) == published[round(float(snr), 1)]          # the fixture IS the published table
        counters, det = link.run(0, n, float(snr), tb_bits=tb_d, noise=zc_d, precoder=F_d, details="verdicts")
        ok = det[0][1]['cb_ok'].cpu().numpy().astype(bool)
        errs = int((~ok).sum())
        assert abs(errs - published[round(float(snr), 1)]) <= 3, (snr, errs)
        assert int(counters[0]) == errs and int(counters[1]) == 4 * n
        assert (ok == ref_crc).mean() >= 0.99, (snr, int((ok != ref_crc).sum()))
        if abs(float(snr) - 5.6) < 1e-9:
            st = olink.static_from_link(link, slots=range(n))
            jobs = [(st, s, float(snr), tb[s].astype(np.int8), zc[s], F[s]) for s in range(n)]
            refs = olink.run_slots_parallel(jobs, max(1, min(16, os.cpu_count() or 1)))
            cpu_ok = np.stack([r['crc'] for r in refs])
            assert abs(int((~cpu_ok).sum()) - 124) <= 3
            assert int((cpu_ok != ok).sum()) <= 2, int((cpu_ok != ok).sum())
