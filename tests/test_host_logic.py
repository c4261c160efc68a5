"""CPU: host-side logic of neoradium_amd (no GPU compute) against fixtures produced by the reference, plus the
C-ABI load/export check (every symbol include/nrx.h declares is exported by libnrx.so and bound by ctypes)."""
import json
import os
import re

import numpy as np
import pytest

import neoradium_amd as ma
from neoradium_amd import _lib

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ROOT = os.path.dirname(GOLD.rstrip('/').rsplit('/tests', 1)[0] + '/x')


def test_abi_exports_match_header():
    hdr = open(os.path.join(os.path.dirname(GOLD), '..', 'include', 'nrx.h')).read()
    declared = set(re.findall(r'\b(nrx_[a-z0-9_]+)\s*\(', hdr))
    assert declared, "no declarations parsed"
    lib = _lib.lib()                                   # raises if the .so is missing / a symbol is absent
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name)
    assert lib.nrx_version() >= 100


def test_host_only_entry_points():
    cfg = _lib.ldpc_config(1, 606504 + 24)            # the metric configuration
    assert (cfg.C, cfg.Zc, cfg.iLS, cfg.K, cfg.N, cfg.F) == (72, 384, 1, 8448, 25344, 0)
    g = np.load(os.path.join(GOLD, 'coding.npz'))
    for bg, B, C, Zc, iLS, K in g['seg_anchors']:
        c = _lib.ldpc_config(int(bg), int(B))
        assert (c.C, c.Zc, c.iLS, c.K) == (C, Zc, iLS, K)
    assert _lib.ldpc_cb_lens(943488, 72, 4, 6) == [13104] * 72
    assert _lib.ldpc_cb_lens(100, 3, 2, 2) == [32, 32, 36]
    with pytest.raises(ValueError):
        _lib.ldpc_config(3, 100)
    fx = np.load(os.path.join(GOLD, 'phy.npz'))
    for ci in (32769, 1, 123456789):
        assert np.array_equal(_lib.gold_sequence(ci, 2000), fx[f'gold_{ci}'])
    # rows whose extension LLRs the decoder entries read for a request (what the rate-recovering demapper must initialise): the row
    # count of the instantiation that runs at (BG1, Zc 384), all rows elsewhere; never fewer than asked for
    from neoradium_amd import ops
    assert [ops.ldpc_rows_read(cfg, r, True) for r in (4, 13, 14, 15, 16, 17, 22, 23, 31, 32, 46, None)] == \
        [13, 13, 15, 15, 16, 22, 22, 31, 31, 46, 46, 46]
    assert [ops.ldpc_rows_read(cfg, r, False) for r in (4, 13, 14, 15, 16, 31, 32, None)] == [13, 13, 15, 15, 31, 31, 46, 46]
    c2 = _lib.ldpc_config(2, 2408 + 16)
    assert c2.Zc == 256 and ops.ldpc_rows_read(c2, 21, False) == 42 and ops.ldpc_rows_read(_lib.ldpc_config(1, 20000), 13, True) == 46


def _build(c):
    kw = dict(numRbs=c['numRbs'], spacing=c['spacing'])
    if 'startRb' in c:
        kw['startRb'] = c['startRb']
    car = ma.Carrier(**kw)
    p = ma.PDSCH(car.curBwp, numLayers=c['layers'], modulation=c['mod'], **c['pk'])
    p.setDMRS(**c['dm'])
    return car, p


def test_carrier_pdsch_dmrs_vs_reference():
    g = np.load(os.path.join(GOLD, 'host.npz'))
    cfgs = json.loads(str(g['cfgs']))
    for i, c in enumerate(cfgs):
        car, p = _build(c)
        bwp = car.curBwp
        assert [bwp.nFFT] + bwp.symbolLens.tolist() == g[f'h{i}_numerology'].tolist()
        for slot in (0, 7):
            car.slotNo = slot
            grid = p.getGrid()
            assert np.array_equal(grid.reTypeIds, g[f'h{i}_s{slot}_types'])
            idx = np.nonzero(grid.reTypeIds == grid.retNameToId['DMRS'])
            assert np.abs(grid.grid[idx] - g[f'h{i}_s{slot}_dmrs']).max() < 1e-15
            assert np.array_equal(np.int32(p.dataIndices), g[f'h{i}_s{slot}_data'])
            assert np.array_equal(np.int32(p.getLayerMapIndexes(p.dataIndices)[0]), g[f'h{i}_s{slot}_lm'])
            # the pilot table the estimator kernel consumes = the DMRS REs of the grid
            pil, ks, ds = p.dmrs.getPilots()
            for port in range(len(p.portSet)):
                for li, l in enumerate(ds):
                    assert np.array_equal(grid.grid[port, l, ks[port]], pil[port, li])
        assert [p.getTxBlockSize(r)[0] for r in (0.2, 0.3, 0.5, 666 / 1024, 0.75, 0.92)] == g[f'h{i}_tbs'].tolist()
        assert p.getBitSizes(grid) == g[f'h{i}_bits'].tolist()
        assert p.dmrs.dataREs == g[f'h{i}_dataREs'].tolist()


def test_273prb_extension_numerology():
    """Beyond the reference (it raises for numRbs >= nFFT/12): nFFT and sample rate scale together."""
    car = ma.Carrier(numRbs=273, spacing=30)
    b = car.curBwp
    assert (b.nFFT, b.sampleRate, b.getSlotLen(0)) == (4096, 122.88e6, 61440)
    assert b.getCpLens(0).tolist() == [352] + [288] * 13
    p = ma.PDSCH(b, numLayers=4, modulation='64QAM')
    p.setDMRS(configType=1, additionalPos=1)
    g = p.getGrid()
    assert p.getTxBlockSize(666 / 1024) == [606504] and p.getBitSizes(g) == [943488]      # SURVEY 8 metric row
    # inside the reference's range nothing changes
    assert ma.Carrier(numRbs=84, spacing=30).curBwp.nFFT == 1024
    assert ma.Carrier(numRbs=169, spacing=15).curBwp.nFFT == 2048


def test_argument_errors_like_reference():
    car = ma.Carrier(numRbs=25, spacing=15)
    with pytest.raises(ValueError):
        ma.Carrier(numRbs=25, spacing=17)
    with pytest.raises(ValueError):
        ma.PDSCH(car.curBwp, modulation='8PSK')
    with pytest.raises(ValueError):
        ma.PDSCH(car.curBwp, prgSize=3)
    p = ma.PDSCH(car.curBwp, numLayers=2)
    with pytest.raises(ValueError):
        p.setDMRS(configType=3)
    with pytest.raises(ValueError):
        ma.LdpcEncoder(baseGraphNo=3)
    with pytest.raises(ValueError):
        ma.SnrScheduler(0, -1)


def test_channel_static_coefficients_vs_reference(monkeypatch):
    """CDL/TDL: the host-built static ray tensors reproduce the reference's per-slot gains (no GPU involved:
    sum_m A exp(j 2 pi t nu) evaluated here in NumPy; the GPU kernel is checked against the same formula)."""
    from neoradium_amd import channelmodel
    monkeypatch.setattr(channelmodel.ChannelModel, 'prepareForNextSlot', lambda self: None)
    g = np.load(os.path.join(GOLD, 'channels.npz'))
    specs = [('cdl', 'C', dict(delaySpread=300, carrierFreq=4e9, dopplerShift=5), ([1, 2], [1, 2])),
             ('cdl', 'D', dict(delaySpread=100, dopplerShift=50, ueDirAZ=[30, 80]), ([1, 2], [1, 1])),
             ('cdl', 'A', dict(delaySpread=30, dopplerShift=100, angleScaling=([120, 200, 90, 95], [10, 30, 5, 8])), ([2, 2], [1, 1])),
             ('tdl', 'A', dict(delaySpread=30, dopplerShift=5), None),
             ('tdl', 'C', dict(delaySpread=300, dopplerShift=100, txAntennaCount=2, rxAntennaCount=2, mimoCorrelation='Medium'), None),
             ('tdl', 'D', dict(delaySpread=100, dopplerShift=30, txAntennaCount=4, rxAntennaCount=2, mimoCorrelation='High'), None)]
    for i, (kind, prof, kw, ant) in enumerate(specs):
        ma.random.setSeed(100 + i)
        car = ma.Carrier(numRbs=25, spacing=15)
        if kind == 'cdl':
            ch = ma.CdlChannel(car.curBwp, prof, txAntenna=ma.AntennaPanel(ant[0], polarization='x'),
                               rxAntenna=ma.AntennaPanel(ant[1], polarization='x'), **kw)
        else:
            ch = ma.TdlChannel(car.curBwp, prof, **kw)
        A, nu, Alos, nulos = ch.staticCoefficients()
        t = g[f'ch{i}_samples'] / ch.sampleRate
        gains = np.einsum('rtnm,cnm->crtn', A, np.exp(2j * np.pi * t[:, None, None] * nu[None]))
        if Alos is not None:
            los = Alos[None] * np.exp(2j * np.pi * t * nulos)[:, None, None]
            gains = np.concatenate([los[..., None], gains], axis=3)
        gains = gains * ch._normalisation()
        ref = g[f'ch{i}_gains1']
        assert np.abs(gains - ref).max() < 1e-12 * np.abs(ref).max()
        assert np.array_equal(ch.getCoeffMatrix(), g[f'ch{i}_coeff'])
        assert ch.getMaxDelay() == g[f'ch{i}_misc'][1]


def test_snr_scheduler_walks():
    g = np.load(os.path.join(GOLD, 'snr_walks.npz'))
    for t, (mid, width, snr0, step, n) in enumerate(g['walk_params']):
        walk = g[f'walk{t}']
        s = ma.SnrScheduler(int(snr0), float(step))
        seen = []
        for snr in s:
            row = walk[len(seen)]
            assert abs(snr - row[0]) < 1e-9
            seen.append(snr)
            s.setData(float(row[1]))
        assert len(seen) == int(n)


def test_random_stream_is_numpy_pcg64():
    """bits -> awgn draw order reproduces NumPy's PCG64 stream (the reproducibility contract)."""
    ma.random.setSeed(123)
    b = ma.random.bits(30216)
    import hashlib
    assert hashlib.sha256(b.tobytes()).hexdigest()[:16] == '86601f2723df4cb3'       # SURVEY 8c anchor
    ref = np.random.Generator(np.random.PCG64(123))
    assert np.array_equal(b, ref.integers(0, 2, 30216, dtype=np.int8))
    z = ma.random.awgn((2, 3), 0.5)
    zz = ref.normal(0, 0.5 / np.sqrt(2), (2, 3, 2))
    assert np.array_equal(z, zz[..., 0] + 1j * zz[..., 1])


def test_no_cpu_fallback():
    """Product operators refuse to run without a GPU instead of silently falling back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises((RuntimeError, ValueError)):
        ma.Modem('QPSK').modulate(np.int8([0, 1, 1, 0]))
    with pytest.raises((RuntimeError, ValueError)):
        ma.LdpcEncoder().getRateMatchedCodeBlocks(np.zeros(100, dtype=np.int8), 400)


def test_tdl_xiao_coefficients_vs_reference(monkeypatch):
    """TDL sosType='Xiao' (tdl.py:1043-1067): a statistical model that draws new angles and phases for every slot.  The
    host-built ray tensors reproduce the reference's gains of two consecutive preparations (same generator, same draw
    order), SISO Rayleigh and 2x2 correlated with a LOS tap."""
    from neoradium_amd import channelmodel
    monkeypatch.setattr(channelmodel.ChannelModel, 'prepareForNextSlot', lambda self: None)
    g = np.load(os.path.join(GOLD, 'channels_xiao.npz'))
    specs = [('B', dict(delaySpread=100, dopplerShift=70, sosType='Xiao')),
             ('D', dict(delaySpread=30, dopplerShift=20, sosType='Xiao', txAntennaCount=2, rxAntennaCount=2, mimoCorrelation='Medium'))]
    for i, (prof, kw) in enumerate(specs):
        ma.random.setSeed(300 + i)
        car = ma.Carrier(numRbs=25, spacing=15)
        ch = ma.TdlChannel(car.curBwp, prof, **kw)
        for k in range(2):                                  # construction prepares slot 0, the next use prepares slot 1
            A, nu, Alos, nulos = ch.staticCoefficients()
            t = g[f'x{i}_samples{k}'][:-1] / ch.sampleRate
            gains = np.einsum('rtnm,cnm->crtn', A, np.exp(2j * np.pi * t[:, None, None] * nu[None])) * ch._normalisation()
            ref = g[f'x{i}_gains{k}']
            assert Alos is None and np.abs(gains - ref).max() < 1e-11 * np.abs(ref).max()
    with pytest.raises(ValueError):
        ma.TdlChannel(car.curBwp, 'A', sosType='Jakes')


def test_ptrs_vs_reference():
    """PTRS (dmrs.py:554-797): densities from the MCS / bandwidth thresholds or set directly, the PTRS symbol set, the
    inserted values and positions (with VRB interleaving, mapping type B, a partial allocation, DMRS type 2, both EPRE
    ratios), and the data bits that remain -- against the reference, for two slots of the frame."""
    import json
    g = np.load(os.path.join(GOLD, 'ptrs.npz'))
    cfgs = json.loads(str(g['cfgs']))
    for i, c in enumerate(cfgs):
        car = ma.Carrier(numRbs=c['numRbs'], spacing=c['spacing'])
        pt = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c['pt'].items()}     # (JSON turned the tuples into lists)
        p = ma.PDSCH(car.curBwp, numLayers=c['layers'], modulation=c['mod'], **c['pk'])
        p.setDMRS(**c['dm'])
        p.setPTRS(**pt)
        assert [p.dmrs.ptrs.timeDensity, p.dmrs.ptrs.freqDensity] == g[f'p{i}_dens'].tolist()
        assert list(p.dmrs.ptrs.symSet) == g[f'p{i}_syms'].tolist() and p.dmrs.ptrsEnabled
        for slot in (0, 7):
            car.slotNo = slot
            grid = p.getGrid()
            idx = np.nonzero(grid.reTypeIds == grid.retNameToId['PTRS'])
            assert np.array_equal(np.int32(np.stack(idx)), g[f'p{i}_s{slot}_idx'])
            assert np.abs(grid.grid[idx] - g[f'p{i}_s{slot}_val']).max() < 1e-14
            assert [int(v) for v in p.getBitSizes(grid)] == g[f'p{i}_s{slot}_bits'].tolist()
    assert "timeDensity" in repr(p.dmrs.ptrs)
    with pytest.raises(ValueError):
        p.setPTRS(timeDensity=3)
    with pytest.raises(ValueError):
        p.setPTRS(mcsi=[5, 10, 20], iMCS=12, nRBi=(10, 40))          # the reference rejects Python lists here (dmrs.py:640)
    q = ma.PDSCH(car.curBwp)
    with pytest.raises(ValueError):
        q.setPTRS()                                                    # no DMRS yet


def test_csirs_vs_reference():
    """CSI-RS (csirs.py): the row of TS 38.211 Table 7.4.1.5.3-1 inferred from the parameters, the populated values and
    positions of ten configurations (all CDM types, both densities, one- and two-symbol rows, a partial RB range with
    power and scrambling settings, several slot numbers), the reserved map, and which slots a periodic NZP set and a
    semi-persistent ZP set occupy -- against the reference."""
    g = np.load(os.path.join(GOLD, 'csirs.npz'))
    for i, c in enumerate(json.loads(str(g['cfgs']))):
        car = ma.Carrier(numRbs=c['rb'], spacing=c['sp'])
        car.slotNo = c['slot']
        bwp = car.curBwp
        cc = ma.CsiRsConfig(csiType='NZP', bwp=bwp, **c['kw'])
        r = cc.csiRsSetList[0].csiRsList[0]
        assert [r.row] + list(r.ks) + list(r.ls) == g[f'c{i}_row'].tolist()
        grid = bwp.createGrid(cc.numPorts)
        cc.populateGrid(grid)
        idx = np.nonzero(grid.reTypeIds == grid.retNameToId['CSIRS_NZP'])
        assert np.array_equal(np.int32(np.stack(idx)), g[f'c{i}_idx']), i
        assert np.abs(grid.grid[idx] - g[f'c{i}_val']).max() < 1e-14
        assert int((grid.reTypeIds == 0).sum()) == int(g[f'c{i}_untouched'][0])
        g2 = bwp.createGrid(3)
        cc.reserveGridResources(g2)
        assert np.array_equal(np.int32(np.stack(np.nonzero(g2.reTypeIds == g2.retNameToId['CSIRS_NZP']))), g[f'c{i}_res'])
        if cc.numPorts <= 3:
          with pytest.raises(AssertionError):
            cc.populateGrid(g2)                                # NZP values onto REs that are already CSIRS_NZP
    assert "Table Row" in repr(cc) and r.period == 5 and r.startRb == 4
    car = ma.Carrier(numRbs=24, spacing=15)
    bwp = car.curBwp
    zp = ma.CsiRsSet("ZP", bwp, csiRsList=[ma.CsiRs(offset=1, symbols=[1], numPorts=1, freqMap="000001000000", density=0.5)],
                     resourceType='semiPersistent', period=10)
    nz = ma.CsiRsSet("NZP", bwp, csiRsList=[ma.CsiRs(offset=0, symbols=[1], numPorts=1, freqMap="0010", density=3),
                                             ma.CsiRs(offset=3, symbols=[3], numPorts=1, freqMap="000000001000", density=1)],
                     resourceType='periodic', period=5)
    cc = ma.CsiRsConfig([zp, nz])
    counts = []
    for slot in range(10):
        car.slotNo = slot
        grid = bwp.createGrid(1)
        cc.populateGrid(grid)
        counts.append([int((grid.reTypeIds == grid.retNameToId[t]).sum()) for t in ('CSIRS_ZP', 'CSIRS_NZP')])
        if slot in (0, 1, 3):
            assert np.array_equal(grid.reTypeIds, g[f'mix_s{slot}_types'])
            assert np.abs(grid.grid - g[f'mix_s{slot}_grid']).max() < 1e-14
    assert counts == g['mix_counts'].tolist()
    zp.active = False
    car.slotNo = 1
    grid = bwp.createGrid(1)
    cc.populateGrid(grid)
    assert not (grid.reTypeIds == grid.retNameToId['CSIRS_ZP']).any()
    for bad in (dict(numPorts=3), dict(numPorts=4, density=0.5), dict(numPorts=8, cdmSize=3), dict(numPorts=2, freqMap='0110')):
        with pytest.raises((ValueError, KeyError)):
            ma.CsiRsConfig(csiType='NZP', bwp=bwp, **bad)
    with pytest.raises(ValueError):
        ma.CsiRsConfig().populateGrid(grid)


def test_interpolation_tap_tables_vs_scipy():
    """neoradium_amd.interp: the tap tables reproduce the routines the reference calls (utils.py:26-35 interp1d /
    RBFInterpolator with nearest neighbours, grid.py:853-861 in two dimensions) when applied to sample values."""
    from scipy.interpolate import RBFInterpolator, interp1d
    from neoradium_amd.interp import rbf_taps, taps_1d
    rng = np.random.default_rng(11)
    x = np.arange(0.5, 288, 4.0)
    xn = np.arange(288.0)
    v = rng.standard_normal((len(x), 3)) + 1j * rng.standard_normal((len(x), 3))
    for kind in ('nearest', 'quadratic', 'linear'):
        idx, w = taps_1d(x, xn, kind)
        ref = interp1d(x, v, kind=kind, axis=0, fill_value='extrapolate')(xn)
        assert np.abs((w[:, :, None] * v[idx]).sum(1) - ref).max() < 1e-12, kind
    for kind in ('thin_plate_spline', 'multiquadric'):
        idx, w = taps_1d(x, xn, kind, 12, 0.0)
        ref = RBFInterpolator(x[:, None], v, 12, 0.0, kind, 1)(xn[:, None])
        assert idx.shape == (288, 12) and np.abs((w[:, :, None] * v[idx]).sum(1) - ref).max() < 1e-10, kind
    pts = np.float64(np.meshgrid(np.arange(96.), [2., 5., 8., 11.])).reshape(2, -1).T
    qs = np.float64(np.meshgrid(range(96), range(14))).reshape(2, -1).T
    vv = rng.standard_normal((len(pts), 2)) + 1j * rng.standard_normal((len(pts), 2))
    for kern, nb, sm, deg in (('thin_plate_spline', 12, 0.0, None), ('thin_plate_spline', 16, 0.1, None), ('linear', 12, 0.0, 1),
                              ('cubic', 14, 0.0, None)):
        idx, w = rbf_taps(pts, qs, kern, nb, sm, None, deg)
        ref = RBFInterpolator(pts, vv, nb, sm, kern, degree=deg)(qs)
        assert np.abs((w[:, :, None] * vv[idx]).sum(1) - ref).max() < 1e-10, kern
    with pytest.raises(np.linalg.LinAlgError):          # collinear neighbour sets are singular there as well
        rbf_taps(pts, qs, 'thin_plate_spline', 6, 0.0, None, None)
    with pytest.raises(ValueError):
        rbf_taps(pts, qs, 'multiquadric', 12, 0.0, None, None)          # needs epsilon
    with pytest.raises(ValueError):
        taps_1d(x, xn, 'cubic')
