"""GPU: reference-signal based estimation beyond the default -- CSI-RS pilots (csirs.py + grid.py:746-752),
every interpolation kind of estimateChannelLS / estimateChannelLsEx (utils.py:26-42, grid.py:853-866 incl. the 2-D radial
basis path) and the CSI-RS timing estimate (grid.py:592-622), against outputs of the reference (tests/golden/csirs.npz,
tools/gen_golden.py csirs; the fixture keeps every fifth subcarrier of each estimate)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _close(a, b, tol):
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


def test_csirs_channel_estimation_all_kinds():
    import neoradium_amd as ma
    g = np.load(os.path.join(GOLD, 'csirs.npz'))
    for i, c in enumerate(json.loads(str(g['est_cfgs']))):
        car = ma.Carrier(numRbs=c['rb'], spacing=c['sp'])
        bwp = car.curBwp
        cc = ma.CsiRsConfig(csiType='NZP', bwp=bwp, **c['kw'])
        rx = ma.Grid(bwp, numPlanes=c['nr'])
        rx.grid = g[f'e{i}_rx']
        he, nv = rx.estimateChannelLS(cc)
        assert he.shape == (14, 12 * c['rb'], c['nr'], cc.numPorts)
        assert _close(he[:, ::5], g[f'e{i}_lin'], 1e-10), i
        assert abs(nv - g[f'e{i}_lin_nv'][0]) <= 1e-8 * g[f'e{i}_lin_nv'][0]
        he, nv = rx.estimateChannelLS(cc, polarInt=True)
        assert _close(he[:, ::5], g[f'e{i}_pol'], 1e-10), i
        assert abs(nv - g[f'e{i}_pol_nv'][0]) <= 1e-8 * g[f'e{i}_pol_nv'][0]
        if f'e{i}_nearest' not in g:
            continue
        for kern in ('nearest', 'quadratic', 'thin_plate_spline', 'multiquadric'):
            he, nv = rx.estimateChannelLS(cc, kernel=kern)
            assert _close(he[:, ::5], g[f'e{i}_{kern}'], 1e-9), (i, kern)
            assert abs(nv - g[f'e{i}_{kern}_nv'][0]) <= 1e-7 * g[f'e{i}_{kern}_nv'][0], (i, kern)
        if c['kw'].get('cdmSize', 2) <= 2:
            he, nv = rx.estimateChannelLS(cc, meanCdm=False)
            assert _close(he[:, ::5], g[f'e{i}_nomean'], 1e-10), i
            assert abs(nv - g[f'e{i}_nomean_nv'][0]) <= 1e-8 * g[f'e{i}_nomean_nv'][0]
        if i == 3:
            he, nv, hps = rx.estimateChannelLsEx(cc)
            assert _close(he[:, ::5], g['e3_ex'], 1e-9) and len(hps) == 24 and hps[0].shape == (1, 288, 1)
            he, nv, _ = rx.estimateChannelLsEx(cc, polarInt=False, int2d=True, kernel='thin_plate_spline', neighbors=9, smoothing=0.1)
            assert _close(he[:, ::5], g['e3_ex2'], 1e-9)
            assert abs(nv - g['e3_ex2_nv'][0]) <= 1e-7 * g['e3_ex2_nv'][0]
    with pytest.raises(ValueError):
        rx.estimateChannelLS(cc, kernel='cubic')
    with pytest.raises(ValueError):
        rx.estimateChannelLS("DMRS")


def test_dmrs_estimation_symbol_axis_kinds_and_2d_rbf():
    """Four DMRS symbols: the symbol axis goes through every 1-D kind, and estimateChannelLsEx through the 2-D RBF
    interpolation over (subcarrier, symbol) -- the reference's defaults (polar + thin-plate spline, 12 neighbours), a
    smoothed 14-neighbour spline and the RBF 'linear' kernel with a degree-1 tail."""
    import neoradium_amd as ma
    g = np.load(os.path.join(GOLD, 'csirs.npz'))
    car = ma.Carrier(numRbs=24, spacing=15)
    bwp = car.curBwp
    pd = ma.PDSCH(bwp, numLayers=2, modulation='16QAM')
    pd.setDMRS(configType=1, additionalPos=3)
    rx = ma.Grid(bwp, numPlanes=2)
    rx.grid = g['dm_rx']
    for kern in ('linear', 'nearest', 'quadratic', 'thin_plate_spline', 'multiquadric'):
        he, nv = rx.estimateChannelLS(pd.dmrs, kernel=kern, polarInt=(kern == 'quadratic'))
        assert _close(he[:, ::5], g[f'dm_{kern}'], 1e-9), kern
        assert abs(nv - g[f'dm_{kern}_nv'][0]) <= 1e-7 * g[f'dm_{kern}_nv'][0], kern
    he, nv, hps = rx.estimateChannelLsEx(pd.dmrs)
    assert _close(he[:, ::5], g['dm_ex'], 1e-9) and _close(hps[0][:, ::5], g['dm_ex_hk0'], 1e-9)
    assert abs(nv - g['dm_ex_nv'][0]) <= 1e-7 * g['dm_ex_nv'][0]
    he, nv, _ = rx.estimateChannelLsEx(pd.dmrs, polarInt=False, neighbors=14, smoothing=0.05)
    assert _close(he[:, ::5], g['dm_ex2'], 1e-9)
    he, nv, _ = rx.estimateChannelLsEx(pd.dmrs, polarInt=False, kernel='linear', neighbors=12, degree=1)
    assert _close(he[:, ::5], g['dm_ex3'], 1e-9)
    with pytest.raises(np.linalg.LinAlgError):                  # 6 nearest pilots are collinear: singular in scipy as well
        rx.estimateChannelLsEx(pd.dmrs, polarInt=False, neighbors=6)
    with pytest.raises(ValueError):                             # multiquadric needs epsilon in RBFInterpolator's 2-D call
        rx.estimateChannelLsEx(pd.dmrs, polarInt=False, kernel='multiquadric')


def test_csirs_timing_offset():
    import neoradium_amd as ma
    g = np.load(os.path.join(GOLD, 'csirs.npz'))
    car = ma.Carrier(numRbs=24, spacing=15)
    bwp = car.curBwp
    cc = ma.CsiRsConfig(csiType='NZP', bwp=bwp, numPorts=2, symbols=[3])
    tx = bwp.createGrid(2)
    cc.populateGrid(tx)
    for d in (0, 7, 23):
        off = tx.estimateTimingOffset(ma.Waveform(np.complex128(g[f't{d}_rx'])))
        assert off == int(g[f't{d}_off'][0]) == d
    # the correlation itself against a direct NumPy evaluation at a few lags
    from neoradium_amd import ops
    from neoradium_amd._dev import D, N
    ref = tx.ofdmModulate(windowing="NONE").waveform
    rxw = np.complex128(g['t7_rx'])
    nz = np.flatnonzero(np.abs(ref).max(0) > 0)
    xc = N(ops.xcorr_abs(D(rxw), D(ref), int(nz[0]), int(nz[-1] - nz[0] + 1)))
    for lag in (0, 7, 100, rxw.shape[1] - 5):
        want = sum(abs(np.vdot(ref[p][:rxw.shape[1] - lag], rxw[r][lag:])) for r in range(2) for p in range(2))
        assert abs(xc[lag] - want) <= 1e-9 * max(1.0, want)
