"""GPU: the PMI / rank search of the CSI report (csifeedback.py:419-536) on seeded channels against the reference's
choices and SINRs (tests/golden/csifeedback.npz, tools/gen_golden.py csifeedback)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def csi_channel(seed, L, K, nr, nt):
    """The fixture's channel (tools/gen_golden.py csi_channel): four delayed taps with slow phase drift, from the seed."""
    rng = np.random.default_rng(seed)
    taps = (rng.standard_normal((4, nr, nt)) + 1j * rng.standard_normal((4, nr, nt))) * np.float64([1, .7, .4, .2])[:, None, None]
    k = np.arange(K)[None, :, None]
    l = np.arange(L)[:, None, None]
    ph = np.exp(-2j * np.pi * (k * np.float64([0, 1.7, 3.1, 6.4]) / 512 - l * np.float64([.004, -.006, .002, 0])))
    return (taps[None, None] * ph[..., None, None]).sum(2) / 2


def test_pmi_and_rank_search_vs_reference():
    import neoradium_amd as ma
    g = np.load(os.path.join(GOLD, 'csifeedback.npz'))
    ties = []
    for i, c in enumerate(json.loads(str(g['cfgs']))):
        car = ma.Carrier(numRbs=c['rb'], spacing=15)
        cc = ma.CsiRsConfig(csiType='NZP', bwp=car.curBwp, numPorts=c['ports'], cdmSize=c['cdm'], **c['kw'])
        rep = ma.CsiReport(cc, **c['rep'])
        h = csi_channel(100 + i, 14, 12 * c['rb'], c['nr'], c['ports'])
        for rank in c['ranks']:
            pmi, ws, sb = rep.bestPmiForRank(h, rank, c['nv'])
            sinr = np.concatenate([np.asarray(v) for v in sb])
            ref = g[f'r{i}_{rank}_sinr']
            assert sinr.shape == ref.shape
            if list(pmi[0]) + list(pmi[1]) == g[f'r{i}_{rank}_pmi'].tolist():
                w = np.array([np.asarray(x).reshape(c['ports'], rank) for x in ws])
                assert np.abs(w - g[f'r{i}_{rank}_w']).max() < 1e-14
                assert np.abs(sinr - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), (i, rank)
            else:
                # an exact tie of the metric: codebook entries that hold the same beams in another column order (e.g. i11 = 3,
                # k1 = 3 O1 and i11 = 15, k1 = O1 on N1 = 4) have equal summed SINR, and rounding picks one.  The choice must then
                # be worth the same as the reference's in every sub-band, and the reference's entry must score the same here.
                ties.append((i, rank))
                edges = np.cumsum([0] + [len(v) for v in sb])
                for a, b in zip(edges[:-1], edges[1:]):
                    assert abs(sinr[a:b].sum() - ref[a:b].sum()) <= 1e-9 * abs(ref[a:b].sum()), (i, rank)
                rp = g[f'r{i}_{rank}_pmi'].tolist()
                w_ref = np.stack([rep.getType1SpPrecoder(rank, rp[:3], i2) for i2 in rp[3:]])
                assert np.abs(w_ref - g[f'r{i}_{rank}_w']).max() < 1e-14
        rank, pmi, sb = rep.getBestRank(h, c['nv'])
        best = g[f'r{i}_best'].tolist()
        assert rank == best[0] and ((i, rank) in ties or list(pmi[0]) + list(pmi[1]) == best[1:]), i
    assert len(ties) <= 3, ties
    with pytest.raises(ValueError):
        rep.getBestRank(h[..., :1], 0.01)


def test_sinr_kernel_vs_svd_formula():
    """nrx_csi_sinr_f64 against the reference's SVD expression (csifeedback.py:424-433) evaluated in NumPy."""
    from neoradium_amd import ops
    from neoradium_amd._dev import D, N
    rng = np.random.default_rng(3)
    for nr_, nt, nl, ncb in ((1, 2, 1, 4), (2, 4, 2, 7), (4, 8, 3, 5), (8, 32, 5, 3), (8, 16, 8, 2)):
        h = rng.standard_normal((37, nr_, nt)) + 1j * rng.standard_normal((37, nr_, nt))
        w = (rng.standard_normal((ncb, nt, nl)) + 1j * rng.standard_normal((ncb, nt, nl))) / np.sqrt(nt * nl)
        nv = 0.03
        heff = h[None] @ w[:, None]
        _, s, vh = np.linalg.svd(heff, full_matrices=True)
        ref = 1 / (nv * ((1 / (s ** 2 + nv))[..., None] * np.abs(vh) ** 2).sum(2)) - 1
        got = N(ops.csi_sinr(D(h), D(w), nv))
        assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), (nr_, nt, nl)
    with pytest.raises(Exception):
        ops.csi_sinr(D(h[:, :2]), D(w), nv)                   # more layers than receive antennas
