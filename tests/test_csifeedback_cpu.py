"""CPU: host logic of the CSI report (csifeedback.py) against reference-generated fixtures -- codebook enumeration and
precoders of the Type-I single-panel codebooks, sub-band layouts, CQI tables.  (The SINR search is a GPU test.)"""
import json
import os

import numpy as np
import pytest

import neoradium_amd as ma
from neoradium_amd import csifeedback

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _report(c):
    car = ma.Carrier(numRbs=c['rb'], spacing=15)
    cc = ma.CsiRsConfig(csiType='NZP', bwp=car.curBwp, numPorts=c['ports'], cdmSize=c['cdm'], **c['kw'])
    return ma.CsiReport(cc, **c['rep'])


def test_type1_single_panel_codebooks_and_layouts():
    g = np.load(os.path.join(GOLD, 'csifeedback.npz'))
    for i, c in enumerate(json.loads(str(g['cfgs']))):
        rep = _report(c)
        for rank in c['ranks']:
            idx, cb = rep.getCodebook(rank)
            assert np.array_equal(np.int32([list(a) + [b] for a, b in idx]), g[f'r{i}_{rank}_idx']), (i, rank)
            assert cb.shape == (len(idx), c['ports'], rank)
            assert np.abs(cb[::max(1, len(cb) // 24)] - g[f'r{i}_{rank}_cb']).max() < 1e-14, (i, rank)
            assert np.abs(np.linalg.norm(cb, axis=(1, 2)) - 1).max() < 1e-12           # unit total power
        assert list(rep.subbands(4)) + [-1] + list(rep.subbands(8)) == g[f'r{i}_subbands'].tolist()
        assert [rep.getCqiToPmiIdxes(0), rep.getCqiToPmiIdxes(2), rep.getCqiToPmiIdxes(4)] == json.loads(str(g[f'r{i}_cqi2pmi']))
    car = ma.Carrier(startRb=3, numRbs=50, spacing=15)
    rep = ma.CsiReport(ma.CsiRsConfig(csiType='NZP', bwp=car.curBwp, numPorts=8), n1=4, n2=1, subbandSizeCqi=8, subbandSizePmi=4)
    assert list(rep.subbands(4)) + [-1] + list(rep.subbands(8)) == g['off_subbands'].tolist()
    assert [rep.getCqiToPmiIdxes(4), rep.getCqiToPmiIdxes(2)] == json.loads(str(g['off_cqi2pmi']))
    assert csifeedback.cqiTables == json.loads(str(g['cqi_tables']))
    assert "codebookMode" in repr(rep)


def test_report_validation_and_unsupported_corners():
    car = ma.Carrier(numRbs=24, spacing=15)
    cc = ma.CsiRsConfig(csiType='NZP', bwp=car.curBwp, numPorts=8)
    for bad in (dict(n1=5, n2=1), dict(n1=4), dict(n1=4, n2=1, codebookMode=3), dict(n1=4, n2=1, period=7),
                dict(n1=4, n2=1, subbandSize=16), dict(n1=4, n2=1, prgSize=3), dict(n1=4, n2=1, cqiTable=5),
                dict(n1=2, n2=1, ng=2, codebookType='Type1MP', txAntenna=ma.AntennaPanel([1, 4]))):
        with pytest.raises(ValueError):
            ma.CsiReport(cc, **bad)
    zp = ma.CsiRsConfig(csiType='ZP', bwp=car.curBwp, numPorts=8)
    with pytest.raises(ValueError):
        ma.CsiReport(zp, n1=4, n2=1)
    rep = ma.CsiReport(cc, txAntenna=ma.AntennaPanel([1, 4], polarization="x"))
    assert (rep.n1, rep.n2, rep.numPorts) == (4, 1, 8)
    with pytest.raises(NotImplementedError):                # ranks >= 6 stop on an undefined attribute in the reference
        rep.getCodebook(6)
    rep22 = ma.CsiReport(cc, n1=2, n2=2)
    assert len(list(rep22.type1SpIndexes(1))) == 8 * 8 * 4
    with pytest.raises(NotImplementedError):                # N2 > 1: the reference's precoder is not (ports, layers) there
        rep22.getCodebook(1)
    with pytest.raises(NotImplementedError):
        ma.CsiReport(cc, n1=2, n2=1, ng=2, codebookType='Type1MP').getCodebook(1)
    rep2 = ma.CsiReport(ma.CsiRsConfig(csiType='NZP', bwp=car.curBwp, numPorts=2), n1=1, n2=1)
    with pytest.raises(IndexError):                         # two layers on two ports run off the 2-bit field (reference too)
        rep2.getCodebook(2)
    # the restriction bitmaps prune the enumeration
    pruned = ma.CsiReport(cc, n1=4, n2=1, cbSubsetRestriction='0' * 8 + '1' * 24)
    assert all(i1[0] >= 8 for i1, _ in pruned.type1SpIndexes(1)) and len(list(pruned.type1SpIndexes(1))) == 8 * 4


def test_multi_panel_codebooks_fail_in_the_reference_itself():
    """Type-I multi-panel (csifeedback.py:566-577, 1040-1327): tests/golden/csifeedback_multipanel.json records what the REFERENCE's
    getCodebook does for every Ng-N1-N2 combination of TS 38.214 Table 5.2.2.2.2-1 x codebook mode x 1..4 layers (tools/gen_golden.py
    csifeedback_multipanel): mode 1 and every one-layer case raise, mode 2 with more layers returns arrays that are not
    (ports x layers) -- no configuration yields a usable codebook.  neoradium_amd therefore raises NotImplementedError for the
    codebook type (the constructor validates like the reference) instead of restating code that cannot be pinned."""
    d = json.load(open(os.path.join(GOLD, 'csifeedback_multipanel.json')))
    rows = d['rows']
    assert len(rows) == 52 and {(r['ng'], r['n1'], r['n2']) for r in rows} == {(2, 2, 1), (2, 4, 1), (4, 2, 1), (2, 2, 2), (2, 8, 1), (4, 4, 1), (2, 4, 2), (4, 2, 2)}
    assert not any(r.get('is_ports_by_layers') for r in rows)
    raises = [r for r in rows if r['outcome'] == 'raises']
    assert all(r['file'] == 'csifeedback.py' and 1040 <= r['line'] <= 1327 for r in raises)
    assert all(r['outcome'] == 'raises' for r in rows if r['mode'] == 1 or r['layers'] == 1)
    assert all(r['outcome'] == 'returns' and r['shape'][1:] != [r['ports'], r['layers']] for r in rows if r['mode'] == 2 and r['layers'] >= 2)
    car = ma.Carrier(numRbs=24, spacing=15)
    for r in rows[::7]:
        ports = r['ports']
        cc = ma.CsiRsConfig(csiType='NZP', bwp=car.curBwp, numPorts=ports, cdmSize=(2 if ports <= 12 else (4 if ports <= 16 else 8)))
        rep = ma.CsiReport(cc, codebookType='Type1MP', ng=r['ng'], n1=r['n1'], n2=r['n2'], codebookMode=r['mode'])
        with pytest.raises(NotImplementedError):
            rep.getCodebook(r['layers'])
