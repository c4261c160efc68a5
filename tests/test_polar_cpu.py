"""CPU: pin the polar oracle (oracle/polar.py) against (a) the MATLAB 5G-Toolbox vectors the reference ships
(Playground/CompareWithMatlab/Polar) and (b) fixtures generated from the reference (tools/gen_golden.py: polar),
and check the host-side code construction of neoradium_amd.polar against the same fixtures (no GPU needed)."""
import os

import numpy as np
import pytest
import scipy.io

from oracle import phy as op
from oracle.polar import PolarCode

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def mat(name):
    return scipy.io.loadmat(os.path.join(GOLD, 'matlab_polar', name + '.mat'))[name]


def test_polar_chain_vs_matlab():
    """PolarMatlab.ipynb: DCI, A=30, E=120 -> K=54, N=128; QPSK over AWGN (noiseVar 0.9241819678918566), L=8."""
    pc = PolarCode(30, 120, 'dci', 8)
    assert (pc.K, pc.N) == (54, 128)
    msg = mat('msg').reshape(-1).astype(np.int8)
    cbs = pc.segment(msg)
    assert np.array_equal(cbs[0], mat('msgcrc').reshape(-1))
    coded = pc.encode(cbs)
    assert np.array_equal(coded[0], mat('encOut').reshape(-1))
    rm = pc.rate_match(coded)
    assert np.array_equal(rm[0], mat('modIn').reshape(-1))
    sym = op.modulate(rm[0], 2)
    assert np.abs(sym - mat('modOut').reshape(-1)).max() < 1e-12
    rx = sym + mat('chanNoise').reshape(-1)
    llr = op.demap_maxlog(rx, 0.9241819678918566, 2)
    assert np.abs(llr - mat('rxLLR').reshape(-1)).max() < 1e-9
    rr = pc.rate_recover(llr[None, :])
    assert np.abs(rr[0] - mat('decIn').reshape(-1)).max() < 1e-9
    bits, nerr = pc.decode(rr)
    assert nerr == 0 and np.array_equal(bits, mat('decBits').reshape(-1)[:30]) and np.array_equal(bits, msg)


def _cases():
    g = np.load(os.path.join(GOLD, 'polar.npz'))
    for i, c in enumerate(g['cases']):
        typ, A, E = str(c).split(',')
        yield g, f'c{i}_', typ, int(A), int(E)


def test_polar_oracle_vs_reference_fixtures():
    n = 0
    for g, p, typ, A, E in _cases():
        pc = PolarCode(A, E, typ, 8)
        assert [pc.K, pc.N, pc.E, pc.n_pc] == g[p + 'params'].tolist()
        assert np.array_equal(pc.msg, g[p + 'msgBits']) and np.array_equal(pc.frozen, g[p + 'frozenBits'])
        assert np.array_equal(pc.pc, g[p + 'pcBits'])
        cbs = pc.segment(g[p + 'tb'])
        assert np.array_equal(cbs, g[p + 'cbs'])
        coded = pc.encode(cbs)
        assert np.array_equal(coded, g[p + 'coded'])
        assert np.array_equal(pc.rate_match(coded), g[p + 'rm'])
        for s in ('s0_', 's1_'):
            rr = pc.rate_recover(g[p + s + 'llr'])
            assert np.array_equal(rr, g[p + s + 'rr'])
            bits, nerr = pc.decode(rr)
            assert np.array_equal(bits, g[p + s + 'bits']) and nerr == int(g[p + s + 'nerr'])
            inv = None if pc.in_il is None else np.argsort(pc.in_il)
            for r, row in enumerate(np.clip(rr, -20, 20)):
                u, cost = pc.scl(row)
                m = u[:, pc.msg]
                m = m if inv is None else m[:, inv]
                assert np.array_equal(m, g[p + s + 'cands'][r]) and np.array_equal(cost, g[p + s + 'costs'][r])
            n += 1
    assert n == 30


def test_polar_repetition_round_trip():
    """E >= N (AL8 DCI: E=864 > N=512): reference crashes; the oracle follows TS 38.212 5.4.1.2 (sum of repeats)."""
    rng = np.random.default_rng(5)
    for typ, A, E in (('dci', 40, 864), ('uci', 100, 1500)):
        pc = PolarCode(A, E, typ, 8)
        tb = rng.integers(0, 2, A).astype(np.int8)
        rm = pc.rate_match(pc.encode(pc.segment(tb)))
        assert rm.shape[1] == pc.E >= pc.N
        rr = pc.rate_recover(4.0 * (1 - 2.0 * rm))
        counts = np.bincount(np.arange(pc.E) % pc.N, minlength=pc.N)
        assert np.array_equal(np.sort(np.abs(rr[0])), np.sort(4.0 * counts))
        bits, nerr = pc.decode(rr + 0.5 * rng.standard_normal(rr.shape))
        assert nerr == 0 and np.array_equal(bits, tb)


def test_polar_host_construction_vs_fixtures():
    """neoradium_amd.polar builds the same code (sizes, sets, interleavers) without touching the GPU."""
    from neoradium_amd.polar import PolarEncoder, PolarDecoder
    for g, p, typ, A, E in _cases():
        enc, dec = PolarEncoder(A, E, typ), PolarDecoder(A, E, typ, sclListSize=8)
        pc = PolarCode(A, E, typ)
        for o in (enc, dec):
            assert [o.codeBlockSize, o.polarCodeSize, o.rateMatchedBlockLen, o.nPC] == g[p + 'params'].tolist()
            assert np.array_equal(o.msgBits, g[p + 'msgBits']) and np.array_equal(o.frozenBits, g[p + 'frozenBits'])
            assert np.array_equal(o.pcBits, g[p + 'pcBits'])
        assert np.array_equal(enc._gather(), pc.rate_match(np.arange(pc.N)[None])[0])
        assert np.array_equal(dec.sbInterleaveIndexes, np.argsort(pc.sb_il))
        assert np.array_equal(enc.generator[:4, :4], [[1, 0, 0, 0], [1, 1, 0, 0], [1, 0, 1, 0], [1, 1, 1, 1]])
    with pytest.raises(ValueError):
        PolarEncoder(8, 60, 'uci')
    with pytest.raises(ValueError):
        PolarEncoder(30, 120, 'xyz')
    with pytest.raises(NotImplementedError):
        PolarDecoder(30, 120, 'dci', sclListSize=16)
    assert "Polar Encoder Properties" in repr(PolarEncoder(30, 120, 'dci'))
