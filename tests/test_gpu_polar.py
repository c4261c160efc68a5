"""GPU parity: polar codec through the C ABI (class surface neoradium_amd.polar) vs the reference fixtures, the
MATLAB vectors and the NumPy oracle.  Bits, list order and path costs are compared exactly (float64 arithmetic is
operation-for-operation that of the reference)."""
import os

import numpy as np
import pytest
import scipy.io

from oracle.polar import PolarCode

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def mat(name):
    return scipy.io.loadmat(os.path.join(GOLD, 'matlab_polar', name + '.mat'))[name]


def test_polar_matlab_notebook(dev):
    """PolarMatlab.ipynb through the class surface."""
    import neoradium_amd as nr
    enc = nr.PolarEncoder(30, 120, 'dci')
    dec = nr.PolarDecoder(30, 120, 'dci', sclListSize=8)
    msg = mat('msg').reshape(-1).astype(np.int8)
    cbs = enc.doSegmentation(msg)
    assert np.array_equal(cbs[0], mat('msgcrc').reshape(-1))
    coded = enc.encode(cbs)
    assert np.array_equal(coded[0], mat('encOut').reshape(-1))
    rm = enc.rateMatch(coded)
    assert np.array_equal(rm[0], mat('modIn').reshape(-1))
    modem = nr.Modem('QPSK')
    sym = modem.modulate(rm[0])
    assert np.abs(sym - mat('modOut').reshape(-1)).max() < 1e-12
    llr = modem.getLLRsFromSymbols(sym + mat('chanNoise').reshape(-1), 0.9241819678918566)
    assert np.abs(llr - mat('rxLLR').reshape(-1)).max() < 1e-9
    rr = dec.recoverRate(llr[None, :])
    assert np.abs(rr[0] - mat('decIn').reshape(-1)).max() < 1e-9
    bits, nerr = dec.decode(rr)
    assert nerr == 0 and np.array_equal(bits, mat('decBits').reshape(-1)[:30])


def test_polar_vs_reference_fixtures(dev):
    from neoradium_amd.polar import PolarEncoder, PolarDecoder
    from neoradium_amd._dev import D, N
    g = np.load(os.path.join(GOLD, 'polar.npz'))
    for i, c in enumerate(g['cases']):
        typ, A, E = str(c).split(',')
        A, E, p = int(A), int(E), f'c{i}_'
        enc, dec = PolarEncoder(A, E, typ), PolarDecoder(A, E, typ, sclListSize=8)
        cbs = enc.doSegmentation(g[p + 'tb'])
        assert np.array_equal(cbs, g[p + 'cbs'])
        coded = enc.encode(cbs)
        assert np.array_equal(coded, g[p + 'coded']), (typ, A, E)
        assert np.array_equal(enc.rateMatch(coded), g[p + 'rm'])
        for s in ('s0_', 's1_'):
            rr = dec.recoverRate(g[p + s + 'llr'])
            assert np.array_equal(rr, g[p + s + 'rr'])
            bits, nerr = dec.decode(rr)
            assert np.array_equal(bits, g[p + s + 'bits']) and nerr == int(g[p + s + 'nerr']), (typ, A, E, s)
            _, _, cands, costs = dec.decodeDevice(D(rr), wantCandidates=True)
            assert np.array_equal(N(cands), g[p + s + 'cands']), (typ, A, E, s)
            assert np.array_equal(N(costs), g[p + s + 'costs']), (typ, A, E, s)


@pytest.mark.parametrize("typ,A,E,L", [('dci', 40, 864, 8), ('dci', 140, 864, 8), ('dci', 57, 216, 4),
                                       ('uci', 100, 1500, 8), ('uci', 1013, 2100, 8), ('dci', 33, 108, 1),
                                       ('pbch', 32, 864, 2), ('uci', 300, 700, 8)])
def test_polar_batch_vs_oracle(dev, typ, A, E, L):
    """Batched blind-decode candidates incl. the repetition cases (E >= N, AL8) the reference cannot run, list
    sizes 1..8 and N = 1024; noisy enough that some candidates fail the CRC.  Every row must match the oracle."""
    from neoradium_amd.polar import PolarEncoder, PolarDecoder
    from neoradium_amd._dev import D, N
    rng = np.random.default_rng(A + E)
    pc = PolarCode(A, E, typ, L)
    enc, dec = PolarEncoder(A, E, typ), PolarDecoder(A, E, typ, sclListSize=L)
    n = 12
    tbs = rng.integers(0, 2, (n, A)).astype(np.int8)
    cbs = np.concatenate([enc.doSegmentation(t) for t in tbs])
    assert np.array_equal(cbs, np.concatenate([pc.segment(t) for t in tbs]))
    coded = enc.encode(cbs)
    assert np.array_equal(coded, pc.encode(cbs))
    rm = enc.rateMatch(coded)
    assert np.array_equal(rm, pc.rate_match(coded))
    snr = 10 * np.log10(pc.K / pc.E) + 1.0            # around the waterfall of each code
    sig = 10 ** (-snr / 20)
    llr = 2 * (1 - 2.0 * rm + sig * rng.standard_normal(rm.shape)) / sig ** 2
    llr[1] = -llr[1]                                   # a candidate that is not a code word of this format
    rr = dec.recoverRate(llr)
    assert np.array_equal(rr, pc.rate_recover(llr))
    msg, ok, cands, costs = dec.decodeDevice(D(rr), wantCandidates=True)
    msg, ok, cands, costs = N(msg), N(ok), N(cands), N(costs)
    inv = None if pc.in_il is None else np.argsort(pc.in_il)
    n_fail = 0
    for r, row in enumerate(np.clip(rr, -20, 20)):
        u, cost = pc.scl(row)
        m = u[:, pc.msg]
        m = m if inv is None else m[:, inv]
        assert np.array_equal(cands[r, :len(m)], m), (r,)
        assert np.array_equal(costs[r, :len(m)], cost), (r,)
        bits, nerr = pc.decode(rr[r:r + 1]) if not pc.seg else (None, None)
        if bits is not None:
            crc = 0 if pc.crc is None else int(pc.crc[:2]) if pc.crc[:2] == '24' else int(pc.crc)
            assert np.array_equal(msg[r, :pc.K - crc][-A:], bits) and bool(ok[r]) == (nerr == 0)
            n_fail += nerr
    if not pc.seg:
        bits, okc = dec.decodeCandidates(rr)
        assert np.array_equal(okc, ok.astype(bool)) and bits.shape == (n, A)
        assert n_fail >= 1                                  # the inverted candidate at least
    else:
        for t in range(n):
            bits, nerr = dec.decode(rr[2 * t:2 * t + 2])
            ob, on = pc.decode(rr[2 * t:2 * t + 2])
            assert np.array_equal(bits, ob) and nerr == on


def test_polar_ties_are_stable(dev):
    """Saturated LLRs (+-20 after the clip) give exactly tied path costs; the ranking keeps the lower index first."""
    from neoradium_amd.polar import PolarEncoder, PolarDecoder
    from neoradium_amd._dev import D, N
    rng = np.random.default_rng(3)
    pc = PolarCode(30, 120, 'dci', 8)
    enc, dec = PolarEncoder(30, 120, 'dci'), PolarDecoder(30, 120, 'dci')
    tb = rng.integers(0, 2, (6, 30)).astype(np.int8)
    rm = enc.rateMatch(enc.encode(np.concatenate([enc.doSegmentation(t) for t in tb])))
    llr = 50.0 * (1 - 2.0 * rm)
    llr[:, ::7] *= -1                                       # a few hard errors
    llr[:, ::5] = 0.0                                       # and erasures
    rr = dec.recoverRate(llr)
    _, _, cands, costs = dec.decodeDevice(D(rr), wantCandidates=True)
    inv = np.argsort(pc.in_il)
    for r, row in enumerate(np.clip(rr, -20, 20)):
        u, cost = pc.scl(row)
        assert np.array_equal(N(cands)[r], u[:, pc.msg][:, inv]) and np.array_equal(N(costs)[r], cost)


def test_polar_bad_arguments(dev):
    from neoradium_amd.polar import PolarEncoder, PolarDecoder
    enc, dec = PolarEncoder(30, 120, 'dci'), PolarDecoder(30, 120, 'dci')
    with pytest.raises(ValueError):
        enc.encode(np.zeros((1, 50), np.int8))
    with pytest.raises(ValueError):
        enc.rateMatch(np.zeros((1, 64), np.int8))
    with pytest.raises(ValueError):
        dec.recoverRate(np.zeros((1, 100)))
    with pytest.raises(ValueError):
        dec.decode(np.zeros((1, 64)))


def test_pdcch_blind_decoding_vs_oracle(dev):
    """The PDCCH candidate layer (neoradium_amd/pdcch.py; absent from the reference, defined by TS 38.212 7.3 / 38.211 7.3.2):
    DCI encoding bit-identical to the oracle for every aggregation level (AL 8 / 16 repeat: E > N), and a batch of
    monitoring occasions -- two DCIs for this UE at different aggregation levels, one for another UE, noise -- blind-decoded
    over all 31 aligned candidates of a 16-CCE CORESET: the detections (RNTI-masked CRC tested inside the SCL kernel) and
    payloads equal the oracle's, which tests the CRC the long way."""
    import torch
    import neoradium_amd as nr
    from oracle import pdcch as opd
    rng = np.random.default_rng(12)
    A, rnti, other, n_id, n_cce = 44, 0x1A2B, 0x0C0D, 77, 16
    pd = nr.PDCCH(n_cce, nID=n_id, rnti=rnti)
    for al in (1, 2, 4, 8, 16):
        a = rng.integers(0, 2, (3, A)).astype(np.uint8)
        assert np.array_equal(pd.dciEncode(a, al).cpu().numpy(), opd.dci_encode(a, al, rnti))
        sym = pd.encode(a, al).cpu().numpy()
        assert np.abs(sym - opd.pdcch_symbols(opd.dci_encode(a, al, rnti), rnti, n_id)).max() < 1e-12
    n_occ = 6
    grid = np.zeros((n_occ, n_cce * 54), dtype=np.complex128)
    sent = []
    for o in range(n_occ):
        a4, a2, ax = (rng.integers(0, 2, (1, A)).astype(np.uint8) for _ in range(3))
        grid[o, 4 * 54:8 * 54] = pd.encode(a4, 4).cpu().numpy()[0]                 # this UE, AL 4 at CCE 4
        grid[o, 10 * 54:12 * 54] = pd.encode(a2, 2).cpu().numpy()[0]               # this UE, AL 2 at CCE 10
        grid[o, 0:54] = pd.encode(ax, 1, rnti=other).cpu().numpy()[0]              # another UE, AL 1 at CCE 0
        sent.append((a4[0], a2[0]))
    sigma = 0.45
    noisy = grid + sigma / np.sqrt(2) * (rng.standard_normal(grid.shape) + 1j * rng.standard_normal(grid.shape))
    found, bits, cands = pd.blindDecode(noisy, sigma ** 2, A)
    found, bits = found.cpu().numpy(), bits.cpu().numpy()
    assert len(cands) == 31 and found.shape == (n_occ, 31)
    for o in range(n_occ):
        ref = opd.blind_decode(noisy[o], sigma ** 2, A, rnti, n_id, cands)
        assert [f for f, _ in ref] == found[o].tolist()
        for i, (f, b) in enumerate(ref):
            if f:
                assert np.array_equal(bits[o, i], b)
        assert found[o, cands.index((4, 4))] and np.array_equal(bits[o, cands.index((4, 4))], sent[o][0])
        assert not found[o, cands.index((1, 0))]                                    # the other UE's DCI is not ours
    hits = found.sum()
    assert hits >= 2 * n_occ - 2                                                    # (AL 2 at this SNR may be missed)
    # the other UE sees its own DCI and none of ours
    f2, b2, _ = pd.blindDecode(noisy, sigma ** 2, A, rnti=other)
    f2 = f2.cpu().numpy()
    assert not f2[:, cands.index((4, 4))].any() and not f2[:, cands.index((2, 10))].any()
