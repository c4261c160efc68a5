"""CPU: the PDCCH candidate layer's oracle (oracle/pdcch.py, TS 38.212 7.3 / 38.211 7.3.2 on top of the pinned polar oracle)
and the host arithmetic of the product layer (neoradium_amd/pdcch.py: the CRC-mask identity its kernel path relies on).
The reference has no PDCCH, so there are no reference vectors: round trips, RNTI selectivity and the linearity identity."""
import numpy as np
import pytest

from oracle import coding as oc
from oracle import pdcch as opd


def test_crc_mask_identity():
    """A DCI's parity = plain CRC24C(payload) XOR a payload-independent mask (24 prepended ones + RNTI), and the plain CRC
    register over [payload, masked parity] ends at PDCCH.crcExpect -- what the SCL kernel tests instead of zero."""
    from neoradium_amd.pdcch import PDCCH
    rng = np.random.default_rng(1)
    for A, rnti in [(12, 0), (40, 0x1234), (64, 0xFFFF), (140, 77)]:
        a = rng.integers(0, 2, (5, A)).astype(np.int8)
        cbs = opd.dci_crc_attach(a, rnti)
        plain = oc.crc_bits(a, '24C')
        mask = PDCCH.parityMask(A, rnti)
        mbits = np.int8([(mask >> (23 - i)) & 1 for i in range(24)])
        assert np.array_equal(cbs[:, A:], plain ^ mbits[None, :])
        reg = oc.crc_bits(cbs, '24C')                                   # register over the whole word, as bits
        want = PDCCH.crcExpect(A, rnti)
        assert all(int(''.join(map(str, r)), 2) == want for r in reg)
        assert (PDCCH.crcExpect(A, 0) == 0) == (PDCCH.parityMask(A, 0) == 0)


@pytest.mark.parametrize("al", [1, 2, 4, 8, 16])
def test_oracle_dci_round_trip_and_rnti_selectivity(al):
    rng = np.random.default_rng(al)
    A, rnti, n_id = 41, 0x2B5C, 321
    a = rng.integers(0, 2, (1, A)).astype(np.int8)
    coded = opd.dci_encode(a, al, rnti)
    assert coded.shape == (1, 108 * al)
    sym = opd.pdcch_symbols(coded, rnti, n_id)[0]
    n_cce = 16
    grid = np.zeros(n_cce * 54, dtype=np.complex128)
    c0 = (n_cce - al) // al * al
    grid[c0 * 54:(c0 + al) * 54] = sym
    sigma = 0.5 if al > 1 else 0.2
    noisy = grid + sigma / np.sqrt(2) * (rng.standard_normal(grid.shape) + 1j * rng.standard_normal(grid.shape))
    cands = [(al, c) for c in range(0, n_cce - al + 1, al)]
    res = opd.blind_decode(noisy, sigma ** 2, A, rnti, n_id, cands)
    for (l, c), (found, bits) in zip(cands, res):
        assert found == (c == c0)
        if found:
            assert np.array_equal(bits, a[0])
    # another UE's RNTI finds nothing, not even on the occupied candidate
    res2 = opd.blind_decode(noisy, sigma ** 2, A, rnti ^ 0x0101, n_id, [(al, c0)])
    assert not res2[0][0]
