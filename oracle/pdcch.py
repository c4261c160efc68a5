"""Oracle (CPU, NumPy) for the PDCCH candidate layer -- TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference has NO PDCCH (neoradium/dmrs.py:199-203: "only PDSCH is implemented"): BASELINE cfg4's
"batched DCI blind-decode candidates" only exists there as the polar codec (polar.py).  PARITY UNPINNED: this layer is
defined by the specifications it cites, on top of the polar oracle (oracle/polar.py, which IS pinned by the MATLAB and
reference fixtures), and is checked by round trips and by the product/oracle cross-check:

  TS 38.212 7.3.2  CRC attachment: CRC24C over [24 ones, payload]; the last 16 parity bits are XORed with the RNTI
  TS 38.212 7.3.3/7.3.4  polar coding (n_max 9, input interleaving) and rate matching to E = 108 * aggregation level bits
                   (one CCE = 6 REGs x (12 - 3 DMRS) REs x 2 bits)
  TS 38.211 7.3.2.3/7.3.2.4  scrambling with c_init = (n_RNTI * 2^16 + n_ID) mod 2^31, QPSK
  blind decoding: a candidate = (aggregation level, first CCE); its LLR window is descrambled, rate-recovered, SCL decoded,
                   and accepted iff some list entry's CRC matches under the RNTI mask.

Written directly from these statements (the CRC test recomputes CRC24C([ones, payload]) and compares), NOT with the
register-shift identity the product's kernel uses -- that identity is one of the things the cross-check verifies."""
import numpy as np

from . import coding as oc
from . import phy as op
from .polar import PolarCode

BITS_PER_CCE = 108


def rnti_bits(rnti):
    return np.int8([(int(rnti) >> (15 - i)) & 1 for i in range(16)])


def dci_crc_attach(payload, rnti):
    """TS 38.212 7.3.2: (n, A) bits -> (n, A + 24)."""
    a = np.atleast_2d(np.asarray(payload)).astype(np.int8)
    par = oc.crc_bits(np.concatenate([np.ones((a.shape[0], 24), np.int8), a], axis=1), '24C')
    par[:, 8:] ^= rnti_bits(rnti)[None, :]
    return np.concatenate([a, par], axis=1)


def dci_encode(payload, agg_level, rnti):
    """(n, A) DCI payloads -> (n, 108 * agg_level) coded bits."""
    a = np.atleast_2d(np.asarray(payload))
    pc = PolarCode(a.shape[1], BITS_PER_CCE * agg_level, 'dci')
    return pc.rate_match(pc.encode(dci_crc_attach(a, rnti)))


def pdcch_symbols(coded, rnti, n_id):
    """TS 38.211 7.3.2.3-7.3.2.4: scramble + QPSK, (n, E) bits -> (n, E/2) symbols."""
    coded = np.atleast_2d(np.asarray(coded)).astype(np.int8)
    c = op.gold((int(rnti) * 65536 + int(n_id)) % (1 << 31), coded.shape[1])
    return np.stack([op.modulate(row ^ c, 2) for row in coded])


def blind_decode(symbols, noise_var, A, rnti, n_id, candidates, list_size=8):
    """symbols: (numCces * 54,) equalised QPSK symbols of one monitoring occasion.  candidates: [(aggLevel, firstCce)].
    Returns [(found, payload bits (A,))] per candidate."""
    out = []
    for al, cce in candidates:
        E = BITS_PER_CCE * al
        pc = PolarCode(A, E, 'dci', list_size)
        llr = op.demap_maxlog(symbols[cce * 54:(cce + al) * 54], noise_var, 2)
        llr = llr * (1 - 2 * op.gold((int(rnti) * 65536 + int(n_id)) % (1 << 31), E).astype(np.float64))
        rr = np.clip(pc.rate_recover(llr[None, :]), -20, 20)[0]
        u, _ = pc.scl(rr)
        m = u[:, pc.msg]
        if pc.in_il is not None:
            m = m[:, np.argsort(pc.in_il)]
        found, best = False, m[0][:A]
        for cand in m:
            if np.array_equal(dci_crc_attach(cand[:A], rnti)[0][A:], cand[A:]):
                found, best = True, cand[:A]
                break
        out.append((found, np.int8(best)))
    return out
