"""Oracle (CPU, NumPy float64/int) for CRC + 5G LDPC.  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Every function names the reference lines (InterDigitalInc/NeoRadium v0.4.0) whose behaviour it restates.
Quirks of the reference that parity depends on are kept on purpose and flagged with "QUIRK".
"""
import os
import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'neoradium_amd', 'data')

LARGE_LLR = 1e20                     # chancodebase.py:52
LLR_CLIP = 1e10                      # ldpc.py:1536

# ----------------------------------------------------------------------------------------------------------- CRC
# chancodebase.py:37-44 -- generator polynomials (MSB first, including the leading 1)
CRC_POLY = {'6': 0x61, '11': 0xE21, '16': 0x11021, '24A': 0x1864CFB, '24B': 0x1800063, '24C': 0x1B2B117}
CRC_LEN = {'6': 6, '11': 11, '16': 16, '24A': 24, '24B': 24, '24C': 24}


def crc_bits(bits, poly):
    """chancodebase.py:83-128 getCrc: remainder of bits*x^L by the generator, zero init, no final xor, MSB first.

    ``bits``: (n,) or (m,n) 0/1 array.  Returns (L,) or (m,L) int8.  (The reference does the long division one
    message bit at a time over a padded array; the remainder is the same polynomial remainder computed here with
    a shift register, vectorised over rows.)"""
    bits = np.asarray(bits)
    flat = bits.ndim == 1
    b = np.atleast_2d(bits).astype(np.int64)
    L, g = CRC_LEN[poly], CRC_POLY[poly]
    low = g & ((1 << L) - 1)
    reg = np.zeros(b.shape[0], dtype=np.int64)
    for d in range(b.shape[1]):
        top = ((reg >> (L - 1)) & 1) ^ b[:, d]
        reg = ((reg << 1) & ((1 << L) - 1)) ^ (top * low)
    out = ((reg[:, None] >> np.arange(L - 1, -1, -1)[None, :]) & 1).astype(np.int8)
    return out[0] if flat else out


def crc_append(bits, poly):
    """chancodebase.py:161-189 appendCrc."""
    bits = np.asarray(bits)
    return np.concatenate([bits.astype(np.int8), crc_bits(bits, poly)], axis=-1)


def crc_check(bits, poly):
    """chancodebase.py:132-157 checkCrc: remainder of the whole stream (payload+CRC) is all-zero."""
    return np.count_nonzero(crc_bits(bits, poly), axis=-1) == 0


# ---------------------------------------------------------------------------------------------------------- LDPC
LIFTING_SETS = [[a << j for j in range(8) if (a << j) <= 384] for a in (2, 3, 5, 7, 9, 11, 13, 15)]  # ldpc.py:657-666
_bgdata = None


def _edges(bgn):
    global _bgdata
    if _bgdata is None:
        _bgdata = np.load(os.path.join(_DATA, 'ldpc_bg.npz'))
    return _bgdata[f'bg{bgn}_row'], _bgdata[f'bg{bgn}_col'], _bgdata[f'bg{bgn}_shift']


class LdpcParams:
    """ldpc.py:859-892 initialize + the derived sizes used by doSegmentation/recoverRate.

    ``B`` is the transport-block size *including* the 24-bit TB CRC."""

    def __init__(self, bgn, B):
        self.bgn, self.B = bgn, int(B)
        kcb = 8448 if bgn == 1 else 3840
        if B <= kcb:
            self.C, tot = 1, B
        else:
            self.C = int(np.ceil(B / (kcb - 24)))
            tot = B + self.C * 24
        kprime = tot / self.C                       # may be fractional (ldpc.py:874)
        if bgn == 1:   kb = 22
        elif B > 640:  kb = 10                      # QUIRK: keyed on B, not on K' (ldpc.py:876-879)
        elif B > 560:  kb = 9
        elif B > 192:  kb = 8
        else:          kb = 6
        best, ils = 10000, -1
        for i, zs in enumerate(LIFTING_SETS):       # first strictly-smaller Z wins (ldpc.py:884-889)
            for z in zs:
                if kb * z >= kprime and z < best:
                    best, ils = z, i
        self.Zc, self.iLS = best, ils
        self.K = (22 if bgn == 1 else 10) * self.Zc
        self.N = (66 if bgn == 1 else 50) * self.Zc
        self.bitsPerCb = int(np.ceil(B / self.C)) + (24 if self.C > 1 else 0)   # ldpc.py:1014,1367-1368
        self.F = self.K - self.bitsPerCb                                          # ldpc.py:1026,1369


def base_graph(bgn, ils, zc):
    """ldpc.py:776-789: (rows x cols) int16, -1 = no edge, else table value mod Zc."""
    r, c, s = _edges(bgn)
    bg = -np.ones((46, 68) if bgn == 1 else (42, 52), dtype=np.int16)
    bg[r, c] = s[:, ils] % zc
    return bg


def _rot(x, k):
    """ldpc.py:793-810 mulShift on the last axis: out[..., i] = x[..., (i+k) mod z]  (k==z is identity)."""
    return np.roll(x, -int(k), axis=-1)


def cb_lens(G, C, nl, qm):
    """ldpc.py:846-856 getRateMatchedCbLens."""
    f = nl * qm
    gb = int(np.ceil(G / f))
    e = np.full(C, (gb // C) * f, dtype=np.int64)
    if gb % C:
        e[C - gb % C:] += f
    return e


def segment(tb_with_crc, bgn):
    """ldpc.py:981-1030 doSegmentation -> (C,K) int8, params.  Fillers are ZERO bits (QUIRK, ldpc.py:1025-1028);
    the zero padding sits at the end of the last block before its CRC (ldpc.py:1014-1016)."""
    tb = np.asarray(tb_with_crc, dtype=np.int8)
    p = LdpcParams(bgn, len(tb))
    per = int(np.ceil(p.B / p.C))
    cbs = np.zeros(p.C * per, dtype=np.int8)
    cbs[:p.B] = tb
    cbs = cbs.reshape(p.C, per)
    if p.C > 1:
        cbs = crc_append(cbs, '24B')
    out = np.zeros((p.C, p.K), dtype=np.int8)
    out[:, :cbs.shape[1]] = cbs
    return out, p


def encode(cbs, bgn, ils, zc, puncture=True):
    """ldpc.py:1033-1090 encode: (C,K) -> (C,N) (first 2Zc systematic bits dropped when ``puncture``)."""
    bg = base_graph(bgn, ils, zc)
    nrow, ncol = bg.shape
    k = ncol - nrow
    C = cbs.shape[0]
    w = np.zeros((C, ncol, zc), dtype=np.int8)
    w[:, :k] = cbs.reshape(C, k, zc)

    def rowsum(r, cols):
        acc = np.zeros((C, zc), dtype=np.int8)
        for j in cols:
            if bg[r, j] >= 0:
                acc ^= _rot(w[:, j], bg[r, j])
        return acc

    core = [rowsum(r, range(k)) for r in range(4)]
    undo = zc - (bg[2, k] if bg[1, k] == -1 else bg[1, k])          # ldpc.py:1068
    w[:, k] = _rot(core[0] ^ core[1] ^ core[2] ^ core[3], undo)      # p0
    for i in range(3):                                               # p1..p3 (ldpc.py:1077-1080)
        w[:, k + i + 1] = core[i] ^ rowsum(i, range(k, k + i + 1))
    for r in range(4, nrow):                                         # extension parities (ldpc.py:1083-1084)
        w[:, k + r] = rowsum(r, range(k + 4))
    w = w.reshape(C, -1)
    return w[:, 2 * zc:] if puncture else w


_K0 = {1: (0, 17, 33, 56), 2: (0, 13, 25, 43)}


def rate_match(coded, p, G, nl, qm, rv=0, nref=0, concat=True):
    """ldpc.py:1093-1159 rateMatch.  QUIRK: fillers are removed from the circular buffer but k0 is NOT reduced by F
    (ldpc.py:1139-1148)."""
    C, N = coded.shape
    z = p.Zc
    ncb = N if nref == 0 else min(N, nref)
    sys_len = p.K - 2 * z
    circ = np.concatenate([coded[:, :sys_len - p.F], coded[:, sys_len:ncb]], axis=1)
    k0 = (_K0[p.bgn][rv] * ncb // N) * z
    E = cb_lens(G, C, nl, qm)
    out = []
    for r in range(C):
        e = int(E[r])
        sel = circ[r][(np.arange(e) + k0) % circ.shape[1]]
        out.append(sel.reshape(qm, e // qm).T.reshape(-1))            # bit interleaver (38.212 5.4.2.2)
    return np.concatenate(out) if concat else out


def rate_recover(llr, p, nl, qm, rv=0, nref=0, circ=None):
    """ldpc.py:1330-1418 recoverRate: returns ((C,N) float64, updated circular buffer (C, Ncb-F)).
    ``circ`` is the HARQ soft buffer to accumulate into (None = zeros)."""
    llr = np.asarray(llr, dtype=np.float64)
    z, C, N = p.Zc, p.C, p.N
    ncb = N if nref == 0 else min(N, nref)
    cs = ncb - p.F
    circ = np.zeros((C, cs)) if circ is None else circ
    sys_len = p.K - p.F - 2 * z
    E = cb_lens(len(llr), C, nl, qm)
    off = np.concatenate([[0], np.cumsum(E)])
    k0 = (_K0[p.bgn][rv] * ncb // N) * z
    idx = (np.arange(E.max()) + k0) % cs
    for r in range(C):
        e = int(E[r])
        x = llr[off[r]:off[r + 1]]
        if len(x) < e:                                                # ldpc.py:1401-1402
            x = np.concatenate([x, np.zeros(e - len(x))])
        x = x.reshape(e // qm, qm).T.reshape(-1)
        for s in range(0, e, cs):                                     # wrap chunks accumulate (ldpc.py:1407-1410)
            t = min(e, s + cs)
            circ[r, idx[s:t]] += x[s:t]
    full = np.concatenate([circ[:, :sys_len], np.full((C, p.F), LARGE_LLR), circ[:, sys_len:]], axis=1)
    return full, circ


def decode(rx, bgn, ils, zc, num_iter=5, only_info=True, belief=False, dtype=np.float64, rows=None):
    """ldpc.py:1495-1581 decode: layered normalised (0.75) min-sum, fixed iteration count, float64.

    QUIRKs kept: clip to +-1e10; sign(0)=+1 via (v<0); first-index argmin; second minimum obtained by adding
    +100000 to the (signed) argmin entry before taking min|.| (ldpc.py:1563); scaling after the un-shift.
    ``dtype=np.float32`` gives the single-precision statement used to bound the fp32 GPU variant.
    ``rows`` (test hook, not in the reference): run only the first `rows` rows of the base graph -- identical for the
    information/core columns whenever the extension columns of the dropped rows carry all-zero LLRs (punctured parity):
    such a row sends +-0 to every core column."""
    T = dtype
    rx = np.clip(np.asarray(rx, dtype=np.float64), -LLR_CLIP, LLR_CLIP).astype(T)
    C = rx.shape[0]
    bg = base_graph(bgn, ils, zc)
    r = np.concatenate([np.zeros((C, 2, zc), dtype=T), rx.reshape(C, -1, zc)], axis=1)
    assert r.shape[1] == bg.shape[1]
    cols = [np.nonzero(row >= 0)[0] for row in bg][:rows]
    msg = [np.zeros((C, len(c), zc), dtype=T) for c in cols]
    ci = np.arange(C)[:, None]
    zi = np.arange(zc)[None, :]
    for _ in range(num_iter):
        for l, cl in enumerate(cols):
            sh = bg[l, cl]
            t = r[:, cl, :] - msg[l]                                       # extrinsic
            ts = np.stack([_rot(t[:, q], sh[q]) for q in range(len(cl))], axis=1)
            neg = ts < 0
            par = (neg.sum(1) & 1).astype(bool)                            # product of signs
            a = np.abs(ts)
            am = np.argmin(a, axis=1)
            m1 = a[ci, am, zi]
            bumped = ts.copy()
            bumped[ci, am, zi] += T(100000)
            m2 = np.abs(bumped).min(axis=1)
            mag = np.repeat(m1[:, None, :], len(cl), axis=1)
            mag[ci, am, zi] = m2
            sgn = np.where(neg ^ par[:, None, :], T(-1), T(1))
            new = mag * sgn
            new = np.stack([_rot(new[:, q], zc - sh[q]) for q in range(len(cl))], axis=1) * T(0.75)
            msg[l] = new
            r[:, cl, :] = t + new
    r = r.reshape(C, -1)
    if only_info:
        r = r[:, :(22 if bgn == 1 else 10) * zc]
    return r if belief else (r < 0).astype(np.int8)


def crc_check_and_merge(dec, p):
    """ldpc.py:1584-1619 checkCrcAndMerge: (C,K) hard bits -> (TB+CRC24A bits, per-block CRC ok)."""
    nf = dec[:, :p.K - p.F]
    if p.C == 1:
        tb = nf.reshape(-1)
        return tb, np.array([bool(crc_check(tb, '24A'))])
    return nf[:, :-24].reshape(-1), crc_check(nf, '24B')


def encode_chain(tb, bgn, G, nl, qm, rv=0, nref=0):
    """ldpc.py:1167-1204 getRateMatchedCodeBlocks (+ the intermediate results for fixtures)."""
    tbc = crc_append(tb, '24A')
    cbs, p = segment(tbc, bgn)
    coded = encode(cbs, bgn, p.iLS, p.Zc)
    return rate_match(coded, p, G, nl, qm, rv, nref), dict(tbc=tbc, cbs=cbs, coded=coded, p=p)
