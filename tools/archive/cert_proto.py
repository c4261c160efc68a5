"""Prototype (CPU, NumPy) of the STABILITY CERTIFICATE for early termination of the layered min-sum decoder.

Runs the oracle's float64 decoder (oracle/coding.py:decode = ldpc.py:1495-1581) on 64-QAM / AWGN LLRs and, after chosen
iterations, evaluates the certificate on the frozen state.  Reports at which check a block certifies and verifies that every
certified block's hard decisions equal those of the full fixed schedule.  Development tool: not imported by the product.

    python tools/archive/cert_proto.py [--zc 384] [--blocks 64] [--snr 17.5] [--rows 15] [--iters 50]
"""
import argparse
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import coding as oc          # noqa: E402
from oracle import certificate as cert   # noqa: E402


def llrs_64qam(bits, snr_db, rng):
    """Max-log LLRs of Gray 64-QAM over AWGN (unit average power); sign convention: positive = bit 0."""
    lev = np.array([-7, -5, -3, -1, 1, 3, 5, 7]) / np.sqrt(42.0)
    # 3 bits per axis, Gray (38.211 5.1.5 per axis: b0 sign, b2 inner/outer, b4)
    g = np.array([[1, 1, 1], [1, 1, 0], [1, 0, 0], [1, 0, 1], [0, 0, 1], [0, 0, 0], [0, 1, 0], [0, 1, 1]])  # level index -> bits
    n = len(bits) // 3
    b = bits[:3 * n].reshape(n, 3)
    idx = np.zeros(n, dtype=int)
    for k in range(8):
        idx[np.all(b == g[k], axis=1)] = k
    nv = 10 ** (-snr_db / 10) / 2          # per real dimension
    y = lev[idx] + rng.standard_normal(n) * np.sqrt(nv)
    d = (y[:, None] - lev[None, :]) ** 2
    out = np.empty((n, 3))
    for q in range(3):
        d0 = np.where(g[:, q] == 0, d, np.inf).min(1)
        d1 = np.where(g[:, q] == 1, d, np.inf).min(1)
        out[:, q] = (d1 - d0) / (2 * nv)
    return out.reshape(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--zc', type=int, default=384)
    ap.add_argument('--bgn', type=int, default=1)
    ap.add_argument('--blocks', type=int, default=32)
    ap.add_argument('--snr', type=float, default=17.0)
    ap.add_argument('--rows', type=int, default=15)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--fillers', type=int, default=0)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--every', type=int, default=1)
    ap.add_argument('--real', default='', help='npz of tools/r4/llr_stats.py (real LLRs of the metric configuration)')
    ap.add_argument('--key', default='31.0')
    ap.add_argument('--flags', type=int, default=0)
    ap.add_argument('--sweeps', type=int, default=6)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    if a.real:
        d = np.load(a.real)
        a.zc, a.rows, a.fillers, a.bgn = int(d['Zc']), int(d['rows']), int(d['F']), 1
        llr_real = d['llr_' + a.key][:a.blocks]
        info_real = d['cbs_' + a.key][:a.blocks].astype(np.int8)
        a.blocks = len(llr_real)
    ils = [i for i, zs in enumerate(oc.LIFTING_SETS) if a.zc in zs][0]
    kb = 22 if a.bgn == 1 else 10
    ncore = kb + 4
    K = kb * a.zc
    info = rng.integers(0, 2, (a.blocks, K)).astype(np.int8)
    if a.fillers:
        info[:, K - a.fillers:] = 0
    coded = oc.encode(info, a.bgn, ils, a.zc)                # (C, N) punctured
    ncols_rx = ncore - 2 + (a.rows - 4)                       # received columns: core (minus the 2 punctured) + live extensions
    n_tx = ncols_rx * a.zc
    llr = np.zeros(coded.shape, dtype=np.float64)
    for c in range(a.blocks):
        tx = coded[c, :n_tx]
        pad = (-len(tx)) % 3
        l = llrs_64qam(np.concatenate([tx, np.zeros(pad, dtype=np.int8)]), a.snr, rng)[:n_tx]
        llr[c, :n_tx] = l
    if a.fillers:
        llr[:, K - 2 * a.zc - a.fillers:K - 2 * a.zc] = 1e10
    if a.real:
        info = info_real
        llr[:] = 0
        llr[:, :llr_real.shape[1]] = llr_real
    filler_cols = cert.filler_columns(a.bgn, a.zc, a.fillers)
    gam = cert.growth_bounds(a.bgn, a.rows, filler_cols)
    print(gam, f'|LLR| max {np.abs(llr[np.abs(llr) < 1e9]).max():.1f}')
    checks = list(range(a.every, a.iters, a.every))
    res = cert.decode_certified(llr, a.bgn, ils, a.zc, a.iters, a.rows, checks, filler_cols, gam, a.flags, a.sweeps)
    final = res['bits']
    ok_final = (final == info).all(1)
    print(f'blocks {a.blocks}: decoded correctly by the full run {ok_final.sum()}')
    first = np.full(a.blocks, -1)
    synd = np.full(a.blocks, -1)
    bad = 0
    for k in checks:
        c = res['cert'][k]
        s = res['syndrome_ok'][k]
        for b in range(a.blocks):
            if s[b] and synd[b] < 0:
                synd[b] = k
            if c[b] and first[b] < 0:
                first[b] = k
            if c[b] and not np.array_equal(res['bits_at'][k][b], final[b]):
                bad += 1
    print('certified-but-different (must be 0):', bad)
    print('first syndrome-zero iteration histogram:', np.bincount(synd[synd >= 0], minlength=1))
    print('first certified iteration histogram:    ', np.bincount(first[first >= 0], minlength=1))
    print('never certified:', (first < 0).sum(), ' of which decoded OK:', ((first < 0) & ok_final).sum())
    lag = first[(first >= 0) & (synd >= 0)] - synd[(first >= 0) & (synd >= 0)]
    if len(lag):
        print('lag certificate - syndrome: mean %.2f max %d' % (lag.mean(), lag.max()))
    for k in checks[:24]:
        st = res['stats'][k]
        sw = [x[1] for x in st if x[0]]
        print(k, 'certified', len(sw), 'sweeps used', np.bincount(sw) if sw else [], 'max slack', max([x[2] for x in st if x[0]], default=0))
    if 'why' in res:
        for k in checks[:20]:
            print(k, {n: int(v.sum()) for n, v in res['why'][k].items()})


if __name__ == '__main__':
    main()
