"""Oracle (CPU, NumPy float64) of the STABILITY CERTIFICATE for early termination.  TEST INFRASTRUCTURE -- see oracle/__init__.py.

The reference runs a fixed number of iterations (ldpc.py:1545); it has no early stop.  The certificate below is evaluated on
the FROZEN state of the reference's own decoder (posteriors r, messages ll[row]; ldpc.py:1546-1576) after some iteration k
and, when it holds, proves that every later iteration of that same float64 recursion leaves every hard decision unchanged
(DESIGN.md 4.3 has the proof).  `decode_certified` is the oracle decoder (oracle/coding.py:decode) with that check evaluated
after chosen iterations; the HIP kernels' certificate flags and certified bits are tested against it and against the full run.
"""
import numpy as np
from . import coding as oc

U53 = 2.0 ** -53
HUGE = 1.0e9                 # |LLR| at or above this is treated as a known bit (fillers arrive as LARGE_LLR clipped to 1e10)


def active_cols(bgn, rows):
    bg = oc.base_graph(bgn, 0, 384)
    return [np.nonzero(r >= 0)[0] for r in bg][:rows], bg.shape[1]


def filler_columns(bgn, zc, F):
    """Base-graph columns that hold at least one filler bit (the last F of the K information bits, ldpc.py:1025-1028)."""
    kb = 22 if bgn == 1 else 10
    return tuple(range((kb * zc - F) // zc, kb)) if F > 0 else ()


def _lfp(cols, ncol, lam):
    """Least fixed point of  V_{i,c} = lam_c + sum_{j != i} U_{j,c},  U_{i,c} = 0.75 min_{k != c} V_{i,k}  (Kleene iteration from 0).
    Returns (V per row, U per row) or None when it does not settle."""
    U = [np.zeros(len(c)) for c in cols]
    for it in range(100000):
        S = np.zeros(ncol)              # sums of the finite messages, and how many infinite ones a column receives
        ninf = np.zeros(ncol, dtype=int)
        for c, u in zip(cols, U):
            f = np.isfinite(u)
            S[c[f]] += u[f]
            ninf[c[~f]] += 1
        change = 0.0
        Vs = []
        for i, c in enumerate(cols):
            f = np.isfinite(U[i])
            V = lam[c] + S[c] - np.where(f, U[i], 0.0)
            V = np.where(ninf[c] - (~f) > 0, np.inf, V)
            Vs.append(V)
            o = np.argsort(V, kind='stable')
            new = np.full(len(c), 0.75 * V[o[0]])
            new[o[0]] = 0.75 * V[o[1]]
            both = np.isfinite(new) & f
            if (np.isfinite(new) != f).any():
                change = np.inf
            elif both.any():
                change = max(change, np.abs(new[both] - U[i][both]).max())
            U[i] = new
        fin = [u[np.isfinite(u)] for u in U]
        if max((x.max() if len(x) else 0.0) for x in fin) > 1e9:
            return None
        if change < 1e-13:
            return Vs, U
    return None


def growth_bounds(bgn, rows, filler_cols=()):
    """A-priori magnitude bounds of the recursion (scale invariant, independent of the lifting size).

    |t_{i,c}| <= |L_c| + sum_{j != i} |m_{j,c}|  and  |m_{i,c}| <= 0.75 min_{k != c} |t_{i,k}|  hold in every iteration whatever the
    signs are (the +1e5 quirk only lowers a second minimum), so all magnitudes stay below the least fixed point of
        V_{i,c} = lam_c + sum_{j != i} U_{j,c},   U_{i,c} = 0.75 min_{k != c} V_{i,k}
    on the base graph with lam_c >= |L_c| (0 for the two punctured columns).  Returns a dict:
      gamma   bound of |r| = |t + m| per unit of max|LLR| over every column that holds no filler (lam = 1 there, infinity on
              the columns with fillers): prices the rounding error;
      gamma1  bound of the smallest |t| of every CORE row (the rows without a degree-1 column) per unit of the maximum over the
              core-parity and extension columns alone (lam = infinity on every transmitted information column): the rate-matching
              interleaver puts the low-order bits of the symbols there, so this maximum is several times smaller;
      dmax    the largest column degree.
    gamma / gamma1 are inf when the iteration does not settle (no certificate then)."""
    cols, ncol = active_cols(bgn, rows)
    kb = 22 if bgn == 1 else 10
    deg = np.zeros(ncol, dtype=int)
    for c in cols:
        deg[c] += 1
    out = dict(gamma=np.inf, gamma1=np.inf, dmax=int(deg.max()))
    lam = np.ones(ncol)
    lam[:2] = 0.0
    for c in filler_cols:
        lam[c] = np.inf
    got = _lfp(cols, ncol, lam)
    if got is not None:
        Vs, U = got
        # |r_c| <= lam_c + sum_j U_{j,c} on every column -- INCLUDING the columns that hold fillers: their non-filler elements carry
        # real LLRs (|L| <= the maximum: lam = 1 for them), so a message of unbounded size into such a column means no bound at all
        # (nrx_ldpc_cert_bounds does the same)
        S = np.zeros(ncol)
        unbounded = np.zeros(ncol, dtype=bool)
        for i, c in enumerate(cols):
            fin = np.isfinite(U[i])
            np.add.at(S, c[fin], U[i][fin])
            unbounded[c[~fin]] = True
        used = deg > 0
        if not unbounded[used].any():
            lam1 = np.where(np.arange(ncol) < 2, 0.0, 1.0)
            out['gamma'] = float((lam1 + S)[used].max()) * (1 + 1e-9)
    lam = np.ones(ncol)
    lam[:2] = 0.0
    lam[2:kb] = np.inf
    got = _lfp(cols, ncol, lam)
    if got is not None:
        Vs, U = got
        g1 = max(Vs[i].min() for i in range(min(4, len(cols))))
        out['gamma1'] = g1 * (1 + 1e-9) if np.isfinite(g1) else np.inf
    return out


def margins(gb, lam_all, lam_pe, num_iter=50):
    """The error budget of the proof.  E bounds |r_c - (L_c + sum_j m_{j,c})| over the whole run: two roundings per
    (row, column) visit, each at most 2^-53 of a magnitude below beta = gamma * max|LLR|.  zeta = the slack of a floor under its
    frozen message, delta = what a future extrinsic value of a core column can lose against the frozen one (kept for reference: the
    relaxation works with per-message slacks), G = the sign margin asked of every posterior, mcap / mcapx = the largest frozen message
    of a core row / of a row that owns a degree-1 column for which the +1e5 quirk (ldpc.py:1563) cannot push a later message under
    its floor: 0.75 * (1e5 - the largest minimum such a row can ever have), that minimum being <= gamma1 * lam_pe resp. <= lam_pe + E."""
    beta = gb['gamma'] * lam_all
    E = 2.0 * num_iter * gb['dmax'] * U53 * beta * 1.0625
    zeta = 4.0 * E
    delta = (gb['dmax'] - 1) * zeta + 2.0 * E
    rmin = max(gb['gamma1'], 1.0) * lam_pe * (1 + 1e-9) + E
    mcap = 0.75 * (1.0e5 - rmin) * (1 - 1e-9) - E                                   # core rows
    mcapx = 0.75 * (1.0e5 - (lam_pe * (1 + 1e-9) + E)) * (1 - 1e-9) - E              # rows that own a degree-1 column
    return dict(beta=beta, E=E, zeta=zeta, delta=delta, G=4.0 * E, mcap=mcap, mcapx=mcapx)


def certify(r, msg, cols, bg, zc, ncore, gb, lam_all, lam_pe, num_iter=50, flags=0, sweeps=6, stats=None):
    """The certificate on the frozen state of ONE code block: r (1, ncol, zc) posteriors, msg[l] (1, deg_l, zc) stored messages in
    column coordinates.  Returns ((1,) bool, dict of per-condition failure flags).
    `flags`: bit 0 skips (S), bit 1 skips (M) -- deliberately BROKEN certificates the tests must catch.

    Everything is normalised by the frozen hard decision of its column, s_c = sign(r_c):  tau = s_c (r_c - m_{i,c})  is what row i
    would read next, nu = s_c m_{i,c} the stored message.  A SLACK w_{i,c} >= 0 per message defines the floor nu - w; W_c is the sum
    of the slacks of the messages into column c and  loss_{i,k} = W_k - w_{i,k} + 2E  what a future tau_{i,k} can lose.
      (B) the a-priori bounds are finite, no LLR outside the filler columns is saturated, mcap > 0
      (Q) |m| <= mcap on every edge
      (S) the hard decisions satisfy every parity check that runs; |r_c| >= W_c + G on every column element; in every check row
          at most ONE edge has tau - loss <= 0
      (M) closure:  0.75 min_{k != c} (tau_k - loss_k) >= nu_c - w_c + E  for every edge c into a core column (the message into a
          row's own degree-1 column needs no floor: nothing else reads that column).
    ANY non-negative w that satisfies (S) and (M) certifies; it is looked for by relaxation: start at w = zeta and raise every
    slack to what (M) asks of it, at most `sweeps` times."""
    C = r.shape[0]
    assert C == 1
    mg = margins(gb, lam_all, lam_pe, num_iter)
    E, zeta, G, mcap = mg['E'], mg['zeta'], mg['G'], mg['mcap']
    ok_b = bool(np.isfinite(mg['beta']) and np.isfinite(mcap) and mcap > 0 and lam_all < HUGE)
    ncol = r.shape[1]
    rows = []
    fail_q = False
    par_bad = False
    for l, cl in enumerate(cols):
        sh = bg[l, cl]
        rs = np.stack([oc._rot(r[0, cl[q]], sh[q]) for q in range(len(cl))])          # (d, zc): the row's view
        ms = np.stack([oc._rot(msg[l][0, q], sh[q]) for q in range(len(cl))])
        sr = np.signbit(rs)
        sg = np.where(sr, -1.0, 1.0)
        ext = cl[-1] >= ncore
        rows.append(dict(cl=cl, sh=sh, tau=(rs - ms) * sg, nu=ms * sg, ext=ext, absr=np.abs(rs)))
        fail_q |= bool((np.abs(ms) > (mg['mcapx'] if ext else mcap)).any())
        par_bad |= bool(((sr.sum(0) & 1) == 1).any())
    w = [np.full(rw['tau'].shape, zeta) for rw in rows]
    for rw, wi in zip(rows, w):
        if rw['ext']:
            wi[-1] = 0.0
    s_ok = m_ok = False
    used = 0
    for sweep in range(sweeps + 1):
        W = np.zeros((ncol, zc))
        for rw, wi in zip(rows, w):
            for q in range(len(rw['cl'])):
                W[rw['cl'][q]] += oc._rot(wi[q], zc - rw['sh'][q])              # back to column coordinates
        s_ok = not par_bad
        m_ok = True
        neww = []
        for rw, wi in zip(rows, w):
            cl, sh = rw['cl'], rw['sh']
            Wr = np.stack([oc._rot(W[cl[q]], sh[q]) for q in range(len(cl))])
            loss = Wr - wi + 2 * E
            tl = rw['tau'] - loss
            s_ok &= not ((tl <= 0).sum(0) > 1).any() and not (rw['absr'] < Wr + G).any()
            o = np.argsort(tl, axis=0, kind='stable')
            a1 = np.take_along_axis(tl, o[:1], 0)[0]
            a2 = np.take_along_axis(tl, o[1:2], 0)[0]
            is_idx = np.arange(len(cl))[:, None] == o[0][None, :]
            lhs = 0.75 * np.where(is_idx, a2[None, :], a1[None, :])
            need = rw['nu'] - lhs + E                                   # the slack (M) asks of every message
            if rw['ext']:
                need[-1] = 0.0
            m_ok &= not (need > wi).any()
            neww.append(np.maximum(wi, need * 1.25 + zeta))
        used = sweep
        if (s_ok or flags & 1) and (m_ok or flags & 2):
            break
        w = neww
    if stats is not None:
        stats['sweeps'] = used
        stats['wmax'] = max(float(x.max()) for x in w)
    ok = ok_b and not fail_q
    if not flags & 1:
        ok = ok and s_ok
    if not flags & 2:
        ok = ok and m_ok
    return np.array([ok]), dict(bound=np.array([not ok_b]), cap=np.array([fail_q]), signs=np.array([not s_ok]), closure=np.array([not m_ok]))


def decode_certified(rx, bgn, ils, zc, num_iter, rows, checks, filler_cols=(), gam=None, flags=0, sweeps=6):
    """oracle/coding.py:decode (= ldpc.py:1495-1581, float64) with the certificate evaluated after the iterations in `checks`.
    Returns the final hard decisions (all columns), and per check: the certificate, the hard decisions at that point, whether
    every parity check of the rows that run is satisfied, and the per-condition failure flags."""
    rx = np.clip(np.asarray(rx, dtype=np.float64), -oc.LLR_CLIP, oc.LLR_CLIP)
    C = rx.shape[0]
    bg = oc.base_graph(bgn, ils, zc)
    ncore = (22 if bgn == 1 else 10) + 4
    r = np.concatenate([np.zeros((C, 2, zc)), rx.reshape(C, -1, zc)], axis=1)
    cols = [np.nonzero(row >= 0)[0] for row in bg][:rows]
    if gam is None:
        gam = growth_bounds(bgn, rows, filler_cols)
    # max |LLR| over what the rows that run can see, columns holding fillers excluded (they are +infinity in the bound), and
    # over the core-parity + extension columns alone
    seen = np.zeros(bg.shape[1], dtype=bool)
    for c in cols:
        seen[c] = True
    pe = seen.copy()
    pe[:ncore - 4] = False
    for c in filler_cols:
        seen[c] = False
    lam_max = np.abs(r[:, seen]).reshape(C, -1).max(1)
    lam_pe = np.abs(r[:, pe]).reshape(C, -1).max(1)
    msg = [np.zeros((C, len(c), zc)) for c in cols]
    ci = np.arange(C)[:, None]
    zi = np.arange(zc)[None, :]
    out = dict(cert={}, bits_at={}, syndrome_ok={}, why={})
    kinfo = (22 if bgn == 1 else 10)
    for it in range(num_iter):
        for l, cl in enumerate(cols):
            sh = bg[l, cl]
            t = r[:, cl, :] - msg[l]
            ts = np.stack([oc._rot(t[:, q], sh[q]) for q in range(len(cl))], axis=1)
            neg = ts < 0
            par = (neg.sum(1) & 1).astype(bool)
            a = np.abs(ts)
            am = np.argmin(a, axis=1)
            m1 = a[ci, am, zi]
            bumped = ts.copy()
            bumped[ci, am, zi] += 100000.0
            m2 = np.abs(bumped).min(axis=1)
            mag = np.repeat(m1[:, None, :], len(cl), axis=1)
            mag[ci, am, zi] = m2
            sgn = np.where(neg ^ par[:, None, :], -1.0, 1.0)
            new = mag * sgn
            new = np.stack([oc._rot(new[:, q], zc - sh[q]) for q in range(len(cl))], axis=1) * 0.75
            msg[l] = new
            r[:, cl, :] = t + new
        k = it + 1
        if k in checks:
            okc = np.zeros(C, dtype=bool)
            why = {}
            for b0 in range(0, C, 64):                      # (per-block lam_max: evaluated block by block)
                for b in range(b0, min(C, b0 + 64)):
                    st = {}
                    o, w = certify(r[b:b + 1], [m[b:b + 1] for m in msg], cols, bg, zc, ncore, gam, lam_max[b], lam_pe[b], num_iter, flags, sweeps, st)
                    out.setdefault('stats', {}).setdefault(k, []).append((bool(o[0]), st.get('sweeps'), st.get('wmax')))
                    okc[b] = o[0]
                    for n, v in w.items():
                        why.setdefault(n, np.zeros(C, dtype=bool))[b] = v[0]
            out['cert'][k] = okc
            out['why'][k] = why
            out['bits_at'][k] = (r[:, :kinfo] < 0).reshape(C, -1).astype(np.int8)
            hard = (r < 0)
            syn = np.zeros(C, dtype=bool)
            for l, cl in enumerate(cols):
                sh = bg[l, cl]
                acc = np.zeros((C, zc), dtype=bool)
                for q in range(len(cl)):
                    acc ^= oc._rot(hard[:, cl[q]], sh[q])
                syn |= acc.any(1)
            out['syndrome_ok'][k] = ~syn
    out['bits'] = (r[:, :kinfo] < 0).reshape(C, -1).astype(np.int8)
    out['lam_max'] = lam_max
    out['lam_pe'] = lam_pe
    return out
