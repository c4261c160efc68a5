"""CPU: the early-termination certificate at the oracle level (oracle/certificate.py) and the host side of the library's
(nrx_ldpc_cert_bounds).  The reference has no early stop (ldpc.py:1545): what is pinned here is the PROPERTY the certificate
claims -- a block it certifies after iteration k has, at k, exactly the hard decisions the reference's full fixed schedule ends
on -- on the reference's own recursion (oracle/coding.py:decode, itself pinned bit for bit by tests/test_oracle_golden.py), for
both base graphs, several lifting sizes, filler bits, exact zeros and saturated LLRs; and that a broken certificate is caught."""
import numpy as np
import pytest

from oracle import coding as oc
from oracle import certificate as cert


def _llrs(bgn, zc, rows, blocks, sigma, rng, fillers=0):
    ils = [i for i, zs in enumerate(oc.LIFTING_SETS) if zc in zs][0]
    kb = 22 if bgn == 1 else 10
    K = kb * zc
    info = rng.integers(0, 2, (blocks, K)).astype(np.int8)
    if fillers:
        info[:, K - fillers:] = 0
    coded = oc.encode(info, bgn, ils, zc)
    n_tx = (kb + 4 - 2 + rows - 4) * zc
    llr = np.zeros(coded.shape)
    bits = coded[:, :n_tx].astype(np.float64)
    llr[:, :n_tx] = (2 / sigma ** 2) * ((1 - 2 * bits) + sigma * rng.standard_normal(bits.shape))
    llr[:, :n_tx][rng.random(bits.shape) < 0.002] = 0.0                 # exact zeros among the received LLRs
    if fillers:
        llr[:, K - 2 * zc - fillers:K - 2 * zc] = oc.LARGE_LLR
    return ils, info, llr


@pytest.mark.parametrize("bgn,zc,rows,sigma,fillers", [
    (1, 16, 15, 0.62, 0), (1, 52, 13, 0.58, 20), (1, 96, 22, 0.75, 0),
    (2, 16, 15, 0.95, 0), (2, 52, 10, 0.70, 30), (2, 96, 22, 1.10, 0),
])
def test_certified_bits_are_the_full_runs_bits(bgn, zc, rows, sigma, fillers):
    rng = np.random.default_rng(1000 * bgn + zc + rows)
    blocks = 24
    ils, info, llr = _llrs(bgn, zc, rows, blocks, sigma, rng, fillers)
    fcols = cert.filler_columns(bgn, zc, fillers)
    checks = [4, 6, 8, 12, 16, 24]
    res = cert.decode_certified(llr, bgn, ils, zc, 30, rows, checks, fcols, sweeps=12)
    # the decoder inside is the oracle's, bit for bit
    ref = oc.decode(llr, bgn, ils, zc, num_iter=30, rows=rows)
    assert np.array_equal(res['bits'], ref)
    n_cert = 0
    for k in checks:
        c = res['cert'][k]
        n_cert += int(c.sum())
        assert np.array_equal(res['bits_at'][k][c], res['bits'][c]), f"a block certified after iteration {k} changed later"
        assert not (c & ~res['syndrome_ok'][k]).any(), "certified with an unsatisfied parity check"
    converged = res['syndrome_ok'][checks[-1]]
    assert converged.any() and n_cert > 0, "want blocks that converge and certify"
    # every block that converged early enough holds a certificate by the last check
    early = res['syndrome_ok'][checks[2]]
    assert res['cert'][checks[-1]][early].all()


def test_saturated_llrs_are_refused_not_mishandled():
    """An LLR at the clip (1e10) outside the filler positions voids the a-priori bounds: no certificate, whatever the state."""
    rng = np.random.default_rng(5)
    ils, info, llr = _llrs(1, 32, 15, 6, 0.5, rng)
    llr[:, 7] = 1e10 * (1 - 2 * info[:, 7 + 2 * 32])
    res = cert.decode_certified(llr, 1, ils, 32, 12, 15, [6, 10], (), sweeps=12)
    assert not res['cert'][6].any() and not res['cert'][10].any()
    assert res['syndrome_ok'][10].all()


def test_a_broken_certificate_is_caught():
    """flags = 3 drops the sign and closure conditions: blocks that have not converged are 'certified' and differ from the full run."""
    rng = np.random.default_rng(9)
    ils, info, llr = _llrs(1, 48, 15, 16, 0.78, rng)
    res = cert.decode_certified(llr, 1, ils, 48, 30, 15, [2, 3], (), flags=3, sweeps=2)
    bad = sum(int((res['bits_at'][k][res['cert'][k]] != res['bits'][res['cert'][k]]).any(1).sum()) for k in (2, 3))
    assert res['cert'][2].all() and bad > 0


def test_the_magnitude_condition_alone_refuses_converged_blocks():
    """(Q) BINDS: clean code words at LLR magnitudes of a few thousand -- (B) still holds, Lambda_pe < 1e5 / gamma_1 -- converge at
    once, every parity check satisfied, (S) and (M) hold, and the stored messages exceed mcap = 0.75 (1e5 - gamma_1 Lambda_pe): the +1e5 quirk's
    territory (ldpc.py:1563).  The certificate refuses them for that reason alone; the same noise at ordinary magnitudes certifies.  (The
    refused blocks' bits do equal the final ones: the conditions are sufficient, not necessary.)"""
    rng = np.random.default_rng(2)
    bgn, zc, rows = 1, 16, 15
    ils, info, llr = _llrs(bgn, zc, rows, 12, 0.6, rng)
    big = llr * (1400.0 / (2 / 0.6 ** 2))      # largest parity LLR about 4 500 < 1e5 / gamma_1 = 4 694
    res = cert.decode_certified(big, bgn, ils, zc, 50, rows, [8, 12], (), sweeps=12)
    n_only_q = 0
    for k in (8, 12):
        why = res['why'][k]
        only_q = res['syndrome_ok'][k] & why['cap'] & ~why['signs'] & ~why['closure'] & ~why['bound']
        n_only_q += int(only_q.sum())
        assert not (res['cert'][k] & why['cap']).any()
        assert np.array_equal(res['bits_at'][k][only_q], res['bits'][only_q])
    assert n_only_q >= 4, "want converged blocks that (Q) alone refuses"
    small = cert.decode_certified(llr, bgn, ils, zc, 50, rows, [8, 12], (), sweeps=12)
    assert small['cert'][12].sum() >= 8 and not small['why'][12]['cap'].any()


def _weak_posterior(base, bits, bgn, ils, zc, rows, k, pos, width):
    """The LLR of transmitted information element `pos` moved against its bit until the posterior of that element after k iterations is
    positive (in the bit's own sign) but below `width`: bisection on the oracle's decoder.  None when the element cannot be brought there."""
    sgn = 1 - 2 * bits[0, pos]

    def post(L):
        x = base.copy()
        x[0, pos] = L * sgn
        return oc.decode(x, bgn, ils, zc, num_iter=k, rows=rows, only_info=False, belief=True)[0, pos + 2 * zc] * sgn, x
    lo, hi = -60 * np.abs(base).max(), 0.0
    if post(lo)[0] > 0 or post(hi)[0] < 0:
        return None
    best = None
    for _ in range(80):
        mid = 0.5 * (lo + hi)
        f, x = post(mid)
        if f > width[1]:
            hi = mid
        elif f < width[0]:
            lo = mid
        else:
            best = (x, f)
            break
    return best


def test_the_posterior_margin_alone_refuses_a_converged_block():
    """(S) BINDS, construction: in a block that holds the full certificate after 16 iterations, one information LLR is
    moved against its bit until that element's posterior after iteration 16 is +4e-9 (2 ... 6e-9; the margin W_c + G is a few 1e-8 here).
    Every parity check is still satisfied, (Q) and the closure condition (M) still hold, and rho_c >= W_c + G fails: refused by (S)
    alone -- flags = 1 (no (S)) certifies it.  In THIS state the bits do not change afterwards (the messages into the element only grow);
    the next test shows a state where they do."""
    rng = np.random.default_rng(3)
    bgn, zc, rows, k = 1, 16, 15, 16
    ils = [i for i, zs in enumerate(oc.LIFTING_SETS) if zc in zs][0]
    K = 22 * zc
    n_tx = (22 + 4 - 2 + rows - 4) * zc
    info = rng.integers(0, 2, (1, K)).astype(np.int8)
    bits = oc.encode(info, bgn, ils, zc)[:, :n_tx].astype(np.float64)
    base = np.zeros((1, (66) * zc))
    base[:, :n_tx] = (2 / 0.6 ** 2) * ((1 - 2 * bits) + 0.6 * rng.standard_normal(bits.shape))
    assert cert.decode_certified(base, bgn, ils, zc, 50, rows, [k], (), sweeps=12)['cert'][k][0]
    got = _weak_posterior(base, bits, bgn, ils, zc, rows, k, 5 * zc + 3, (2e-9, 6e-9))
    assert got is not None
    x, rho = got
    full = cert.decode_certified(x, bgn, ils, zc, 50, rows, [k], (), sweeps=12)
    no_s = cert.decode_certified(x, bgn, ils, zc, 50, rows, [k], (), flags=1, sweeps=12)
    why = {n: bool(v[0]) for n, v in full['why'][k].items()}
    assert full['syndrome_ok'][k][0] and why == dict(bound=False, cap=False, signs=True, closure=False), why
    assert not full['cert'][k][0] and no_s['cert'][k][0]
    assert np.array_equal(full['bits_at'][k], full['bits'])


def test_a_bit_that_flips_after_every_check_passed_is_refused_by_the_closure_condition():
    """What the conditions are FOR: a block whose parity checks (and CRC) all pass after iteration k while one information bit is held by a
    posterior of 1e-12 ... 1e-8 -- right after convergence some messages into it still shrink, and after iteration k + 1 (the fixed
    schedule's last one here) the bit has flipped.  Found by search on the oracle's recursion (about one weak element in ten).  Without the
    closure condition (flags 2, flags 3) such a block 'certifies' at k and its bits are NOT the fixed schedule's: at the initial slacks the
    posterior margin of (S) still holds, it is (M)'s demand on the slacks of the rows that read the weak element that never closes.  The
    full certificate and (M) alone (flags 1) refuse it.  [A state in which (M) closes, (S) alone refuses AND the bits change later was
    searched for (tools/r6/sq_search.py: 1.5e3 weak elements) and not found: (M)'s slacks are twice what a message can lose, so a posterior
    above none of them has rows whose demands do not close either -- DESIGN 4.3.]"""
    rng = np.random.default_rng(11)
    bgn, zc, rows = 1, 16, 15
    ils = [i for i, zs in enumerate(oc.LIFTING_SETS) if zc in zs][0]
    K = 22 * zc
    n_tx = (22 + 4 - 2 + rows - 4) * zc
    found = 0
    for trial in range(40):
        sigma = 0.68
        info = rng.integers(0, 2, (1, K)).astype(np.int8)
        bits = oc.encode(info, bgn, ils, zc)[:, :n_tx].astype(np.float64)
        base = np.zeros((1, 66 * zc))
        base[:, :n_tx] = (2 / sigma ** 2) * ((1 - 2 * bits) + sigma * rng.standard_normal(bits.shape))
        k = int(rng.choice([10, 12]))
        b0 = oc.decode(base, bgn, ils, zc, num_iter=k, rows=rows)
        if not np.array_equal(b0, info):
            continue
        for _ in range(3):
            pos = int(rng.integers(0, K - 2 * zc))
            got = _weak_posterior(base, bits, bgn, ils, zc, rows, k, pos, (1e-12, 1e-8))
            if got is None:
                continue
            x, rho = got
            at_k = oc.decode(x, bgn, ils, zc, num_iter=k, rows=rows)
            fin = oc.decode(x, bgn, ils, zc, num_iter=k + 1, rows=rows)
            if not np.array_equal(at_k, info) or np.array_equal(fin, at_k):
                continue
            r = {f: cert.decode_certified(x, bgn, ils, zc, k + 1, rows, [k], (), flags=f, sweeps=12) for f in (0, 1, 2, 3)}
            if not r[0]['syndrome_ok'][k][0]:      # (a parity bit is wrong at k: the CRC-then-certificate path never sees this block)
                continue
            assert r[3]['cert'][k][0] and not np.array_equal(r[3]['bits_at'][k], r[3]['bits'])      # no conditions: certified, and wrong
            assert r[2]['cert'][k][0] and not np.array_equal(r[2]['bits_at'][k], r[2]['bits'])      # (S) at the initial slacks does not see it
            assert not r[1]['cert'][k][0] and r[1]['why'][k]['closure'][0]                            # (M) alone refuses it
            assert not r[0]['cert'][k][0] and r[0]['why'][k]['closure'][0]
            found += 1
        if found >= 2:
            break
    assert found >= 2, "want states whose bits change after the check"


def test_witness_without_the_posterior_margin_a_block_is_stopped_on_bits_that_change():
    """The witness VERDICT r5 #9 asked for (tests/golden/cert_witness_S.npz, found by tools/r6/sq_search.py seed 14: LLRs of one BG1 / Zc 16
    code block, 15 rows, one information LLR moved against its bit): after iteration k = 12 every parity check passes and the decoded
    information bits are the transmitted ones; the weak element's posterior is +1.9e-9.  (Q) holds, the closure condition (M) CLOSES, and
    rho_c >= W_c + G fails: (S) alone refuses.  After iteration 13 -- the fixed schedule's last one with numIter = 13 -- that bit has
    flipped: a certificate without (S) (flags 1) certifies the block at 12 on bits that are NOT the fixed schedule's."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'cert_witness_S.npz'))
    x, k, ni, zc, rows, bgn = g['llr'], int(g['k']), int(g['num_iter']), int(g['zc']), int(g['rows']), int(g['bgn'])
    ils = [i for i, zs in enumerate(oc.LIFTING_SETS) if zc in zs][0]
    assert ni == k + 1
    full = cert.decode_certified(x, bgn, ils, zc, ni, rows, [k], (), sweeps=12)
    no_s = cert.decode_certified(x, bgn, ils, zc, ni, rows, [k], (), flags=1, sweeps=12)
    why = {n: bool(v[0]) for n, v in full['why'][k].items()}
    assert full['syndrome_ok'][k][0] and np.array_equal(full['bits_at'][k], g['info'])
    assert why == dict(bound=False, cap=False, signs=True, closure=False), why
    assert not full['cert'][k][0]
    assert no_s['cert'][k][0]
    changed = np.nonzero((no_s['bits_at'][k] != no_s['bits'])[0])[0]
    assert changed.tolist() == [int(g['pos']) + 2 * zc]                      # exactly the weak element, one iteration later
    assert np.array_equal(no_s['bits'], oc.decode(x, bgn, ils, zc, num_iter=ni, rows=rows))


def test_library_bounds_match_the_oracle():
    from neoradium_amd import _lib, ops
    for bgn, B, rows in [(1, 606504 + 24, 15), (1, 606504 + 24, 13), (1, 606504 + 24, 46), (1, 25000 + 24, 15), (2, 3000, 22), (2, 3000, 42),
                         (2, 640, 10), (1, 10024, 31)]:
        cfg = _lib.ldpc_config(bgn, B)
        g, g1, d = ops.ldpc_cert_bounds(cfg, rows)
        ref = cert.growth_bounds(bgn, rows, cert.filler_columns(bgn, cfg.Zc, cfg.F))
        assert d == ref['dmax']
        assert np.isfinite(g) and abs(g - ref['gamma']) <= 1e-9 * ref['gamma'], (bgn, rows, g, ref)
        assert np.isfinite(g1) and abs(g1 - ref['gamma1']) <= 1e-9 * ref['gamma1'], (bgn, rows, g1, ref)
    with pytest.raises(ValueError):
        ops.ldpc_cert_bounds(_lib.ldpc_config(1, 10024), 3)
