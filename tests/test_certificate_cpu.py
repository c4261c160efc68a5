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
