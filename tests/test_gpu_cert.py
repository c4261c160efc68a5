"""GPU: the certified early exit (nrx_ldpc_stage_decode_merge_f64 + nrx_ldpc_certify_f64) through the C ABI.

The reference runs a fixed number of iterations (ldpc.py:1545); a block the certificate stops early must carry exactly the bits
the fixed schedule ends on.  Checked bit for bit against the fixed schedule of the same library (itself bit-identical to the
oracle, tests/test_gpu_ldpc.py), over >= 1e5 code blocks across the waterfall at the metric configuration, with filler bits,
exact zeros and saturated LLRs, for both on-chip instantiations (13 / 15 rows); against the oracle's own certificate on a
slot; and a deliberately broken certificate must be caught."""
import numpy as np
import pytest

from oracle import coding as oc
from oracle import certificate as cert

pytestmark = pytest.mark.gpu


def _metric_link(**kw):
    import neoradium_amd as nr
    import bench
    return bench.build_link(nr, decoder="f64", num_iter=50, **kw)


@pytest.mark.parametrize("in_kernel", ["persistent", True, False])
def test_certified_blocks_equal_the_fixed_schedule_over_the_waterfall(dev, in_kernel):
    """>= 1e5 code blocks, 29 ... 35 dB (code-block error rate ~0.6 ... 0): transport-block bits and CRC verdicts of the certified
    schedule identical to the fixed 50-iteration schedule for EVERY block, certified or not -- as ONE persistent launch (the default), as
    staged launches with the certificate in the stage kernel's tail, and with the certificate as its own launch on the parked states."""
    import torch
    fixed = _metric_link()
    certd = _metric_link(certifiedExit=(8, 14, 24), certInKernel=bool(in_kernel), certPersistent=in_kernel == "persistent")
    C, pay = fixed.cfg.C, fixed.cfg.cb_len - 24
    total = certified = 0
    hist = {}
    for k, snr in enumerate((29.0, 30.0, 31.0, 31.5, 32.0, 33.0, 35.0)):
        for b in range(2):
            n = 100
            slot0 = 7000 * k + 300 * b
            _, d0 = fixed.run(slot0, n, snr, seed=11, details="verdicts")
            _, d1 = certd.run(slot0, n, snr, seed=11, details="verdicts")
            ex = certd.last_exit_iter
            assert torch.equal(d0[0][1]['cb_ok'], d1[0][1]['cb_ok']), f"CRC verdicts differ at {snr} dB"
            same = (d0[0][1]['tb_out'].reshape(-1, pay) == d1[0][1]['tb_out'].reshape(-1, pay)).all(1)
            bad = (~same).nonzero().reshape(-1)
            assert bad.numel() == 0, f"{bad.numel()} blocks differ at {snr} dB (exit iterations {ex[bad][:8].tolist()})"
            total += ex.numel()
            certified += int((ex > 0).sum())
            for v, c in zip(*np.unique(ex.cpu().numpy(), return_counts=True)):
                hist[int(v)] = hist.get(int(v), 0) + int(c)
    assert total >= 100000
    assert certified > 0.5 * total and set(hist) >= {0, 8, 14}, hist


def test_persistent_schedule_equals_the_staged_launches(dev):
    """The certified schedule as ONE launch (nrx_ldpc_certified_persistent_f64: code-block slots that draw blocks from a device queue and take
    each through all its stages behind barriers of their own) against the staged launches with the certificate in the kernel's tail: the
    same certificate code on the same frozen states, so the same exit iteration for every block, the same bits, the same CRC verdicts --
    and both equal to the fixed schedule.  Across the waterfall, with one, two and three checks, and on a batch smaller than the grid."""
    import torch
    from neoradium_amd import ops
    fixed = _metric_link()
    pay = fixed.cfg.cb_len - 24
    total = 0
    for stages, snr, slot0, n in (((8, 16), 31.0, 50, 40), ((8,), 29.5, 400, 16), ((6, 10, 18), 33.0, 900, 24), ((8, 16), 35.0, 1300, 3)):
        staged = _metric_link(certifiedExit=stages, certPersistent=False)
        pers = _metric_link(certifiedExit=stages)
        assert pers.certPersistent and not staged.certPersistent
        _, d0 = fixed.run(slot0, n, snr, seed=5, details="verdicts")
        _, d1 = staged.run(slot0, n, snr, seed=5, details="verdicts")
        _, d2 = pers.run(slot0, n, snr, seed=5, details="verdicts")
        assert ops.persistent_error() == 0, "a slot barrier of the persistent kernel gave up"
        e1, e2 = staged.last_exit_iter.cpu().numpy(), pers.last_exit_iter.cpu().numpy()
        assert np.array_equal(e1, e2), (stages, snr, int((e1 != e2).sum()))
        for d in (d1, d2):
            assert torch.equal(d0[0][1]['cb_ok'], d[0][1]['cb_ok'])
            assert torch.equal(d0[0][1]['tb_out'].reshape(-1, pay), d[0][1]['tb_out'].reshape(-1, pay))
        total += e2.size
        assert set(np.unique(e2)) <= {0, *stages}
    assert total > 5000


def test_persistent_entry_argument_errors(dev):
    """nrx_ldpc_certified_persistent_f64 refuses what it cannot run, like the staged entries: more than three checks (the ops layer), a
    stage without iterations, NULL buffers (NRX_E_ARG -> ValueError), and a configuration without a fused instantiation (NRX_E_UNSUPPORTED ->
    the ops layer returns None and the caller takes the separate stages)."""
    import ctypes as C
    import torch
    from neoradium_amd import ops, _lib
    cfg = _lib.ldpc_config(1, 25000 + 24)
    G = cfg.C * 12300
    llr = torch.zeros((2, G), dtype=torch.float64, device=dev)
    with pytest.raises(ValueError):
        ops.ldpc_recover_decode_merge_certified(llr, cfg, 1, 2, (4, 6, 8, 10), 30, rows=15, persistent=True)
    L = _lib.lib()
    bad = (C.c_int32 * 2)(8, 0)
    buf = torch.zeros(64, dtype=torch.uint8, device=dev)
    rc = L.nrx_ldpc_certified_persistent_f64(_lib.ptr(llr), 2, G, C.byref(cfg), 1, 2, bad, 2, 15, _lib.ptr(buf), _lib.ptr(buf), _lib.ptr(buf),
                                             _lib.ptr(buf), _lib.ptr(buf), _lib.ptr(buf), 64, _lib.ptr(buf), 4, 0, None)
    assert rc == -1 and 'stage' in _lib.last_error()
    ok = (C.c_int32 * 2)(8, 8)
    rc = L.nrx_ldpc_certified_persistent_f64(None, 2, G, C.byref(cfg), 1, 2, ok, 2, 15, _lib.ptr(buf), _lib.ptr(buf), _lib.ptr(buf),
                                             _lib.ptr(buf), _lib.ptr(buf), _lib.ptr(buf), 64, _lib.ptr(buf), 4, 0, None)
    assert rc == -1
    cfg2 = _lib.ldpc_config(2, 3000)                       # BG2: no fused instantiation
    assert ops.ldpc_recover_decode_merge_certified(torch.zeros((1, 9000), dtype=torch.float64, device=dev), cfg2, 1, 2, (4,), 10, persistent=True) is None


def test_a_broken_certificate_is_caught(dev):
    """Without its conditions (flags 7: no sign / closure conditions, CRC filter off) everything 'certifies' at the first check
    and blocks that had not converged differ from the fixed schedule: the comparison above would fail."""
    fixed = _metric_link()
    pay = fixed.cfg.cb_len - 24
    _, d0 = fixed.run(50, 16, 30.5, seed=2, details="verdicts")
    for in_kernel in (True, False):
        broken = _metric_link(certifiedExit=(8, 16), certFlags=7, certInKernel=in_kernel)
        _, d1 = broken.run(50, 16, 30.5, seed=2, details="verdicts")
        assert (broken.last_exit_iter == 8).all()
        diff = (d0[0][1]['tb_out'].reshape(-1, pay) != d1[0][1]['tb_out'].reshape(-1, pay)).any(1)
        assert int(diff.sum()) > 0, in_kernel


def test_conditions_alone_on_blocks_whose_crc_passes(dev):
    """flags = 1 (no sign / posterior / magnitude conditions (S), (Q)) and flags = 2 (no closure condition (M)) ALONE, with the CRC
    filter on, at the iteration-8 check.  Every block the full certificate takes is taken without a condition as well; the closure
    condition on its own REFUSES blocks whose CRC passes at 8 (it is what keeps a just-converged block running: strictly more blocks
    stop without it), while (S) / (Q) refuse none of these CRC-passing blocks that (M) would let through -- their work is on blocks the
    CRC filter has already removed (test_a_broken_certificate_is_caught, flags 7).  Both forms; the complete certificate's bits equal the
    fixed schedule's on every block."""
    fixed = _metric_link()
    pay = fixed.cfg.cb_len - 24
    _, d0 = fixed.run(50, 24, 31.0, seed=2, details="verdicts")
    ok8 = None
    for in_kernel in (True, False):
        taken = {}
        for flags in (0, 1, 2, 3):
            link = _metric_link(certifiedExit=(8,), certFlags=flags, certInKernel=in_kernel)
            _, d1 = link.run(50, 24, 31.0, seed=2, details="verdicts")
            taken[flags] = (link.last_exit_iter == 8).cpu().numpy()
            if flags == 0:
                assert (d0[0][1]['tb_out'].reshape(-1, pay) == d1[0][1]['tb_out'].reshape(-1, pay)).all()
        assert taken[0].sum() > 0
        for flags in (1, 2, 3):
            assert (taken[flags] | ~taken[0]).all(), (in_kernel, flags)                   # nothing certified is lost
        assert taken[2].sum() > taken[0].sum(), (in_kernel, int(taken[2].sum()), int(taken[0].sum()))      # (M) binds on CRC-passing blocks
        assert (taken[3] | ~taken[2]).all() and (taken[3] | ~taken[1]).all()
        # with no condition left the certified set is the set of blocks whose CRC passes at iteration 8: the same for both forms
        ok8 = taken[3] if ok8 is None else ok8
        assert np.array_equal(taken[3], ok8)


@pytest.mark.parametrize("tbs,qm,nl,e_bits", [(25000, 6, 4, 13000), (25000, 2, 1, 12300), (33000, 4, 2, 13000)])
def test_fillers_zeros_saturation_and_both_instantiations(dev, tbs, qm, nl, e_bits):
    """Synthetic LLRs straight into the entries: F > 0 (the filler positions' posteriors sit at 1e10), exact zeros, one clean / one
    marginal / one hopeless transport block, E_r reaching 15 or only 13 rows; then the same with a saturated LLR (1e10) outside
    the filler positions in every block: the certificate must refuse those blocks, the bits stay those of the fixed schedule."""
    import torch
    from neoradium_amd import ops, _lib
    cfg = _lib.ldpc_config(1, tbs + 24)
    assert cfg.Zc == 384 and cfg.C > 1 and cfg.F > 0
    n_tb = 6
    rng = np.random.default_rng(tbs + qm)
    e_small = (e_bits // (nl * qm)) * (nl * qm)
    G = cfg.C * e_small
    lens = _lib.ldpc_cb_lens(G, cfg.C, nl, qm)
    tb = torch.from_numpy(rng.integers(0, 2, (n_tb, tbs)).astype(np.uint8)).to(dev)
    coded = ops.ldpc_encode(ops.ldpc_segment(tb, cfg), cfg)
    bits = ops.ldpc_rate_match(coded, cfg, G, nl, qm).cpu().numpy().astype(np.float64)
    sig = np.array([0.45, 0.5, 0.74, 0.76, 0.78, 1.1])[:, None]
    llr = (2 / sig ** 2) * ((1 - 2 * bits) + sig * rng.standard_normal(bits.shape))
    llr[rng.random(llr.shape) < 0.001] = 0.0
    rows = ops.ldpc_active_rows(cfg, max(lens))
    assert rows <= 15 and ops.ldpc_fused_supported(cfg, nl, qm, G, rows)

    def deint(a):
        out = np.empty_like(a)
        off = 0
        for E in lens:
            out[:, off:off + E] = a[:, off:off + E].reshape(n_tb, E // qm, qm).transpose(0, 2, 1).reshape(n_tb, E)
            off += E
        return torch.from_numpy(out).to(dev)

    n_iter = 30
    for saturate in (False, True):
        x = llr.copy()
        if saturate:
            off = 0
            for E in lens:                      # one received LLR per code block at the clip, with the right sign
                x[:, off + 5] = 1e10 * (1 - 2 * bits[:, off + 5])
                off += E
        xd = deint(x)
        tb_ref, ok_ref = ops.ldpc_recover_decode_merge(xd, cfg, nl, qm, n_iter, rows=rows)
        for in_kernel in (True, False, "persistent"):
            tb_c, ok_c, ex = ops.ldpc_recover_decode_merge_certified(xd, cfg, nl, qm, (5, 9, 14), n_iter, rows=rows, in_kernel=bool(in_kernel),
                                                                     persistent=in_kernel == "persistent")
            assert ops.persistent_error() == 0
            assert torch.equal(ok_c, ok_ref) and torch.equal(tb_c, tb_ref), (saturate, in_kernel)
            exn = ex.cpu().numpy().reshape(n_tb, cfg.C)
            okn = ok_ref.cpu().numpy().astype(bool)
            if saturate:
                assert (exn == 0).all(), "a block with a saturated LLR outside the fillers was certified"
            else:
                # (a marginal block may pass its CRC only after the last check: it runs to the end)
                assert (exn[okn] > 0).mean() > 0.5 and (exn[~okn] == 0).all(), (exn, okn, in_kernel)
                assert 0 < okn.sum() < okn.size


def test_the_sign_and_magnitude_conditions_bind_on_converged_blocks(dev):
    """(S) / (Q) BIND on blocks whose CRC passes: clean code words at LLR magnitudes of a few thousand -- (B) holds, the
    largest parity LLR stays under 1e5 / gamma_1 = 4 694 -- converge at once and satisfy the closure condition (M), but their stored
    messages exceed mcap = 0.75 (1e5 - gamma_1 Lambda_pe): the territory of the +1e5 quirk (ldpc.py:1563).  The complete certificate
    (flags 0) and the one without (M) (flags 2) refuse every block; without (S) / (Q) (flags 1) every block whose CRC passes stops at
    the first check.  The same noise at ordinary magnitudes certifies.  Both forms; all bits those of the fixed schedule.
    (CPU, on the oracle's certificate: tests/test_certificate_cpu.py -- (Q) alone, (S) alone by construction, and the state in which
    a certificate without (S) stops a block whose bit flips one iteration later.)"""
    import torch
    from neoradium_amd import ops, _lib
    tbs, qm, nl = 25000, 2, 1
    cfg = _lib.ldpc_config(1, tbs + 24)
    n_tb = 4
    rng = np.random.default_rng(77)
    e_small = (12300 // (nl * qm)) * (nl * qm)
    G = cfg.C * e_small
    lens = _lib.ldpc_cb_lens(G, cfg.C, nl, qm)
    rows = ops.ldpc_active_rows(cfg, max(lens))
    assert rows <= 15 and ops.ldpc_fused_supported(cfg, nl, qm, G, rows)
    tb = torch.from_numpy(rng.integers(0, 2, (n_tb, tbs)).astype(np.uint8)).to(dev)
    bits = ops.ldpc_rate_match(ops.ldpc_encode(ops.ldpc_segment(tb, cfg), cfg), cfg, G, nl, qm).cpu().numpy().astype(np.float64)
    unit = (1 - 2 * bits) + 0.55 * np.clip(rng.standard_normal(bits.shape), -3.5, 3.5)           # |unit| <= 2.93

    def deint(a):
        out = np.empty_like(a)
        off = 0
        for E in lens:
            out[:, off:off + E] = a[:, off:off + E].reshape(n_tb, E // qm, qm).transpose(0, 2, 1).reshape(n_tb, E)
            off += E
        return torch.from_numpy(out).to(dev)

    n_iter = 30
    _, gamma1, _ = ops.ldpc_cert_bounds(cfg, rows)
    top = 0.985e5 / gamma1                                  # largest |LLR| just under 1e5 / gamma_1: (B) holds, mcap = 0.75 * 1.5e3 for a core row
    unit[:, -1] = np.sign(unit[:, -1]) * 2.93              # (an extension LLR sits AT the largest magnitude: Lambda_pe is what the test says)
    for scale, big in ((top / 2.93, True), (6.0, False)):
        xd = deint(scale * unit)
        tb_ref, ok_ref = ops.ldpc_recover_decode_merge(xd, cfg, nl, qm, n_iter, rows=rows)
        assert bool(ok_ref.all())
        for in_kernel in (True, False, "persistent"):
            ex = {}
            for flags in (0, 1, 2):
                tb_c, ok_c, e = ops.ldpc_recover_decode_merge_certified(xd, cfg, nl, qm, (8, 12), n_iter, rows=rows, flags=flags, in_kernel=bool(in_kernel),
                                                                        persistent=in_kernel == "persistent")
                assert torch.equal(ok_c, ok_ref) and torch.equal(tb_c, tb_ref), (scale, flags, in_kernel)
                ex[flags] = e.cpu().numpy()
            if big:
                assert (ex[0] == 0).all() and (ex[2] == 0).all(), (in_kernel, np.unique(ex[0]), np.unique(ex[2]))      # (S) / (Q) refuse
                assert (ex[1] == 8).all(), (in_kernel, np.unique(ex[1]))                                              # ... and nothing else does
            else:
                assert (ex[0] > 0).all() and np.array_equal(ex[0], ex[1])


def test_witness_block_whose_bit_flips_after_every_check_passed(dev):
    """tests/golden/cert_witness_S.npz (tests/test_certificate_cpu.py states what it is: after iteration 12 every parity check passes,
    after iteration 13 one information bit has flipped; the ORACLE's certificate without (S) would stop it at 12) through the library
    (BG1 / Zc 16 / 15 rows, generic certified decoder nrx_ldpc_decode_certified_f64): the float64 decoder reproduces the flip bit for bit;
    the complete certificate refuses the block (it runs all 13 iterations and carries the fixed schedule's bits); so does the library's
    certificate without (S) / (Q) -- its two slacks per row (every pm1 message held against the row's strongest demand) are coarser than
    the oracle's per-edge slacks and do not close here: more conservative, never less; with NO condition (flags 7) the block stops at 12 on
    bits that are not the fixed schedule's.  (The information bits are random -- no CRC to pass -- so the CRC filter is off, flags bit 2.)"""
    import os
    import torch
    from neoradium_amd import ops, _lib
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'cert_witness_S.npz'))
    k, ni, zc, rows = int(g['k']), int(g['num_iter']), int(g['zc']), int(g['rows'])
    cfg = _lib.ldpc_config(1, 22 * zc)
    assert (cfg.Zc, cfg.C, cfg.F, cfg.K) == (zc, 1, 0, 22 * zc)
    x = torch.from_numpy(g['llr']).to(dev)
    fixed = ops.ldpc_decode(x, cfg, ni, rows=rows).cpu().numpy()
    at_k = ops.ldpc_decode(x, cfg, k, rows=rows).cpu().numpy()
    weak = int(g['pos']) + 2 * zc
    assert np.array_equal(at_k, g['info'].astype(np.uint8)) and (fixed != at_k).sum() == 1 and fixed[0, weak] != at_k[0, weak]
    for flags in (4, 4 | 1):
        hard, ex = ops.ldpc_decode_certified(x, cfg, ni, [k], rows=rows, flags=flags, max_sweeps=16)
        assert int(ex[0]) == 0 and np.array_equal(hard.cpu().numpy(), fixed), flags
    hard, ex = ops.ldpc_decode_certified(x, cfg, ni, [k], rows=rows, flags=7)
    assert int(ex[0]) == k and np.array_equal(hard.cpu().numpy(), at_k)          # stopped at 12: not the fixed schedule's bits


def test_gpu_certificate_against_the_oracle_on_a_slot(dev):
    """One slot at the metric configuration (72 blocks of Zc 384): the blocks the kernel certifies after 8 iterations carry, at 8,
    the bits the ORACLE's full 50-iteration run ends on, and the oracle's own certificate (another search for the slacks, same
    theorem) certifies nearly the same set."""
    import torch
    from neoradium_amd import ops
    link = _metric_link()
    cw = link.cw[0]
    cfg = cw['cfg']
    _, det = link.run(4242, 1, 31.5, seed=4, details=True)
    llr = det[0][1]['llr']                                       # (1, G) in the reference's order
    rr = ops.ldpc_rate_recover(llr, cfg, cw['nl'], cw['qm']).cpu().numpy()
    rows = cw['rows']
    res = cert.decode_certified(rr, 1, cfg.iLS, cfg.Zc, 50, rows, [8], (), sweeps=12)
    certd = _metric_link(certifiedExit=(8,))
    _, d1 = certd.run(4242, 1, 31.5, seed=4, details="verdicts")
    ex = certd.last_exit_iter.cpu().numpy()
    gpu8 = ex == 8
    assert gpu8.sum() > 10
    assert np.array_equal(res['bits_at'][8][gpu8], res['bits'][gpu8]), "a GPU-certified block's bits at 8 are not the oracle's final bits"
    pay = cfg.cb_len - 24
    got = d1[0][1]['tb_out'].reshape(-1, pay).cpu().numpy()
    assert np.array_equal(got, res['bits'][:, :pay]), "decoded payload differs from the oracle's 50-iteration run"
    agree = (gpu8 == res['cert'][8]).mean()
    assert agree >= 0.9, (int(gpu8.sum()), int(res['cert'][8].sum()))


# (bg, A, G, nl, qm, sigma): both base graphs, lifting sizes 22 ... 384, with and without fillers, truncated and full row sets
GENERIC_CASES = [
    (1, 10000, 22808, 1, 2, 0.80),      # C=2, Zc=240, F=244
    (2, 2408, 7800, 1, 2, 1.05),        # C=1, Zc=256 (BASELINE cfg1's graph)
    (1, 30216, 63648, 2, 4, 0.78),      # C=4, Zc=352
    (2, 3817, 12000, 1, 6, 1.00),       # BG2, two blocks
    (1, 800, 2400, 2, 4, 0.95),         # Zc=40
    (2, 100, 600, 1, 2, 1.20),          # Zc=22 (less than a wavefront)
    (1, 25344 * 2, 3 * 13104 * 2, 4, 6, 0.62),   # Zc=384, C=7
]


@pytest.mark.parametrize("bg,A,G,nl,qm,sigma", GENERIC_CASES)
def test_generic_certified_decoder_any_code(dev, bg, A, G, nl, qm, sigma):
    """nrx_ldpc_decode_certified_f64 (any base graph / lifting size / row count): hard decisions identical to the fixed schedule of
    nrx_ldpc_decode_rows_f64 for every block, certified or not; the certified blocks' exit iterations are those of the oracle's
    certificate for most blocks, and every block the kernel certifies carries, at its exit, the oracle's final bits."""
    import torch
    from neoradium_amd import ops, _lib
    rng = np.random.default_rng(A + qm)
    n_tb = 4
    cfg = _lib.ldpc_config(bg, A + 24)
    tb = torch.from_numpy(rng.integers(0, 2, (n_tb, A)).astype(np.uint8)).to(dev)
    coded = ops.ldpc_encode(ops.ldpc_segment(tb, cfg), cfg)
    bits = ops.ldpc_rate_match(coded, cfg, G, nl, qm).cpu().numpy().astype(np.float64)
    sig = (sigma * np.array([0.8, 0.95, 1.0, 1.25]))[:, None]
    llr = (2 / sig ** 2) * ((1 - 2 * bits) + sig * rng.standard_normal(bits.shape))
    llr[rng.random(llr.shape) < 0.002] = 0.0
    rr = ops.ldpc_rate_recover(torch.from_numpy(llr).to(dev), cfg, nl, qm)
    lens = _lib.ldpc_cb_lens(G, cfg.C, nl, qm)
    total = 46 if bg == 1 else 42
    rows = min(total, ops.ldpc_active_rows(cfg, max(lens)))
    n_iter, checks = 24, [4, 7, 10, 14, 18]
    ref = ops.ldpc_decode(rr, cfg, n_iter, rows=rows)
    hard, ex = ops.ldpc_decode_certified(rr, cfg, n_iter, checks, rows=rows)
    assert torch.equal(hard, ref), f"{int((hard != ref).any(1).sum())} blocks differ from the fixed schedule"
    exn = ex.cpu().numpy()
    assert (exn > 0).any() and set(np.unique(exn)) <= set([0] + checks)
    # against the oracle's certificate on the same LLRs
    p = oc.LdpcParams(bg, A + 24)
    fcols = cert.filler_columns(bg, p.Zc, p.F)
    res = cert.decode_certified(rr.cpu().numpy(), bg, p.iLS, p.Zc, n_iter, rows, checks, fcols, sweeps=12)
    assert np.array_equal(hard.cpu().numpy(), res['bits'][:, :cfg.K])
    first = np.zeros(len(exn), dtype=int)
    for k in reversed(checks):
        first[res['cert'][k]] = k
    assert (first == exn).mean() >= 0.75, (first, exn)
    for b in np.nonzero(exn)[0]:
        assert np.array_equal(res['bits_at'][int(exn[b])][b], res['bits'][b])
    # and a broken certificate (no conditions) stops blocks that have not converged: caught by the comparison above
    hard_b, ex_b = ops.ldpc_decode_certified(rr, cfg, n_iter, [1], rows=rows, flags=3)
    assert (ex_b == 1).all() and not torch.equal(hard_b, ref)


def test_class_surface_decode_with_certified_exit(dev):
    """LdpcDecoder.decode(..., certifiedExit=...): the reference's call with one opt-in keyword -- same bits as without it."""
    import neoradium_amd as nr
    rng = np.random.default_rng(12)
    enc = nr.LdpcEncoder(baseGraphNo=2, modulation='QPSK', txLayers=1, targetRate=0.4)
    tb = rng.integers(0, 2, 3000).astype(np.int8)
    G = 9000
    tx = enc.getRateMatchedCodeBlocks(tb, G)
    llr = 4.0 * ((1 - 2.0 * np.asarray(tx)) + 0.8 * rng.standard_normal(len(tx)))
    dec = enc.getDecoder()
    rr = dec.recoverRate(llr, len(tb))
    ref = dec.decode(rr, numIter=20)
    got = dec.decode(rr, numIter=20, certifiedExit=(5, 9, 14))
    assert np.array_equal(ref, got) and dec.lastExitIter.shape == (rr.shape[0],) and (dec.lastExitIter > 0).any()
    with pytest.raises(ValueError):
        dec.decode(rr, numIter=20, certifiedExit=(5,), outputBelief=True)


def test_certified_exit_at_baseline_cfg2(dev):
    """BASELINE configs[1] (106 PRB @30 kHz, 64-QAM, 2 layers, 2x2 MMSE, CDL-C, BG1: 16 code blocks of Zc 384 per slot) through the engine:
    the certified schedule's bits and verdicts equal the fixed schedule's across its waterfall."""
    import torch
    import neoradium_amd as nr

    def link(**kw):
        nr.random.setSeed(123)
        car = nr.Carrier(numRbs=106, spacing=30)
        p = nr.PDSCH(car.curBwp, numLayers=2, nID=car.cellId, modulation='64QAM')
        p.setDMRS(configType=1, additionalPos=1)
        ch = nr.CdlChannel(car.curBwp, 'C', delaySpread=300, carrierFreq=4e9, dopplerShift=5,
                           txAntenna=nr.AntennaPanel([1, 1], polarization="x"), rxAntenna=nr.AntennaPanel([1, 1], polarization="x"))
        return nr.PdschLink(p, ch, 666 / 1024, baseGraphNo=1, numIter=50, freqDomain=False, chanEst="LS", decoder="f64", **kw)

    fixed, certd = link(), link(certifiedExit=(8, 16))
    pay = fixed.cfg.cb_len - 24
    stopped = total = 0
    for snr in (16.0, 18.0, 20.0, 23.0):
        _, d0 = fixed.run(40, 192, snr, seed=6, details="verdicts")
        _, d1 = certd.run(40, 192, snr, seed=6, details="verdicts")
        assert torch.equal(d0[0][1]['cb_ok'], d1[0][1]['cb_ok']) and torch.equal(d0[0][1]['tb_out'], d1[0][1]['tb_out']), snr
        stopped += int((certd.last_exit_iter > 0).sum())
        total += certd.last_exit_iter.numel()
    assert stopped > 0.3 * total, (stopped, total)
