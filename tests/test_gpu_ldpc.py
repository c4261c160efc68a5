"""GPU parity: LDPC/CRC chain through the C ABI (neoradium_amd.ops) vs the NumPy oracle on the same seeded inputs."""
import numpy as np
import pytest

from oracle import coding as oc

pytestmark = pytest.mark.gpu

# (bg, A, G, nl, qm, rv)  -- sizes the NumPy oracle finishes in seconds
CASES = [
    (1, 10000, 22808, 1, 2, 0),     # the reference's MATLAB case: C=2, Zc=240, iLS=7, F=244
    (2, 2408, 7800, 1, 2, 0),       # BASELINE cfg1: C=1, Zc=256
    (1, 30216, 63648, 2, 4, 0),     # PDSCH-BLER notebook: C=4, Zc=352
    (2, 3817, 12000, 1, 6, 2),      # BG2 two blocks, rv=2
    (1, 800, 2400, 2, 4, 3),        # small Zc=40, rv=3
    (2, 100, 600, 1, 2, 1),         # Zc=22 (< one wavefront)
    (1, 25344 * 2, 3 * 13104 * 2, 4, 6, 0),   # Zc=384 (the metric's lifting size), C=7
]


def _t(x, dev, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    return t if dtype is None else t.to(dtype)


@pytest.mark.parametrize("bg,A,G,nl,qm,rv", CASES)
def test_tx_chain_bit_exact(dev, bg, A, G, nl, qm, rv):
    from neoradium_amd import ops, _lib
    rng = np.random.default_rng(A + rv)
    n_tb = 3
    tb = rng.integers(0, 2, (n_tb, A)).astype(np.uint8)
    cfg = _lib.ldpc_config(bg, A + 24)
    ref = [oc.encode_chain(tb[i], bg, G, nl, qm, rv) for i in range(n_tb)]
    p = ref[0][1]['p']
    assert (cfg.C, cfg.Zc, cfg.iLS, cfg.K, cfg.N, cfg.F) == (p.C, p.Zc, p.iLS, p.K, p.N, p.F)
    cbs = ops.ldpc_segment(_t(tb, dev), cfg)
    assert np.array_equal(cbs.cpu().numpy().reshape(n_tb, p.C, p.K), np.stack([r[1]['cbs'] for r in ref]))
    coded = ops.ldpc_encode(cbs, cfg)
    assert np.array_equal(coded.cpu().numpy().reshape(n_tb, p.C, p.N), np.stack([r[1]['coded'] for r in ref]))
    full = ops.ldpc_encode(cbs, cfg, puncture=False)
    assert np.array_equal(full.cpu().numpy()[:, 2 * p.Zc:], coded.cpu().numpy())
    rm = ops.ldpc_rate_match(coded, cfg, G, nl, qm, rv)
    assert np.array_equal(rm.cpu().numpy(), np.stack([r[0] for r in ref]))
    # CRC primitive on the raw TBs
    crc = ops.crc(_t(tb, dev), '24A')
    assert np.array_equal(crc.cpu().numpy(), oc.crc_bits(tb, '24A'))


@pytest.mark.parametrize("bg,A,G,nl,qm,rv", CASES)
def test_rx_chain(dev, bg, A, G, nl, qm, rv):
    import torch
    from neoradium_amd import ops, _lib
    rng = np.random.default_rng(7 * A + rv)
    n_tb = 2
    tb = rng.integers(0, 2, (n_tb, A)).astype(np.uint8)
    cfg = _lib.ldpc_config(bg, A + 24)
    n_iter = 8
    llrs, refs = [], []
    for i in range(n_tb):
        rm, d = oc.encode_chain(tb[i], bg, G, nl, qm, rv)
        p = d['p']
        hi_rate = A / G > 0.5
        # slot 0 decodes cleanly, slot 1 sits near/below the waterfall
        sigma = (0.5 if hi_rate else 0.75) if i == 0 else (0.74 if hi_rate else 1.15)
        llr = (1 - 2.0 * rm) * 2 / sigma ** 2 + rng.normal(0, 2 / sigma, len(rm))
        llrs.append(llr)
        rr, _ = oc.rate_recover(llr, p, nl, qm, rv)
        bel = oc.decode(rr, bg, p.iLS, p.Zc, n_iter, only_info=False, belief=True)
        refs.append((rr, bel))
    llr64 = _t(np.stack(llrs), dev)
    # rate recovery: bit-identical float64 (incl. 1e20 fillers), float32 = rounded inputs
    rr64 = ops.ldpc_rate_recover(llr64, cfg, nl, qm, rv)
    assert np.array_equal(rr64.cpu().numpy().reshape(n_tb, p.C, p.N), np.stack([r[0] for r in refs]))
    rr32 = ops.ldpc_rate_recover(llr64.float(), cfg, nl, qm, rv)
    # float64 decoder: beliefs bit-exact with the reference arithmetic, all columns
    bel64 = ops.ldpc_decode(rr64, cfg, n_iter, only_info=False, belief=True).cpu().numpy()
    ref_bel = np.concatenate([r[1] for r in refs])
    assert np.array_equal(bel64, ref_bel), f"max diff {np.abs(bel64 - ref_bel).max()}"
    hard64 = ops.ldpc_decode(rr64, cfg, n_iter).cpu().numpy()
    assert np.array_equal(hard64, (ref_bel[:, :p.K] < 0).astype(np.uint8))
    # CRC + merge
    tb_out, cb_ok, tb_ok = ops.ldpc_crc_merge(_t(hard64, dev), cfg)
    for i in range(n_tb):
        o, c = oc.crc_check_and_merge(hard64[i * p.C:(i + 1) * p.C], p)
        assert np.array_equal(cb_ok[i].cpu().numpy().astype(bool), c)
        assert np.array_equal(tb_out[i].cpu().numpy()[:len(o)][:p.B], o[:p.B])
        assert bool(tb_ok[i]) == bool(oc.crc_check(o[:p.B], '24A'))
    assert cb_ok[0].all() and np.array_equal(tb_out[0, :A].cpu().numpy(), tb[0])
    # float32 decoder: same CRC verdicts; identical hard bits and <=1e-5 (relative to the block's LLR scale)
    # beliefs on blocks that converged
    bel32 = ops.ldpc_decode(rr32, cfg, n_iter, only_info=False, belief=True).cpu().numpy().astype(np.float64)
    hard32 = ops.ldpc_decode(rr32, cfg, n_iter).cpu().numpy()
    _, cb_ok32, _ = ops.ldpc_crc_merge(_t(hard32, dev), cfg)
    assert np.array_equal(cb_ok32.cpu().numpy(), cb_ok.cpu().numpy())
    ok = cb_ok.cpu().numpy().reshape(-1).astype(bool)
    assert ok.any()
    assert np.array_equal(hard32[ok], hard64[ok])
    nf = np.ones(ref_bel.shape[1], bool)
    nf[p.K - p.F:p.K] = False                   # filler columns carry +-1e10
    for b in np.nonzero(ok)[0]:
        scale = np.abs(ref_bel[b, nf]).max()
        assert np.abs(bel32[b, nf] - ref_bel[b, nf]).max() <= 1e-5 * scale
    # counters
    counters = torch.zeros(4, dtype=torch.int64, device=dev)
    ops.count_errors(cb_ok, tb_out, _t(tb, dev), counters)
    c = counters.cpu().numpy()
    assert c[0] == (~ok).sum() and c[1] == ok.size and c[3] == n_tb * A
    assert c[2] == (tb_out[:, :A].cpu().numpy() != tb).sum()


def test_harq_soft_combining(dev):
    """recoverRate with a HARQ buffer (ldpc.py:1377-1412): 4 redundancy versions accumulate in place."""
    from neoradium_amd import ops, _lib
    bg, A, G, nl, qm = 1, 10000, 20900, 1, 4
    rng = np.random.default_rng(5)
    tb = rng.integers(0, 2, A).astype(np.uint8)
    cfg = _lib.ldpc_config(bg, A + 24)
    circ_ref, circ_dev = None, None
    for rv in (0, 2, 3, 1):
        rm, d = oc.encode_chain(tb, bg, G, nl, qm, rv)
        p = d['p']
        llr = (1 - 2.0 * rm) + rng.normal(0, 1.2, len(rm))
        rr_ref, circ_ref = oc.rate_recover(llr, p, nl, qm, rv, circ=circ_ref)
        if circ_dev is None:
            import torch
            circ_dev = torch.zeros((p.C, p.N - p.F), dtype=torch.float64, device=dev)
        rr = ops.ldpc_rate_recover(_t(llr[None], dev), cfg, nl, qm, rv, circ=circ_dev)
        assert np.array_equal(rr.cpu().numpy(), rr_ref)
        assert np.array_equal(circ_dev.cpu().numpy(), circ_ref)


def test_limited_buffer_rate_matching_vs_reference(dev):
    """nRef > 0 (LBRM; ldpc.py:1093-1159, 1347-1418) against the reference's own outputs (tests/golden/coding_lbrm.npz): rate-matched
    bits, rate-recovered LLRs (the library's matrix is N wide: the reference's (C, Ncb) columns, zeros beyond), HARQ buffers after
    three redundancy versions, both through the C ABI and through the class surface; then a decode through the limited buffer."""
    import os
    import torch
    from neoradium_amd import ops, _lib, LdpcEncoder, LdpcDecoder
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'coding_lbrm.npz'))
    unpack = lambda a, shape: np.unpackbits(a)[:int(np.prod(shape))].reshape(shape).astype(np.uint8)   # noqa: E731
    mods = {2: 'QPSK', 4: '16QAM', 6: '64QAM', 8: '256QAM'}
    for i, (bg, A, G, nl, qm, rv, nref) in enumerate(g['cases'].tolist()):
        p_ = f'c{i}_'
        tb = g[p_ + 'tb'].astype(np.uint8)
        cfg = _lib.ldpc_config(bg, A + 24)
        C_, Zc, K, F, ncb = g[p_ + 'params'].tolist()
        assert (cfg.C, cfg.Zc, cfg.K, cfg.F) == (C_, Zc, K, F) and ncb == min(cfg.N, nref)
        coded = ops.ldpc_encode(ops.ldpc_segment(_t(tb[None], dev), cfg), cfg)
        assert np.array_equal(coded.cpu().numpy().reshape(cfg.C, cfg.N), unpack(g[p_ + 'coded'], (cfg.C, cfg.N)))
        rm = ops.ldpc_rate_match(coded, cfg, G, nl, qm, rv, nref)
        rm_ref = unpack(g[p_ + 'rm'], (G,))
        assert np.array_equal(rm.cpu().numpy()[0], rm_ref)
        llr = np.float64(g[p_ + 'llr'])
        circ = torch.zeros((cfg.C, ncb - cfg.F), dtype=torch.float64, device=dev)
        rr = ops.ldpc_rate_recover(_t(llr[None], dev), cfg, nl, qm, rv, nref, circ=circ).cpu().numpy()
        ref = g[p_ + 'rr']
        assert rr.shape == (cfg.C, cfg.N) and not rr[:, ncb:].any()
        assert np.array_equal(np.where(rr[:, :ncb] > 1e19, np.inf, rr[:, :ncb]), ref)
        buf_ref = np.concatenate([ref[:, :cfg.K - 2 * cfg.Zc - cfg.F], ref[:, cfg.K - 2 * cfg.Zc:]], axis=1)
        assert np.array_equal(circ.cpu().numpy(), buf_ref)
        # class surface: same calls as the reference's (LdpcEncoder(nRef=...).rateMatch, LdpcDecoder(nRef=...).recoverRate)
        enc = LdpcEncoder(baseGraphNo=bg, modulation=mods[qm], txLayers=nl, nRef=nref)
        cw = enc.encode(enc.doSegmentation(enc.appendCrc(tb.astype(np.int8), '24A')))
        assert np.array_equal(enc.rateMatch(cw, G, rv=rv), rm_ref.astype(np.int8))
        dec = LdpcDecoder(bg, mods[qm], nl, nRef=nref)
        rr_c = dec.recoverRate(llr, A)
        if rv == 0:
            assert rr_c.shape == ref.shape and np.array_equal(np.where(rr_c > 1e19, np.inf, rr_c), ref)
        else:
            assert rr_c.shape == ref.shape        # (without a HARQ object the reference takes rv 0)
    bg, A, G, nl, qm, nref = g['harq_params'].tolist()
    cfg = _lib.ldpc_config(bg, A + 24)
    ncb = min(cfg.N, nref)
    circ = torch.zeros((cfg.C, ncb - cfg.F), dtype=torch.float64, device=dev)
    for t, rv in enumerate((0, 2, 3)):
        rr = ops.ldpc_rate_recover(_t(np.float64(g[f'harq_llr{t}'])[None], dev), cfg, nl, qm, rv, nref, circ=circ)
    assert np.array_equal(circ.cpu().numpy(), g['harq_buf'])
    # ... and the soft-combined limited buffer decodes (library: N-wide matrix; class surface: the (C, Ncb) matrix padded by decode)
    hard = ops.ldpc_decode(rr, cfg, 20)
    tb_out, ok, _ = ops.ldpc_crc_merge(hard, cfg)
    assert bool(ok.all()) and np.array_equal(tb_out.cpu().numpy()[0, :A], g['harq_tb'].astype(np.uint8))
    dec = LdpcDecoder(bg, mods[qm], nl, nRef=nref)

    class H:
        pass
    h = H()
    h.decBuffer, h.rv = None, 0
    for t, rv in enumerate((0, 2, 3)):
        h.rv = rv
        rr_c = dec.recoverRate(np.float64(g[f'harq_llr{t}']), A, harq=h)
    assert rr_c.shape == (cfg.C, ncb)
    bits = dec.decode(rr_c, numIter=20)
    merged, crc = dec.checkCrcAndMerge(bits)
    assert np.all(crc) and np.array_equal(merged[:A], g['harq_tb'])


def test_errors_are_valueerrors(dev):
    import torch
    from neoradium_amd import ops, _lib
    cfg = _lib.ldpc_config(1, 10024)
    with pytest.raises(ValueError):
        _lib.ldpc_config(3, 100)
    coded = torch.zeros((cfg.C, cfg.N), dtype=torch.uint8, device=dev)
    with pytest.raises(ValueError):
        ops.ldpc_rate_match(coded, cfg, 22808, 1, 2, rv=4)      # ldpc.py:1131
    with pytest.raises(ValueError):
        ops.ldpc_decode(torch.zeros((2, 17), device=dev), cfg)


@pytest.mark.parametrize("bg,A,G,nl,qm", [(1, 10000, 20900, 1, 4), (2, 3817, 12000, 1, 6), (1, 800, 1400, 2, 2)])
def test_harq_batch_rv_and_reset(dev, bg, A, G, nl, qm):
    """Per-transport-block redundancy versions and soft-buffer restarts (batched HARQ processes) give, row by row,
    what the scalar-rv calls give (bit for bit), including wrap-around repetition (last case: E > circular buffer)."""
    import torch
    from neoradium_amd import ops, _lib
    rng = np.random.default_rng(A)
    cfg = _lib.ldpc_config(bg, A + 24)
    n = 5
    tb = _t(rng.integers(0, 2, (n, A)).astype(np.uint8), dev)
    coded = ops.ldpc_encode(ops.ldpc_segment(tb, cfg), cfg)
    rv = np.int32([0, 2, 3, 1, 2])
    reset = np.uint8([1, 0, 0, 1, 0])
    bits = ops.ldpc_rate_match(coded, cfg, G, nl, qm, rv=_t(rv, dev))
    for i in range(n):
        one = ops.ldpc_rate_match(coded[i * cfg.C:(i + 1) * cfg.C], cfg, G, nl, qm, rv=int(rv[i]))
        assert torch.equal(bits[i], one[0]), i
    for dt in (torch.float32, torch.float64):
        llr = _t(rng.standard_normal((n, bits.shape[1])), dev, dt)
        circ0 = _t(rng.standard_normal((n * cfg.C, cfg.N - cfg.F)), dev, dt)
        circ = circ0.clone()
        rr = ops.ldpc_rate_recover(llr, cfg, nl, qm, rv=_t(rv, dev), circ=circ, reset=_t(reset, dev))
        for i in range(n):
            c1 = circ0[i * cfg.C:(i + 1) * cfg.C].clone()
            if reset[i]:
                c1.zero_()
            r1 = ops.ldpc_rate_recover(llr[i:i + 1], cfg, nl, qm, rv=int(rv[i]), circ=c1)
            assert torch.equal(rr[i * cfg.C:(i + 1) * cfg.C], r1), (i, dt)
            assert torch.equal(circ[i * cfg.C:(i + 1) * cfg.C], c1), (i, dt)
    with pytest.raises(ValueError):
        ops.ldpc_rate_recover(llr, cfg, nl, qm, rv=_t(rv, dev))            # needs the soft buffer


def test_decoder_all_lifting_size_classes(dev):
    """Fast f32 kernel (generic any-Zc path with 1 or 2 code blocks per workgroup, and the specialised sizes) against
    the float64 kernel and the transmitted bits, over lifting sizes of every wave count and both base graphs."""
    import torch
    from neoradium_amd import ops, _lib
    rng = np.random.default_rng(99)
    # (bg, B): chosen to hit Zc = 384 352 320 288 256 208 144 96 64 36 (BG1) and 256 192 120 52 (BG2)
    cases = [(1, 8448 * 3), (1, 7744 * 2), (1, 7040), (1, 6336), (1, 5632), (1, 4576), (1, 3168), (1, 2112), (1, 1408),
             (1, 792), (2, 2560), (2, 1920), (2, 1200), (2, 520)]
    seen = set()
    for bg, B in cases:
        cfg = _lib.ldpc_config(bg, B)
        seen.add((bg, cfg.Zc))
        n_cb = 5 if cfg.C == 1 else cfg.C * 2 + (1 if cfg.C % 2 == 0 else 0) * 0
        n_tb = max(1, n_cb // cfg.C)
        tb = _t(rng.integers(0, 2, (n_tb, B - 24)).astype(np.uint8), dev)
        cbs = ops.ldpc_segment(tb, cfg)
        coded = ops.ldpc_encode(cbs, cfg)
        sig = 0.55 if bg == 1 else 0.7
        llr = (2.0 / sig ** 2) * (1.0 - 2.0 * coded.double() + sig * _t(rng.standard_normal(tuple(coded.shape)), dev))
        llr[:, cfg.K - cfg.F - 2 * cfg.Zc:cfg.K - 2 * cfg.Zc] = 1e20          # filler positions
        d32 = ops.ldpc_decode(llr.float(), cfg, 8)
        d64 = ops.ldpc_decode(llr, cfg, 8)
        assert torch.equal(d32, d64), (bg, cfg.Zc)
        assert torch.equal(d32, cbs), (bg, cfg.Zc)
    assert len(seen) >= 12, seen


@pytest.mark.parametrize("bg,A", [(2, 24), (2, 3816), (2, 3824), (1, 8424), (1, 8425), (1, 1277992)])
def test_chain_round_trip_extreme_sizes(dev, bg, A):
    """Size-independent property at the edges of the size range (smallest block, the one- / two-code-block boundary of
    each base graph, the largest NR transport block: 152 code blocks): segment -> encode -> rate match -> noiseless
    LLRs -> rate recover -> decode -> CRC + merge returns the transport block, every CRC passes."""
    import torch
    from neoradium_amd import ops, _lib
    rng = np.random.default_rng(A)
    cfg = _lib.ldpc_config(bg, A + 24)
    nl, qm = 1, 2
    G = int(np.ceil((A + 24) / 0.5 / (nl * qm))) * nl * qm
    G = max(G, cfg.C * nl * qm)
    tb = _t(rng.integers(0, 2, (2, A)).astype(np.uint8), dev)
    coded = ops.ldpc_encode(ops.ldpc_segment(tb, cfg), cfg)
    bits = ops.ldpc_rate_match(coded, cfg, G, nl, qm)
    llr = (8.0 * (1.0 - 2.0 * bits.float())).contiguous()
    rr = ops.ldpc_rate_recover(llr, cfg, nl, qm)
    dec = ops.ldpc_decode(rr, cfg, 6)
    out, cb_ok, tb_ok = ops.ldpc_crc_merge(dec, cfg)
    assert bool(cb_ok.all()) and bool(tb_ok.all()), (cfg.C, cfg.Zc)
    assert torch.equal(out[:, :A], tb)


def test_empty_batches(dev):
    """Zero transport blocks / code words are a no-op everywhere (no launch, correctly shaped empty outputs)."""
    import torch
    from neoradium_amd import ops, _lib
    from neoradium_amd.polar import PolarEncoder, PolarDecoder
    cfg = _lib.ldpc_config(1, 10024)
    tb = torch.empty((0, 10000), dtype=torch.uint8, device=dev)
    cbs = ops.ldpc_segment(tb, cfg)
    assert tuple(cbs.shape) == (0, cfg.K)
    coded = ops.ldpc_encode(cbs, cfg)
    assert tuple(coded.shape) == (0, cfg.N)
    bits = ops.ldpc_rate_match(coded, cfg, 20000, 1, 2)
    assert tuple(bits.shape) == (0, 20000)
    rr = ops.ldpc_rate_recover(torch.empty((0, 20000), dtype=torch.float32, device=dev), cfg, 1, 2)
    dec = ops.ldpc_decode(rr, cfg, 5)
    assert tuple(dec.shape) == (0, cfg.K)
    assert tuple(ops.crc(torch.empty((0, 100), dtype=torch.uint8, device=dev), '24A').shape) == (0, 24)
    enc, pdec = PolarEncoder(30, 120, 'dci'), PolarDecoder(30, 120, 'dci')
    x = enc.encodeDevice(torch.empty((0, 54), dtype=torch.uint8, device=dev))
    assert tuple(x.shape) == (0, 128) and tuple(enc.rateMatchDevice(x).shape) == (0, 120)
    msg, ok = pdec.decodeDevice(torch.empty((0, 128), dtype=torch.float64, device=dev))
    assert tuple(msg.shape) == (0, 54) and ok.numel() == 0


@pytest.mark.parametrize("bg,zc,n_tx_cols", [(1, 384, 35), (1, 384, 31), (1, 384, 40), (1, 384, 50), (1, 352, 35), (1, 128, 33),
                                             (2, 256, 20), (2, 64, 16)])
def test_punctured_rows_are_exact_no_ops(dev, bg, zc, n_tx_cols):
    """Decoding with only the rows whose extension parity was received gives the same hard bits as running all rows
    (nrx_ldpc_decode_rows_*): a row whose degree-1 extension column holds all-zero LLRs has min1 = 0 there and sends +-0
    to every other column.  Checked for the float32 throughput kernel (compile-time row counts 13/16/22/31 for Zc = 384,
    all rows otherwise), the generic float32 kernel and the float64 kernel (run-time row count), on noisy LLRs where
    part of the blocks do not converge, with a partially filled last column and exact zeros sprinkled into core columns;
    and against the oracle's float64 beliefs."""
    import torch
    from neoradium_amd import ops, _lib
    kb, core, rows_all, ncols = (22, 26, 46, 68) if bg == 1 else (10, 14, 42, 52)
    ils = next(i for i, b in enumerate((2, 3, 5, 7, 9, 11, 13, 15)) if zc % b == 0 and (zc // b) & (zc // b - 1) == 0)
    cfg = _lib.LdpcCfg()
    cfg.bg, cfg.Zc, cfg.iLS, cfg.K, cfg.N, cfg.F, cfg.C, cfg.B, cfg.cb_len = bg, zc, ils, kb * zc, (ncols - 2) * zc, 0, 1, 0, 0
    rng = np.random.default_rng(zc + n_tx_cols)
    n_cb = 10
    sig = 0.95 if bg == 1 else 1.1
    llr = 2 / sig ** 2 + (2 / sig) * rng.standard_normal((n_cb, cfg.N))
    e_max = n_tx_cols * zc - zc // 3                      # the last received column is only partly filled
    llr[:, e_max:] = 0.0
    llr[rng.random(llr.shape) < 0.002] = 0.0              # a few exact zeros among the received LLRs
    rows = ops.ldpc_active_rows(cfg, e_max)
    assert rows == min(rows_all, max(4, (e_max - 1) // zc + 2 - core + 1 + 4)) and rows < rows_all
    for ft in (torch.float32, torch.float64):
        x = torch.from_numpy(llr).to(dev).to(ft)
        full = ops.ldpc_decode(x, cfg, 14)
        part = ops.ldpc_decode(x, cfg, 14, rows=rows)
        more = ops.ldpc_decode(x, cfg, 14, rows=min(rows + 3, rows_all))
        assert torch.equal(full, part) and torch.equal(full, more)
    if zc <= 128 or (bg == 1 and zc == 384 and rows <= 15):   # the oracle (float64 beliefs of the core columns) agrees;
                                                           # Zc 384 with <= 15 rows = the on-chip float64 kernel (nrx_ldpc_dec3.hip)
        from oracle import coding as oc
        a = oc.decode(llr[:3], bg, ils, zc, num_iter=14, only_info=False, belief=True)
        b = oc.decode(llr[:3], bg, ils, zc, num_iter=14, only_info=False, belief=True, rows=rows)
        assert np.array_equal(a[:, :core * zc], b[:, :core * zc])
        assert np.array_equal((a[:, :kb * zc] < 0).astype(np.uint8), ops.ldpc_decode(torch.from_numpy(llr[:3]).to(dev), cfg, 14, rows=rows).cpu().numpy())
    with pytest.raises(ValueError):
        ops.ldpc_decode(torch.from_numpy(llr).to(dev), cfg, 5, rows=3)
    with pytest.raises(ValueError):
        ops.ldpc_decode(torch.from_numpy(llr).to(dev), cfg, 5, rows=rows, belief=True)


@pytest.mark.gpu
@pytest.mark.parametrize("fill", [0, 1, 48, 64, 72, 200, 383])
def test_waves_leave_out_an_all_zero_last_layer(dev, fill):
    """NRX_DEC3_SKIPZ (nrx_ldpc_dec3.hip): a wave whose 64 rows of the last layer all have a zero extension LLR runs the copy of the
    iteration loop without that layer.  `fill` received LLRs in the last layer's extension column (metric configuration: 48 of 384):
    none of the six waves, one, two ... all of them run the layer; hard bits against the oracle's float64 decoder (ldpc.py:1495-1581)
    and against the run with every row."""
    import torch
    from neoradium_amd import ops, _lib
    from oracle import coding as oc
    zc, rows = 384, 15
    cfg = _lib.LdpcCfg()
    cfg.bg, cfg.Zc, cfg.iLS, cfg.K, cfg.N, cfg.F, cfg.C, cfg.B, cfg.cb_len = 1, zc, 1, 22 * zc, 66 * zc, 0, 1, 0, 0
    rng = np.random.default_rng(900 + fill)
    llr = 2 / 0.95 ** 2 + (2 / 0.95) * rng.standard_normal((6, cfg.N))
    llr[:, 34 * zc + fill:] = 0.0                          # column 36 = the extension column of layer 14
    llr[0, 34 * zc:] = 0.0                                 # (one block whose last layer is empty whatever `fill` says)
    x = torch.from_numpy(llr).to(dev)
    got = ops.ldpc_decode(x, cfg, 12, rows=rows)
    assert torch.equal(got, ops.ldpc_decode(x, cfg, 12))
    want = oc.decode(llr, 1, 1, zc, num_iter=12, rows=rows)
    assert np.array_equal(np.asarray(want).astype(np.uint8), got.cpu().numpy())


@pytest.mark.parametrize("tbs,qm,nl,g_extra,e_target", [(25000, 6, 4, 0, 13000), (25000, 6, 4, 7, 13000), (33000, 4, 2, 3, 13000), (16700, 8, 1, 0, 13000),
                                                        (25000, 6, 4, 0, 12100), (25000, 6, 4, 7, 11300), (33000, 4, 2, 3, 10500), (16700, 8, 1, 0, 9800)])
def test_fused_recover_decode_merge_equals_separate_stages(dev, tbs, qm, nl, g_extra, e_target):
    """nrx_ldpc_recover_decode_merge_f64 (initial fill = rate recovery gathering straight from the demapper LLRs, tail =
    CRC24B check + merge) against the three separate entries on the same LLRs: transport block bits and CRC verdicts
    identical -- with unequal E_r across the code blocks (g_extra), filler bits (F > 0), partly filled last columns,
    blocks that converge and blocks that do not, and a slot batch that leaves a lone code block in the last workgroup.  `e_target`
    bits per code block: 15 rows; 12 (the 13-row instantiation with its last layer all zero: the waves leave it out, NRX_DEC3_SKIPZ);
    10 / 8 / 6 rows (the copy of the 13-row kernel that leaves the layers beyond the row count out, MODE bit 3)."""
    import torch
    from neoradium_amd import ops, _lib
    cfg = _lib.ldpc_config(1, tbs + 24)
    assert cfg.Zc == 384 and cfg.C > 1 and cfg.F > 0
    n_tb = 3
    rng = np.random.default_rng(tbs + qm)
    # about 15 rows' worth of bits per code block, E_r = multiples of nl*qm, the last g_extra blocks one step longer
    e_small = (e_target // (nl * qm)) * (nl * qm)
    G = cfg.C * e_small + g_extra * nl * qm
    lens = _lib.ldpc_cb_lens(G, cfg.C, nl, qm)
    assert sum(lens) == G and (g_extra == 0 or len(set(lens)) == 2)
    tb = torch.from_numpy(rng.integers(0, 2, (n_tb, tbs)).astype(np.uint8)).to(dev)
    coded = ops.ldpc_encode(ops.ldpc_segment(tb, cfg), cfg)
    bits = ops.ldpc_rate_match(coded, cfg, G, nl, qm).cpu().numpy().astype(np.float64)
    # one clean, one marginal, one hopeless transport block (the marginal one moves with the code rate)
    sig = np.array([0.5, {13000: 0.78, 12100: 0.72, 11300: 0.66, 10500: 0.6, 9800: 0.55}[e_target], 1.1])[:, None]
    sig[0, 0] = min(0.5, 0.8 * sig[1, 0])
    llr = (2 / sig ** 2) * ((1 - 2 * bits) + sig * rng.standard_normal(bits.shape))
    llr[rng.random(llr.shape) < 0.001] = 0.0
    x = torch.from_numpy(llr).to(dev)
    rows = ops.ldpc_active_rows(cfg, max(lens))
    assert rows <= 15 and (e_target != 13000 or rows == 15) and (e_target != 12100 or rows == 12) and (e_target > 11500 or rows <= 11)
    rr = ops.ldpc_rate_recover(x, cfg, nl, qm)
    dec = ops.ldpc_decode(rr, cfg, 12, rows=rows)
    tb_ref, ok_ref, _ = ops.ldpc_crc_merge(dec, cfg, want_tb_crc=False)
    # the fused entry reads every code block's LLRs de-interleaved (what nrx_qam_demap_cb_* writes): buffer position
    # q*(E_r/Qm) + s instead of s*Qm + q (ldpc.py:1390-1397)
    xd = np.empty_like(llr)
    off = 0
    for E in lens:
        xd[:, off:off + E] = llr[:, off:off + E].reshape(n_tb, E // qm, qm).transpose(0, 2, 1).reshape(n_tb, E)
        off += E
    xd = torch.from_numpy(xd).to(dev)
    assert ops.ldpc_fused_supported(cfg, nl, qm, G, rows)
    got = ops.ldpc_recover_decode_merge(xd, cfg, nl, qm, 12, rows=rows)
    assert got is not None, "no fused instantiation for BG1 / Zc 384 / <= 15 rows"
    tb_out, ok = got
    assert torch.equal(ok, ok_ref) and torch.equal(tb_out, tb_ref)
    n_ok = int(ok.sum())
    assert 0 < n_ok < ok.numel(), f"want both outcomes, got {n_ok}/{ok.numel()} passing"
    # rows = 0: the library works the row count out itself
    tb2, ok2 = ops.ldpc_recover_decode_merge(xd, cfg, nl, qm, 12, rows=0)
    assert torch.equal(ok2, ok_ref) and torch.equal(tb2, tb_ref)
    # the two-pass schedule on the same entry (failing blocks listed and counted on the device, decoded again from scratch):
    # a block that passes after the first pass keeps those bits, every other block gets the 12-iteration result
    fi = next((k for k in (1, 2, 3, 4) if int(ops.ldpc_recover_decode_merge(xd, cfg, nl, qm, k + 1, rows=rows)[1].sum()) == n_ok), None)
    if fi is None:
        assert e_target != 13000          # (a high-rate case whose blocks pass later than that: the schedule checks below want an early pass)
        return
    tb3, ok3 = ops.ldpc_recover_decode_merge(xd, cfg, nl, qm, fi, rows=rows)      # one iteration short of what the clean block needs
    n3 = int(ok3.sum())
    assert n3 < n_ok, "want blocks that pass early (if any), blocks that pass late and blocks that never do"
    pl = cfg.cb_len - 24
    early = ok3.bool()
    want_tb = torch.where(early[:, :, None], tb3.reshape(n_tb, cfg.C, pl), tb_out.reshape(n_tb, cfg.C, pl)).reshape(n_tb, -1)
    for restart in (True, False):       # from scratch / continued from the parked state: the same bits as one 12-iteration run
        tbt, okt = ops.ldpc_recover_decode_merge_two_pass(xd, cfg, nl, qm, fi, 12, rows=rows, restart=restart)
        assert torch.equal(okt, torch.where(early, ok3, ok)) and torch.equal(tbt, want_tb), restart
    # ... and with checks in between (blocks that still fail park their state again): a block keeps the bits of the first check it passes
    tb7, ok7 = ops.ldpc_recover_decode_merge(xd, cfg, nl, qm, 7, rows=rows)
    mid = ok7.bool() & ~early
    want3 = torch.where(mid[:, :, None], tb7.reshape(n_tb, cfg.C, pl), want_tb.reshape(n_tb, cfg.C, pl)).reshape(n_tb, -1)
    tbs, oks = ops.ldpc_recover_decode_merge_two_pass(xd, cfg, nl, qm, fi, 12, rows=rows, stages=(7,))
    assert torch.equal(oks, ok3 | ok7 | ok) and torch.equal(tbs, want3)
    sel = torch.empty(ok.numel(), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().nrx_select_failed(_lib.ptr(ok3), ok3.numel(), _lib.ptr(sel), _lib.ptr(cnt), _lib.stream()))
    assert int(cnt) == ok3.numel() - n3 and torch.equal(sel[:int(cnt)].long(), (ok3.reshape(-1) == 0).nonzero().reshape(-1))
    # unsupported configurations are reported, not approximated: Zc 352
    cfg2 = _lib.ldpc_config(1, 7500 * 3)
    assert cfg2.Zc != 384 and ops.ldpc_recover_decode_merge(xd[:, :cfg2.C * 1200], cfg2, nl, qm, 5) is None
    assert not ops.ldpc_fused_supported(cfg2, nl, qm, cfg2.C * 1200, 15)
    # the demapper's code-block option produces exactly that layout: same values as the symbol-major demap, permuted
    n_sym = G // qm
    sym = torch.from_numpy(rng.standard_normal((2, n_sym)) + 1j * rng.standard_normal((2, n_sym))).to(dev)
    std = ops.qam_demap(sym, 0.3, qm).cpu().numpy()
    dei = ops.qam_demap(sym, 0.3, qm, code_blocks=(cfg.C, nl)).cpu().numpy()
    off = 0
    for E in lens:
        assert np.array_equal(dei[:, off:off + E], std[:, off:off + E].reshape(2, E // qm, qm).transpose(0, 2, 1).reshape(2, E))
        off += E
    # ... and the rate-recovering option (nrx_qam_demap_rr_*) the whole of nrx_ldpc_rate_recover_* for a first transmission: the
    # columns asked for equal rate recovery of the symbol-major LLRs bit for bit (transmitted positions, zeros, LARGE_LLR fillers),
    # for every in/out type, a configuration with filler bits and unequal block lengths included
    for c_, G_, nl_, qm_ in ((cfg, G, nl, qm), (_lib.ldpc_config(1, 20000), 3 * 6600 + 12, 2, 6), (_lib.ldpc_config(2, 3000), 2 * 3000, 1, 4)):
        ns_ = G_ // qm_
        sy = torch.from_numpy(rng.standard_normal((3, ns_)) + 1j * rng.standard_normal((3, ns_))).to(dev)
        for dt_in, dt_out in ((torch.complex128, torch.float64), (torch.complex128, torch.float32), (torch.complex64, torch.float32)):
            want = ops.ldpc_rate_recover(ops.qam_demap(sy.to(dt_in), 0.3, qm_, llr_dtype=dt_out), c_, nl_, qm_)
            for n_cols in (c_.N // c_.Zc, c_.K // c_.Zc - 2 + 15):
                got = ops.qam_demap(sy.to(dt_in), 0.3, qm_, llr_dtype=dt_out, rate_recovered=(c_, nl_, n_cols))
                assert got.shape == want.shape and got.dtype == want.dtype
                assert torch.equal(got[:, :n_cols * c_.Zc], want[:, :n_cols * c_.Zc]), (c_.Zc, c_.F, dt_in, dt_out, n_cols)
    # ops.ldpc_rows_read names the columns the decoder entries read for a row count: decoding the partially initialised buffer
    # (everything behind those columns poisoned) gives the bits of the full one
    sy = torch.from_numpy(rng.standard_normal((2, n_sym)) + 1j * rng.standard_normal((2, n_sym))).to(dev) * 0.3 + \
        torch.from_numpy(rng.choice([-1.0, 1.0], (2, n_sym)) + 1j * rng.choice([-1.0, 1.0], (2, n_sym))).to(dev) * 0.46
    for dt_out in (torch.float32, torch.float64):
        full = ops.ldpc_rate_recover(ops.qam_demap(sy, 0.05, qm, llr_dtype=dt_out), cfg, nl, qm)
        for r_ in (4, 13, 14, 15, 16, 17, 22, 23, 31, 32, 46):
            n_cols = min(cfg.N // cfg.Zc, cfg.K // cfg.Zc - 2 + ops.ldpc_rows_read(cfg, r_, dt_out == torch.float32))
            part = torch.full((2 * cfg.C, cfg.N), float('nan'), dtype=dt_out, device=dev)
            assert ops.qam_demap(sy, 0.05, qm, llr_dtype=dt_out, rate_recovered=(cfg, nl, n_cols, part)) is part
            assert not bool(torch.isnan(part[:, :n_cols * cfg.Zc]).any())
            e_max = max(lens) + cfg.F          # behind the columns asked for only transmitted positions are written
            assert n_cols * cfg.Zc >= min(e_max, cfg.N) or bool(torch.isnan(part[:, max(e_max, n_cols * cfg.Zc):]).all())
            assert torch.equal(ops.ldpc_decode(part, cfg, 6, rows=r_), ops.ldpc_decode(full, cfg, 6, rows=r_)), (dt_out, r_)
    # repetition (E_r beyond the circular buffer) is declined, not approximated
    c3 = _lib.ldpc_config(2, 300)
    sy = torch.zeros((1, 2 * c3.N // 2), dtype=torch.complex128, device=dev)
    assert ops.qam_demap(sy, 0.3, 2, rate_recovered=(c3, 1, c3.N // c3.Zc)) is None


@pytest.mark.parametrize("bg,zc,n_tx_cols,all_rows", [(1, 384, 35, False), (1, 384, 35, True), (1, 352, 40, True), (2, 64, 20, True)])
def test_tied_minimum_under_the_1e5_quirk(dev, bg, zc, n_tx_cols, all_rows):
    """ldpc.py:1563-1570 with SEVERAL entries at the row minimum, all above 5e4, the first of them negative: the reference adds
    1e5 to np.argmin's entry, min2 drops BELOW min1, and only that first entry receives min2 -- the other tied entries keep
    min1.  Saturated LLRs of one magnitude (6e4) produce exactly that within the first iteration (a punctured column that has
    outgrown its neighbours leaves D-1 tied entries as the minimum).  float64 decoders (on-chip kernel for Zc 384 / 15 rows,
    workspace kernel otherwise) against the oracle: beliefs of every column where they are returned, hard bits always."""
    import torch
    from neoradium_amd import ops, _lib
    kb, core, rows_all, ncols = (22, 26, 46, 68) if bg == 1 else (10, 14, 42, 52)
    ils = next(i for i, b in enumerate((2, 3, 5, 7, 9, 11, 13, 15)) if zc % b == 0 and (zc // b) & (zc // b - 1) == 0)
    cfg = _lib.LdpcCfg()
    cfg.bg, cfg.Zc, cfg.iLS, cfg.K, cfg.N, cfg.F, cfg.C, cfg.B, cfg.cb_len = bg, zc, ils, kb * zc, (ncols - 2) * zc, 0, 1, 0, 0
    rng = np.random.default_rng(zc + n_tx_cols + all_rows)
    n_cb = 4
    llr = np.full((n_cb, cfg.N), -6.0e4)
    llr[1] *= rng.choice([-1.0, 1.0], cfg.N)                    # random signs
    llr[2, rng.random(cfg.N) < 0.3] = 6.0e4                     # mostly negative
    llr[3] = np.where(rng.random(cfg.N) < 0.5, -6.0e4, -1.0e10)  # two saturated levels (1e10 = the clipping level)
    e_max = n_tx_cols * zc
    llr[:, e_max:] = 0.0
    rows = None if all_rows else ops.ldpc_active_rows(cfg, e_max)
    n_it = 6
    ref = oc.decode(llr, bg, ils, zc, num_iter=n_it, only_info=False, belief=True, rows=rows)
    # the corner is actually reached: a decoder that hands min2 to every tied entry differs from the reference here
    x = torch.from_numpy(llr).to(dev)
    hard = ops.ldpc_decode(x, cfg, n_it, rows=rows).cpu().numpy()
    assert np.array_equal(hard, (ref[:, :kb * zc] < 0).astype(np.uint8))
    if all_rows:
        bel = ops.ldpc_decode(x, cfg, n_it, only_info=False, belief=True).cpu().numpy()
        assert np.array_equal(bel, ref), f"max diff {np.abs(bel - ref).max()}"


def test_onchip_decoder_runs_exactly_the_rows_asked_for(dev):
    """nrx_ldpc_decode_f64 with n_rows below the row count of the on-chip instantiation that serves it (13 or 15 rows at Zc 384):
    the rows beyond n_rows must not see their extension LLRs even when the caller left something other than zero there -- the
    workspace kernel (every other lifting size) runs exactly n_rows rows, and so must this one."""
    import torch
    from neoradium_amd import ops, _lib
    cfg = _lib.LdpcCfg()
    cfg.bg, cfg.Zc, cfg.iLS, cfg.K, cfg.N, cfg.F, cfg.C, cfg.B, cfg.cb_len = 1, 384, 1, 22 * 384, 66 * 384, 0, 1, 0, 0
    rng = np.random.default_rng(11)
    llr = 2 / 0.9 ** 2 + (2 / 0.9) * rng.standard_normal((4, cfg.N))        # NOTHING punctured: every extension column is live
    for rows in (6, 11, 14):
        ref = oc.decode(llr, 1, 1, 384, num_iter=7, rows=rows)
        got = ops.ldpc_decode(torch.from_numpy(llr).to(dev), cfg, 7, rows=rows).cpu().numpy()
        assert np.array_equal(got, ref.astype(np.uint8)), rows


ALL_Z = sorted(b * 2 ** k for b in (2, 3, 5, 7, 9, 11, 13, 15) for k in range(8) if b * 2 ** k <= 384)


@pytest.mark.parametrize("bg", [1, 2])
def test_onchip_decoder_every_lifting_size(dev, bg):
    """The run-time-Zc on-chip float64 kernel (nrx_ldpc_dec4.hip: <= 15 rows, any lifting size, both base graphs) gives
    the bits of the workspace kernel (NRX_LDPC_NOCHIP64, read at every call) for every one of the 51 lifting sizes -- ragged
    batches (a last workgroup with empty code-block slots), filler LLRs (1e20: the +1e5 quirk path), exact zeros, row counts
    4..15 -- and the oracle's bits for the sizes it finishes quickly."""
    import os
    import torch
    from neoradium_amd import ops, _lib
    from oracle import coding as oc
    kb, core, ncols = (22, 26, 68) if bg == 1 else (10, 14, 52)
    assert len(ALL_Z) == 51
    rng = np.random.default_rng(4000 + bg)
    assert 'NRX_LDPC_NOCHIP64' not in os.environ
    for i, zc in enumerate(ALL_Z):
        ils = next(k for k, b in enumerate((2, 3, 5, 7, 9, 11, 13, 15)) if zc % b == 0 and (zc // b) & (zc // b - 1) == 0)
        cfg = _lib.LdpcCfg()
        cfg.bg, cfg.Zc, cfg.iLS, cfg.K, cfg.N, cfg.F, cfg.C, cfg.B, cfg.cb_len = bg, zc, ils, kb * zc, (ncols - 2) * zc, 0, 1, 0, 0
        rows = 4 + (i * 5) % 12                               # 4..15
        n_cb = 1 + (i * 7) % 29
        sig = 0.8 if bg == 1 else 1.0
        llr = 2 / sig ** 2 + (2 / sig) * rng.standard_normal((n_cb, cfg.N))
        llr[:, (core - 2 + rows - 4) * zc - zc // 3:] = 0.0   # nothing received beyond the rows that run
        llr[rng.random(llr.shape) < 0.002] = 0.0
        nf = zc // 2
        llr[:, (kb - 2) * zc - nf:(kb - 2) * zc] = 1e20       # fillers at the end of the information columns
        x = torch.from_numpy(llr).to(dev)
        got = ops.ldpc_decode(x, cfg, 9, rows=rows)
        os.environ['NRX_LDPC_NOCHIP64'] = '1'
        try:
            ref = ops.ldpc_decode(x, cfg, 9, rows=rows)
        finally:
            del os.environ['NRX_LDPC_NOCHIP64']
        assert torch.equal(got, ref), (bg, zc, rows, n_cb)
        if zc == 384 and bg == 1:                              # the Zc = 384 specialisation served `got`: the generic kernel too
            os.environ['NRX_LDPC_NOCHIP384'] = '1'
            try:
                gen = ops.ldpc_decode(x, cfg, 9, rows=rows)
            finally:
                del os.environ['NRX_LDPC_NOCHIP384']
            assert torch.equal(gen, ref), (zc, rows)
        if zc <= 40 or zc in (88, 208):
            o = oc.decode(llr[:4], bg, ils, zc, num_iter=9, rows=rows)
            assert np.array_equal(o, got[:4].cpu().numpy()), (bg, zc, rows)


@pytest.mark.parametrize("bg,zc,rows", [(1, 352, 16), (1, 352, 31), (1, 352, 46), (1, 208, 32), (1, 36, 46), (1, 2, 20),
                                        (2, 256, 16), (2, 256, 21), (2, 256, 23), (2, 256, 42), (2, 320, 22), (2, 104, 30), (2, 13, 42)])
def test_hybrid_decoder_at_any_lifting_size(dev, bg, zc, rows):
    """More than 15 rows at lifting sizes other than 384 (BG2 at its usual rates, low-rate BG1 blocks): the hybrid instantiations of
    nrx_ldpc_dec4.hip -- BG1 31 / 46 rows, BG2 22 / 42 rows, the sparse rows' state streamed through the workspace, lifting size at
    run time, several code blocks per workgroup, partial last waves -- give the bits of the workspace kernel (NRX_LDPC_NOHYBRID,
    read at every call) on a ragged batch with filler LLRs and exact zeros, and the oracle's bits."""
    import os
    import torch
    from neoradium_amd import ops, _lib
    from oracle import coding as oc
    kb, ncols = (22, 68) if bg == 1 else (10, 52)
    ils = next(i for i, b in enumerate((2, 3, 5, 7, 9, 11, 13, 15)) if zc % b == 0 and (zc // b) & (zc // b - 1) == 0)
    cfg = _lib.LdpcCfg()
    cfg.bg, cfg.Zc, cfg.iLS, cfg.K, cfg.N, cfg.F, cfg.C, cfg.B, cfg.cb_len = bg, zc, ils, kb * zc, (ncols - 2) * zc, 0, 1, 0, 0
    rng = np.random.default_rng(7000 + 100 * bg + zc + rows)
    n_cb = 2 * (12 // -(-zc // 64)) + 1                        # two full workgroups and a lone block
    llr = 2 / 0.9 ** 2 + (2 / 0.9) * rng.standard_normal((n_cb, cfg.N))
    n_rx = (kb - 2 + rows) * zc - zc // 3                      # nothing received beyond the rows that run (last column partly)
    llr[:, n_rx:] = 0.0
    llr[rng.random(llr.shape) < 0.002] = 0.0
    if zc >= 16:
        llr[:, (kb - 2) * zc - zc // 3:(kb - 2) * zc] = 1e20   # fillers at the end of the information columns
    x = torch.from_numpy(llr).to(dev)
    assert 'NRX_LDPC_NOHYBRID' not in os.environ
    got = ops.ldpc_decode(x, cfg, 9, rows=rows)
    os.environ['NRX_LDPC_NOHYBRID'] = '1'
    try:
        ref = ops.ldpc_decode(x, cfg, 9, rows=rows)
    finally:
        del os.environ['NRX_LDPC_NOHYBRID']
    assert torch.equal(got, ref)
    o = oc.decode(llr[:2], bg, ils, zc, num_iter=9, rows=rows)
    assert np.array_equal(o, got[:2].cpu().numpy())


@pytest.mark.parametrize("rows", [16, 31, 32, 46])
def test_hybrid_decoder_for_more_than_15_rows(dev, rows):
    """More than 15 rows at Zc = 384 (rates below ~0.6, HARQ retransmissions, all 46 rows): the hybrid instantiations of
    nrx_ldpc_dec3.hip -- 31 or 46 rows of the graph, the sparse rows' check-node state streamed through the workspace -- give the
    bits of the workspace kernel (NRX_LDPC_NOHYBRID, read at every call) on a ragged batch with filler LLRs and exact zeros,
    and the oracle's bits."""
    import os
    import torch
    from neoradium_amd import ops, _lib
    from oracle import coding as oc
    cfg = _lib.ldpc_config(1, 25000 + 24)
    assert cfg.Zc == 384
    rng = np.random.default_rng(500 + rows)
    n_cb = 7
    llr = 2 / 0.85 ** 2 + (2 / 0.85) * rng.standard_normal((n_cb, cfg.N))
    llr[:, (22 + rows - 4) * 384 - 100:] = 0.0                 # nothing received beyond the rows that run
    llr[rng.random(llr.shape) < 0.002] = 0.0
    llr[:, 20 * 384 - 150:20 * 384] = 1e20                    # fillers at the end of the information columns
    x = torch.from_numpy(llr).to(dev)
    assert 'NRX_LDPC_NOHYBRID' not in os.environ
    got = ops.ldpc_decode(x, cfg, 11, rows=rows)
    os.environ['NRX_LDPC_NOHYBRID'] = '1'
    try:
        ref = ops.ldpc_decode(x, cfg, 11, rows=rows)
    finally:
        del os.environ['NRX_LDPC_NOHYBRID']
    assert torch.equal(got, ref)
    o = oc.decode(llr[:2], 1, 1, 384, num_iter=11, rows=rows)
    assert np.array_equal(o, got[:2].cpu().numpy())


@pytest.mark.parametrize("extra_env", [{}, {"NRX_LDPC_ALLROWS": "1"}, {"NRX_LDPC_NOSPEC": "1"}])
def test_rate_recovering_demapper_leaves_nothing_the_decoder_reads_uninitialised(dev, extra_env):
    """nrx_qam_demap_rr_* initialises only the columns the decoder will read (ops.ldpc_rows_read).  With the buffer poisoned with NaN
    first (NRX_DEBUG_POISON) the float32 chain must count exactly the errors of the route through nrx_ldpc_rate_recover, which
    zero-fills everything -- also under the developer switches that make the float32 decoder read every row (a child process: the
    library reads them once)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = (
        "import sys, json; sys.path.insert(0, %r)\n"
        "import neoradium_amd as nr, bench\n"
        "link = bench.build_link(nr, decoder='f32', num_iter=10)\n"
        "print(json.dumps(link.run(10, 6, 31.0, seed=7).cpu().tolist()))\n" % root)
    res = []
    for env_add in (dict(extra_env, NRX_DEBUG_POISON="1"), dict(extra_env, NRX_SEPARATE_RATE_RECOVERY="1")):
        env = dict(os.environ, **env_add)
        r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(json.loads(r.stdout.strip().split("\n")[-1]))
    assert res[0] == res[1] and res[0][1] == 6 * 72 and 0 < res[0][0] < res[0][1], res
