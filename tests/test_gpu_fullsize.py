"""GPU parity at BASELINE.json's FULL sizes (273 PRB / nFFT 4096 / 72-113 code blocks of Zc = 384) through properties
that need no oracle run (the reference cannot build a 273-PRB carrier; the CPU oracle needs 7 s per slot and is compared
at this size inside bench.py): encode -> erase -> decode round trips, CRC of CRC, OFDM round trip, linearity of the
channel filter, batch-split invariance of the counters, float32 vs float64 decoder verdicts."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope='module')
def link(dev):
    import bench
    import neoradium_amd as nr
    return bench.build_link(nr, decoder="f32", num_iter=20)


def test_metric_sizes(link):
    c = link.cfg
    assert (link.tbs, c.C, c.Zc, c.K, c.N, link.G, link.nfft, link.K) == (606504, 72, 384, 8448, 25344, 943488, 4096, 3276)


@pytest.mark.parametrize("qm,nl", [(6, 4), (8, 4)])
def test_coding_round_trip_with_erasures(link, dev, qm, nl):
    """segment -> encode -> rate match -> (LLRs, 12 % of them erased) -> rate recover -> decode -> CRC/merge returns the
    transport block exactly, in float32 and float64; a code block that received only noise fails alone."""
    import torch
    from neoradium_amd import ops, _lib
    tbs = 606504 if qm == 6 else 950984                   # metric configuration / BASELINE cfg3 (256-QAM, 113 blocks)
    cfg = _lib.ldpc_config(1, tbs + 24)
    G = 39312 * nl * qm
    g = torch.Generator(device=dev)
    g.manual_seed(qm)
    tb = torch.randint(0, 2, (2, tbs), device=dev, generator=g, dtype=torch.uint8)
    coded = ops.ldpc_encode(ops.ldpc_segment(tb, cfg), cfg)
    rm = ops.ldpc_rate_match(coded, cfg, G, nl, qm)
    assert tuple(rm.shape) == (2, G)
    lens = _lib.ldpc_cb_lens(G, cfg.C, nl, qm)
    assert sum(lens) == G and all(v % (nl * qm) == 0 for v in lens)
    llr = (1.0 - 2.0 * rm.to(torch.float64)) * 8.0
    erase = torch.rand((2, G), device=dev, generator=g) < 0.12
    llr[erase] = 0.0
    # code block 5 of transport block 1 receives noise only (random signs at full confidence).  (Erasing it would not
    # do: all-zero LLRs decode to the all-zero word, which passes the zero-initialised CRC24B -- as in the reference.)
    dead = sum(lens[:5])
    llr[1, dead:dead + lens[5]] = 8.0 - 16.0 * torch.randint(0, 2, (lens[5],), device=dev, generator=g).to(torch.float64)
    for ft in (torch.float32, torch.float64):
        rr = ops.ldpc_rate_recover(llr.to(ft), cfg, nl, qm)
        dec = ops.ldpc_decode(rr, cfg, 12)
        tb_out, cb_ok, tb_ok = ops.ldpc_crc_merge(dec, cfg)
        ok = cb_ok.cpu().numpy().astype(bool)
        assert ok[0].all() and ok[1].sum() == cfg.C - 1 and not ok[1, 5]
        assert torch.equal(tb_out[0, :tbs], tb[0]) and bool(tb_ok.cpu().numpy()[0]) and not bool(tb_ok.cpu().numpy()[1])
        # the other 71/112 blocks of the damaged transport block are exact too
        seg = ops.ldpc_segment(tb[1:2], cfg)
        good = [i for i in range(cfg.C) if i != 5]
        assert torch.equal(dec[cfg.C:][good], seg[good])


def test_ofdm_round_trip_and_filter_linearity(link, dev):
    """273 PRB, nFFT 4096: demodulate(modulate(X)) = X to 1e-12; the tapped-delay-line filter is linear."""
    import torch
    from neoradium_amd import ops
    rng = np.random.default_rng(3)
    cps = [int(v) for v in (link.sym_lens[0][:-1] - link.nfft)]
    x = torch.from_numpy(rng.standard_normal((2, 4, 14, 3276)) + 1j * rng.standard_normal((2, 4, 14, 3276))).to(dev)
    w = ops.ofdm_modulate(x, 4096, cps, window_len=0)
    assert w.shape[-1] == link.slot_len[0] == 61440
    y = ops.ofdm_demodulate(w, 4096, cps, 3276)
    assert float((y - x).abs().max()) <= 1e-12 * float(x.abs().max())
    # float32 waveform chain: complex64 modulator / demodulator to float32 accuracy, and the demodulator's complex128 output
    # option (float32 transform, float64 grid: the hand-over to the float64 estimator) = the complex64 grid converted
    w32 = ops.ofdm_modulate(x.to(torch.complex64), 4096, cps, window_len=0)
    y32 = ops.ofdm_demodulate(w32, 4096, cps, 3276)
    assert y32.dtype == torch.complex64 and float((y32.to(torch.complex128) - x).abs().max()) <= 2e-5 * float(x.abs().max())
    assert torch.equal(ops.ofdm_demodulate(w32, 4096, cps, 3276, grid64=True), y32.to(torch.complex128))
    sg = torch.tensor([0.1, 0.2], dtype=torch.float64, device=dev)
    assert torch.equal(ops.ofdm_demodulate(w32, 4096, cps, 3276, awgn=(sg, 5, 2, 7), grid64=True),
                       ops.ofdm_demodulate(w32, 4096, cps, 3276, awgn=(sg, 5, 2, 7)).to(torch.complex128))
    times = torch.from_numpy(link.gain_times(np.arange(2))).to(dev)
    gains = ops.cdl_gains(link.A, link.nu, times, A_los=link.Alos, nu_los=link.nulos)
    sl = [int(v) for v in link.sym_lens[0]]
    a = 0.3 - 1.1j
    w2 = torch.flip(w, dims=[0])
    f = lambda s: ops.apply_td_paths(s, gains, link.taps, link.tap_off, sl, hist=link.td_hist)
    lhs, rhs = f(a * w + w2), a * f(w) + f(w2)
    assert float((lhs - rhs).abs().max()) <= 1e-12 * float(rhs.abs().max())


def test_engine_counters_at_metric_size(link, dev):
    """Whole slots at the metric configuration: counters do not depend on how a slot range is batched (the basis of the
    multi-GPU sharding), clean at very high SNR, and the float64 decoder reaches the same CRC verdicts there."""
    import torch
    import bench
    import neoradium_amd as nr
    whole = link.run(10, 4, 33.0, seed=7).cpu().numpy()
    split = (link.run(10, 1, 33.0, seed=7) + link.run(11, 3, 33.0, seed=7)).cpu().numpy()
    assert np.array_equal(whole, split) and whole[1] == 4 * 72 and whole[3] == 4 * 606504
    hi = link.run(0, 2, 60.0, seed=7).cpu().numpy()
    assert hi[0] == 0 and hi[2] == 0
    l64 = bench.build_link(nr, decoder="f64", num_iter=20)
    _, d32 = link.run(3, 2, 36.0, seed=11, details=True)
    _, d64 = l64.run(3, 2, 36.0, seed=11, details=True)
    a, b = d32[0][1]['cb_ok'].cpu().numpy(), d64[0][1]['cb_ok'].cpu().numpy()
    # float32 fast mode against the float64 chain: measured verdict disagreement 1.1e-3 over 147 168 code blocks across the
    # waterfall of this configuration, worst SNR point 2.4e-3 (profiles/r2_f32_vs_f64_verdicts_metric.json, tools/
    # verdict_compare.py); on these 144 blocks that is an expectation of 0.2 -- more than 2 differing verdicts is a regression
    assert int((a != b).sum()) <= 2
    rel = float((d32[0][1]['llr'].double() - d64[0][1]['llr']).abs().max() / d64[0][1]['llr'].abs().max())
    assert rel <= 1e-5                        # north-star tolerance on float LLRs
    # waveform="f32" (opt-in, with the float32 decoder = the bench's fast mode): Tx grid, OFDM, channel filter and received grid in
    # complex64.  Same generator keys, so the same transport blocks and noise draws: the equalised LLRs stay within float32
    # accuracy of the float64 chain's and the verdicts within the fast mode's disagreement rate.
    lw = bench.build_link(nr, decoder="f32", num_iter=20, waveform="f32")
    _, dw = lw.run(3, 2, 36.0, seed=11, details=True)
    c = dw[0][1]['cb_ok'].cpu().numpy()
    assert int((c != b).sum()) <= 2
    scale = float(d64[0][1]['llr'].abs().max())
    err = (dw[0][1]['llr'].double() - d64[0][1]['llr']).abs()
    assert float(err.max()) <= 2e-3 * scale and float(err.mean()) <= 1e-4 * scale, (float(err.max()) / scale, float(err.mean()) / scale)
    assert torch.equal(lw.run(10, 4, 33.0, seed=7), lw.run(10, 1, 33.0, seed=7) + lw.run(11, 3, 33.0, seed=7))


def test_multi_pass_schedule_at_metric_size(dev):
    """The opt-in multi-pass schedule at the metric configuration (fused float64 entry, failing blocks continued from their
    parked decoder state with the list kept on the device): at the waterfall, where blocks pass early, late and never, the CRC
    verdicts and decoded transport blocks of every slot equal those of the reference schedule (a fixed 30 iterations here),
    for one check and for several; the restart form of the host-compacted path is covered by the small-size engine test."""
    import torch
    import bench
    import neoradium_amd as nr
    ref = bench.build_link(nr, decoder="f64", num_iter=30)
    want = ref.run(4, 3, 31.0, seed=21, details="verdicts")[1][0][1]
    n_ok = int(want['cb_ok'].sum())
    assert 0 < n_ok < want['cb_ok'].numel()
    for marks in (6, (5, 12), (4, 9, 17)):
        mp = bench.build_link(nr, decoder="f64", num_iter=30, firstPassIter=marks)
        got = mp.run(4, 3, 31.0, seed=21, details="verdicts")[1][0][1]
        assert torch.equal(got['cb_ok'], want['cb_ok']) and torch.equal(got['tb_out'], want['tb_out']), marks
        del mp
    # nothing fails the first check: the continuation launches find an empty list
    mp = bench.build_link(nr, decoder="f64", num_iter=30, firstPassIter=(5, 12))
    hi_ref = ref.run(0, 2, 60.0, seed=3, details="verdicts")[1][0][1]
    hi = mp.run(0, 2, 60.0, seed=3, details="verdicts")[1][0][1]
    assert bool(hi['cb_ok'].all()) and torch.equal(hi['cb_ok'], hi_ref['cb_ok']) and torch.equal(hi['tb_out'], hi_ref['tb_out'])
    with pytest.raises(ValueError):
        bench.build_link(nr, decoder="f64", num_iter=30, firstPassIter=(9, 9))


def test_cfg3_link_at_273_prb(dev):
    """BASELINE cfg3 through the engine at its full size: 273 PRB, 256-QAM, 4 layers, 4x4 CDL-D 300 ns, BG1 R = 0.75 =>
    TBS 950 984, 113 code blocks of Zc 384.  With perfect CSI (frequency-domain channel) the link is clean at
    high SNR; with the DMRS-LS estimate every block fails at any SNR -- the reference's own behaviour for this channel
    (tests/golden/e2e_cfg3_*: 24 PRB, generated from the reference), so it is asserted, not hidden."""
    import neoradium_amd as nr
    nr.random.setSeed(123)
    car = nr.Carrier(numRbs=273, spacing=30)
    p = nr.PDSCH(car.curBwp, numLayers=4, nID=car.cellId, modulation='256QAM')
    p.setDMRS(configType=1, additionalPos=1)
    ch = nr.CdlChannel(car.curBwp, 'D', delaySpread=300, carrierFreq=4e9, dopplerShift=5,
                       txAntenna=nr.AntennaPanel([1, 2], polarization="x"), rxAntenna=nr.AntennaPanel([1, 2], polarization="x"))
    perfect = nr.PdschLink(p, ch, 0.75, baseGraphNo=1, numIter=20, freqDomain=True, chanEst="Perfect", decoder="f64")
    c = perfect.cfg
    assert (perfect.tbs, c.C, c.Zc, perfect.G) == (950984, 113, 384, 1257984)
    hi = perfect.run(0, 3, 60.0, seed=5).cpu().numpy()
    assert hi[0] == 0 and hi[2] == 0 and hi[1] == 3 * 113 and hi[3] == 3 * 950984
    ls = nr.PdschLink(p, ch, 0.75, baseGraphNo=1, numIter=20, freqDomain=False, chanEst="LS", decoder="f64")
    lo = ls.run(0, 2, 60.0, seed=5).cpu().numpy()
    assert lo[0] == lo[1] == 2 * 113


def test_headline_path_at_273_prb_against_separate_stages_and_oracle(dev):
    """The path bench.py times -- float64 chain, demapper writing per code block (nrx_qam_demap_cb_f64), rate recovery + on-chip
    decode + CRC/merge in one launch (nrx_ldpc_recover_decode_merge_f64), 50 iterations -- at the metric configuration, on host-
    supplied transport blocks and noise (parity mode):
      (i)  CRC verdicts and decoded transport-block bits identical to the separate stages (details=True: symbol-major demapper,
           nrx_ldpc_rate_recover_f64, nrx_ldpc_decode_f64, nrx_ldpc_crc_merge) at a waterfall SNR where blocks fail;
      (ii) at a clear-cut SNR the CRC vector equals oracle.link.run_slot's and the LLRs agree to 1e-9 of the slot's scale."""
    import torch
    import bench
    import neoradium_amd as nr
    from neoradium_amd._dev import D
    from neoradium_amd import ops
    from oracle import link as olink
    link = bench.build_link(nr, decoder="f64", num_iter=50)
    cw = link.cw[0]
    assert ops.ldpc_fused_supported(cw['cfg'], cw['nl'], cw['qm'], cw['G'], cw['rows']) and cw['rows'] == 15
    rng = np.random.default_rng(2026)
    n = 2
    tb = rng.integers(0, 2, (n, link.tbs)).astype(np.uint8)
    z = rng.standard_normal((n, link.nr, link.slot_len[0] + link.max_delay, 2))
    zc = z[..., 0] + 1j * z[..., 1]
    tbt, zt = torch.from_numpy(tb), D(zc)
    # (i) waterfall: some of the 144 blocks fail, non-converging blocks included
    c_f, dv = link.run(5, n, 31.0, tb_bits=tbt, noise=zt, details="verdicts")
    c_s, ds = link.run(5, n, 31.0, tb_bits=tbt, noise=zt, details=True)
    v, s = dv[0][1], ds[0][1]
    assert torch.equal(v['cb_ok'], s['cb_ok']) and torch.equal(v['tb_out'], s['tb_out']) and torch.equal(c_f, c_s)
    n_ok = int(v['cb_ok'].sum())
    assert 0 < n_ok < v['cb_ok'].numel(), f"want passing and failing blocks at 31 dB, got {n_ok}"
    # (ii) clear-cut SNR against the CPU oracle (its own Tx chain, channel, estimator, equaliser, demapper, decoder)
    _, dv = link.run(5, n, 38.0, tb_bits=tbt, noise=zt, details="verdicts")
    _, ds = link.run(5, n, 38.0, tb_bits=tbt, noise=zt, details=True)
    v, s = dv[0][1], ds[0][1]
    assert torch.equal(v['cb_ok'], s['cb_ok']) and torch.equal(v['tb_out'], s['tb_out'])
    st = olink.static_from_link(link, slots=range(5, 5 + n))
    F = s['F'].cpu().numpy()
    jobs = [(st, 5 + i, 38.0, tb[i].astype(np.int8), zc[i], F[i]) for i in range(n)]
    refs = olink.run_slots_parallel(jobs, n)
    for i, ref in enumerate(refs):
        got = s['llr'][i].cpu().numpy()
        scale = np.abs(ref['llr']).max()
        assert np.abs(got - ref['llr']).max() <= 1e-9 * scale
        assert np.array_equal(v['cb_ok'][i].cpu().numpy().astype(bool), ref['crc'])
        nb = len(ref['tb_out'])
        assert np.array_equal(v['tb_out'][i].cpu().numpy()[:nb], ref['tb_out'].astype(np.uint8))


def test_folded_precoder_against_the_reference_operation_order(dev):
    """The default time-domain link folds the wideband precoder into the channel filter's gains (sum_t g F computed before filtering):
    the reference precodes the grid first (pdsch.py / grid.py order).  NRX_SEPARATE_PRECODER=1 restores that order; the two differ
    by reassociation only: LLRs within 1e-9 of their scale, CRC verdicts equal away from the waterfall."""
    import os
    import bench
    import neoradium_amd as nr
    folded = bench.build_link(nr, decoder="f64", num_iter=10)
    os.environ['NRX_SEPARATE_PRECODER'] = '1'
    try:
        separate = bench.build_link(nr, decoder="f64", num_iter=10)
    finally:
        del os.environ['NRX_SEPARATE_PRECODER']
    assert separate._sep_prec and not folded._sep_prec
    _, d0 = folded.run(77, 2, 36.0, seed=9, details=True)
    _, d1 = separate.run(77, 2, 36.0, seed=9, details=True)
    a, b = d0[0][1]['llr'], d1[0][1]['llr']
    scale = float(a.abs().max())
    assert float((a - b).abs().max()) <= 1e-9 * scale
    assert (d0[0][1]['cb_ok'] == d1[0][1]['cb_ok']).all()
