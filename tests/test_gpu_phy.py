"""GPU parity: modem / grid / OFDM / channel / estimator kernels through the C ABI vs the NumPy oracle (float64).

Tolerances (written out per the north star): complex128 paths <= 1e-10 relative to the array's scale (FFT, FIR,
interpolation and 4x4 solves are mathematically defined; only rounding order differs); LLRs <= 1e-9 absolute in
float64 and <= 1e-5 relative to the LLR scale in float32.
"""
import os

import numpy as np
import pytest

from oracle import phy as op

pytestmark = pytest.mark.gpu


def T(x, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def crandn(rng, *shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2)


@pytest.mark.parametrize("mod", ['BPSK', 'QPSK', '16QAM', '64QAM', '256QAM', '1024QAM'])
def test_modem(dev, mod):
    import torch
    from neoradium_amd import ops
    qm = op.QM[mod]
    rng = np.random.default_rng(qm)
    n, ns = 3, 257
    bits = rng.integers(0, 2, (n, ns * qm)).astype(np.uint8)
    ref = np.stack([op.modulate(b, qm) for b in bits])
    out = ops.qam_map(T(bits, dev), qm).cpu().numpy()
    assert np.array_equal(out, ref)                                   # constellation points are exact
    y = ref + 0.15 * crandn(rng, n, ns)
    nv = np.array([0.02, 0.05, 0.3])
    llr = ops.qam_demap(T(y, dev), T(nv, dev), qm).cpu().numpy()
    refl = np.stack([op.demap_maxlog(y[i], nv[i], qm) for i in range(n)])
    assert np.abs(llr - refl).max() <= 1e-9
    assert np.array_equal(llr < 0, refl < 0) or np.abs(refl[(llr < 0) != (refl < 0)]).max() < 1e-9
    if qm >= 2:
        # bit for bit the per-axis form in IEEE float64 with true divisions, (-m0/nv) - (-m1/nv): the kernel's quotients come
        # from one reciprocal per item and a correction step (nrx_modem.hip div_by) and must be the correctly rounded ones
        h = qm // 2
        cst = op.constellation(qm)
        # level of real-axis bit pattern a (bits b0 b2 b4 .. of the symbol, MSB first), as the kernel forms it: integer * (1 / sqrt(norm))
        idx = [sum(((a >> (h - 1 - i)) & 1) << (qm - 1 - 2 * i) for i in range(h)) for a in range(1 << h)]
        lev = np.rint(cst[idx].real * np.sqrt(op._NORM[qm])) * (1.0 / np.sqrt(float(op._NORM[qm])))
        want = np.empty_like(llr)
        for axis, yv in enumerate((y.real, y.imag)):
            d2 = (yv[..., None] - lev) * (yv[..., None] - lev)
            for q in range(h):
                one = np.array([(a >> (h - 1 - q)) & 1 for a in range(1 << h)], bool)
                m0, m1 = d2[..., ~one].min(-1), d2[..., one].min(-1)
                want[:, 2 * q + axis::qm] = (-m0 / nv[:, None]) - (-m1 / nv[:, None])
        assert np.array_equal(llr, want)
    llr32 = ops.qam_demap(T(y, dev), T(nv, dev), qm, llr_dtype=torch.float32).cpu().numpy()
    assert np.abs(llr32 - refl).max() <= 1e-5 * np.abs(refl).max()
    if qm <= 6:
        ex = ops.qam_demap(T(y, dev), T(nv, dev), qm, exact=True).cpu().numpy()
        refe = np.stack([op.demap_exact(y[i], nv[i], qm) for i in range(n)])
        assert np.abs(ex - refe).max() <= 1e-8 * max(1.0, np.abs(refe).max())


def test_pdsch_map_demap_with_scrambling_and_index(dev):
    from neoradium_amd import ops
    rng = np.random.default_rng(3)
    qm, nl, L, K = 6, 2, 14, 48
    elems = nl * L * K
    nsym = 500
    re_index = rng.permutation(elems)[:nsym].astype(np.int32)
    cinit = op.pdsch_scramble_cinit(1, 0, 17)
    scr = op.gold(cinit, nsym * qm).astype(np.uint8)
    bits = rng.integers(0, 2, (2, nsym * qm)).astype(np.uint8)
    import torch
    grid = torch.zeros((2, nl, L, K), dtype=torch.complex128, device=dev)
    ops.qam_map(T(bits, dev), qm, scr=T(scr, dev), re_index=T(re_index, dev), out=grid)
    g = grid.cpu().numpy().reshape(2, -1)
    for b in range(2):
        assert np.array_equal(g[b][re_index], op.modulate(bits[b] ^ scr, qm))
        mask = np.ones(elems, bool); mask[re_index] = False
        assert not g[b][mask].any()
    eq = g + 0.05 * crandn(rng, 2, elems)
    scales = rng.uniform(0.5, 20, (2, elems))
    nv = np.array([1e-12, 0.01])                                     # first one exercises the 1e-10 floor
    llr = ops.qam_demap(T(eq, dev), T(nv, dev), qm, scr=T(scr, dev), re_index=T(re_index, dev), scales=T(scales, dev),
                        nv_floor=1e-10).cpu().numpy()
    for b in range(2):
        ref = op.pdsch_llrs(eq[b][re_index], scales[b][re_index], nv[b], qm, cinit)
        assert np.abs(llr[b] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("nr,nl", [(1, 1), (2, 1), (2, 2), (4, 2), (4, 4), (8, 3)])
def test_precode_channel_equalize(dev, nr, nl):
    from neoradium_amd import ops
    rng = np.random.default_rng(10 * nr + nl)
    n, L, K, nt = 2, 14, 36, max(nr, nl) * 2
    x = crandn(rng, n, nl, L, K)
    f = crandn(rng, n, nt, nl)
    pg = ops.precode(T(x, dev), T(f, dev)).cpu().numpy()
    assert rel(pg, np.stack([op.precode(x[i], f[i]) for i in range(n)])) < 1e-13
    h = crandn(rng, n, L, K, nr, nt)
    rx = ops.apply_channel_fd(T(pg, dev), T(h, dev)).cpu().numpy()
    assert rel(rx, np.stack([op.apply_channel_fd(pg[i], h[i]) for i in range(n)])) < 1e-13
    hf = h @ f[:, None, None]
    nv = np.array([1e-9, 0.03])
    eq, sc = ops.mmse_equalize(T(rx, dev), T(hf, dev), T(nv, dev))
    for i in range(n):
        e, s = op.equalize_mmse(rx[i], hf[i], nv[i])
        assert rel(eq[i].cpu().numpy(), e) < 1e-10
        assert rel(sc[i].cpu().numpy(), s) < 1e-10


def test_noise_level_and_noise(dev):
    import torch
    from neoradium_amd import ops
    rng = np.random.default_rng(4)
    x = crandn(rng, 3, 4, 14, 60) * np.array([1.0, 3.0, 0.1])[:, None, None, None] + 0.2
    snr_db = np.array([0.0, 10.0, 25.5])
    var, sigma, nv = ops.noise_level(T(x, dev), snr_lin=10 ** (snr_db / 10))
    for i in range(3):
        assert abs(var[i].item() - np.var(x[i])) <= 1e-13 * np.var(x[i])
        assert abs(sigma[i].item() - op.noise_std_grid(x[i], snr_db[i])) <= 1e-13 * sigma[i].item()
        assert abs(nv[i].item() - sigma[i].item() ** 2) <= 1e-15
    z = rng.standard_normal((3, 4, 14, 60, 2))
    zc = z[..., 0] + 1j * z[..., 1]
    y = ops.add_noise(T(x, dev), T(zc, dev), sigma).cpu().numpy()
    s = sigma.cpu().numpy()
    ref = x + (z * (s / np.sqrt(2))[:, None, None, None, None]) @ np.array([1, 1j])
    assert rel(y, ref) < 1e-15
    # counter-based generator: statistics + independence of the batch split, for both transforms (nrx_rng.h normal_pair: Box-Muller on
    # the float32 transcendental unit = the default, or in float64 like the reference's normals, random.py:203)
    big = torch.zeros((4, 200000), dtype=torch.complex128, device=dev)
    sg = torch.tensor([1.0, 2.0, 0.5, 1.0], dtype=torch.float64, device=dev)
    assert ops.noise_precision() == ('f64' if os.environ.get('NRX_RNG_F64', '0') not in ('', '0') else 'f32')
    was = ops.noise_precision() == 'f64'
    draws = {}
    try:
        for f64 in (False, True):
            ops.set_noise_precision(f64)
            assert ops.noise_precision() == ('f64' if f64 else 'f32')
            a = draws[f64] = ops.awgn(big, sg, seed=99).cpu().numpy()
            for i in range(4):
                assert abs(np.var(a[i]) - sg[i].item() ** 2) < 0.02 * sg[i].item() ** 2
                assert abs(a[i].mean()) < 0.02 and abs(np.var(a[i].real) - np.var(a[i].imag)) < 0.03 * sg[i].item() ** 2
            b = ops.awgn(big[2:], sg[2:], seed=99, batch_offset=2).cpu().numpy()
            assert np.array_equal(a[2:], b)
            # ... and its shape: the components are N(0, sigma^2 / 2) -- fourth moment, 3-sigma tail mass, no correlation between the
            # components, uniform phase
            for i in range(4):
                c = np.concatenate([a[i].real, a[i].imag]) / (sg[i].item() / np.sqrt(2))
                assert abs(np.mean(c ** 4) - 3.0) < 0.06 and abs(np.mean(np.abs(c) > 3.0) - 0.0026998) < 3e-4
                assert abs(np.mean(a[i].real * a[i].imag)) < 0.01 * sg[i].item() ** 2
                ph = np.histogram(np.angle(a[i]), bins=16, range=(-np.pi, np.pi))[0] / a[i].size
                assert np.abs(ph - 1 / 16).max() < 0.004
            assert np.abs(a[0]).max() > 3.0 and np.isfinite(a).all()
        # the same counters give the same normals up to the float32 transform's error (24-bit angle, radius steps near u1 -> 1);
        # item 0 has sigma 1: the float32 ones are float32 numbers / sqrt 2, the float64 ones are not
        assert 0 < np.abs(draws[True] - draws[False]).max() < 2e-3
        z32, z64 = draws[False][0].real * np.sqrt(2), draws[True][0].real * np.sqrt(2)
        assert np.mean(np.abs(np.float64(np.float32(z32)) - z32) <= 2e-16 * np.abs(z32)) > 0.99
        assert np.mean(np.float64(np.float32(z64)) == z64) < 0.01
    finally:
        ops.set_noise_precision(was)
    with pytest.raises(ValueError):
        from neoradium_amd import _lib
        _lib.check(_lib.lib().nrx_set_noise_precision(2))


@pytest.mark.parametrize("mu,nfft,K,slot", [(0, 2048, 300, 0), (1, 1024, 612, 1), (1, 4096, 3276, 0), (2, 512, 240, 2)])
def test_ofdm_mod_demod(dev, mu, nfft, K, slot):
    import torch
    from neoradium_amd import ops
    rng = np.random.default_rng(nfft + K)
    n, P, L = 2, 2, 14
    grid = crandn(rng, n, P, L, K)
    cps = op.cp_lens_slot(mu, slot, nfft)
    w = op.window_len_std(cps)
    for win in (0, w):
        wave = ops.ofdm_modulate(T(grid, dev), nfft, cps, window_len=win, pad=7).cpu().numpy()
        ref = np.stack([op.ofdm_modulate(grid[i], nfft, cps, window=win > 0) for i in range(n)])
        assert rel(wave[..., :-7], ref) < 1e-12, win
        assert not wave[..., -7:].any()
    # precoder fused into the (symbol-parallel) modulator == precode, then the sequential modulator: same samples
    F = crandn(rng, n, 4, P)
    for win in (0, w):
        fused = ops.ofdm_modulate(T(grid, dev), nfft, cps, window_len=win, pad=5, f=T(F, dev))
        two = ops.ofdm_modulate(ops.precode(T(grid, dev), T(F, dev)), nfft, cps, window_len=win, pad=5)
        assert fused.shape == two.shape == (n, 4, ref.shape[-1] + 5)
        assert np.array_equal(fused.cpu().numpy(), two.cpu().numpy()), win
    # many rows: the symbol-parallel entry (nrx_ofdm_modulate_sym_*) == the sequential one, sample for sample
    big = crandn(rng, 40, P, L, K)
    for win in (0, w):
        par = ops.ofdm_modulate(T(big, dev), nfft, cps, window_len=win, pad=3)              # 80 rows: symbol-parallel
        seq = torch.cat([ops.ofdm_modulate(T(big[i:i + 10], dev), nfft, cps, window_len=win, pad=3) for i in range(0, 40, 10)])
        assert torch.equal(par, seq), win
    # demodulate a time-shifted noisy waveform with per-item timing offsets
    S = ref.shape[-1]
    rxw = np.concatenate([crandn(rng, n, P, 20), ref, crandn(rng, n, P, 30)], axis=-1)
    toff = np.array([20, 17], dtype=np.int32)
    g = ops.ofdm_demodulate(T(rxw, dev), nfft, cps, K, t_off=T(toff, dev)).cpu().numpy()
    for i in range(n):
        assert rel(g[i], op.ofdm_demodulate(rxw[i][:, toff[i]:], nfft, cps, K)) < 1e-12
    # noise generated inside the demodulator's load == awgn, then demodulate (same Philox stream, same samples)
    sig = T(np.float64([0.3, 0.7]), dev)
    noisy = ops.awgn(T(rxw, dev), sig, 77, stream_id=2, batch_offset=5)
    a = ops.ofdm_demodulate(noisy, nfft, cps, K, t_off=T(toff, dev))
    b = ops.ofdm_demodulate(T(rxw, dev), nfft, cps, K, t_off=T(toff, dev), awgn=(sig, 77, 2, 5))
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    # round trip without windowing recovers the grid (cdlTiming.ipynb cell 3: NMSE ~ 1e-32)
    nw = ops.ofdm_modulate(T(grid, dev), nfft, cps, window_len=0)
    back = ops.ofdm_demodulate(nw, nfft, cps, K).cpu().numpy()
    assert rel(back, grid) < 1e-12


def test_chest_mmse_fused_equals_separate(dev):
    """nrx_chest_ls_mmse = nrx_chest_ls + nrx_mmse_equalize, bit for bit (1 and 2 DMRS time groups)."""
    import torch
    from neoradium_amd import ops
    rng = np.random.default_rng(21)
    n, nr, P, L, K = 3, 4, 4, 14, 96
    for ds in ([2], [2, 11]):
        nk = K // 2
        port_ks = np.int32([np.arange(nk) * 2 + (p % 2) for p in range(P)])
        pilots = np.exp(1j * rng.uniform(0, 6.28, (2, P, len(ds), nk)))
        rx = crandn(rng, n, nr, L, K)
        nv = rng.uniform(0.01, 0.1, n)
        pil_set = T(np.int32([0, 1, 0]), dev)
        hest = ops.chest_ls(T(rx, dev), T(pilots, dev), port_ks, ds, l_cdm=1, k_cdm=2, pil_set=pil_set)
        eq0, sc0 = ops.mmse_equalize(T(rx, dev), hest, T(nv, dev))
        eq1, sc1 = ops.chest_ls_mmse(T(rx, dev), T(pilots, dev), T(port_ks, dev), ds, T(nv, dev), l_cdm=1, k_cdm=2,
                                     pil_set=pil_set)
        assert torch.equal(eq0, eq1) and torch.equal(sc0, sc1), ds
        # ... for selected OFDM symbols only (nrx_chest_ls_mmse_syms_f64: the engine skips the symbols without data REs)
        keep = [l for l in range(L) if l not in ds and l != 5]
        mask = sum(1 << l for l in keep)
        eq2, sc2 = ops.chest_ls_mmse(T(rx, dev), T(pilots, dev), T(port_ks, dev), ds, T(nv, dev), l_cdm=1, k_cdm=2,
                                     pil_set=pil_set, sym_mask=mask)
        assert torch.equal(eq0[:, :, keep], eq2[:, :, keep]) and torch.equal(sc0[:, :, keep], sc2[:, :, keep]), ds


def test_tdl_chain(dev):
    """gains -> CIR -> channel matrix / time-domain filtering, CDL-like random static tensors."""
    from neoradium_amd import ops
    rng = np.random.default_rng(8)
    n, nr, nt, N, M, nfft, K, mu = 2, 2, 4, 5, 20, 1024, 300, 1
    A = crandn(rng, nr, nt, N, M) / np.sqrt(M)
    nu = rng.uniform(-50, 50, (N, M))
    Alos, nulos = crandn(rng, nr, nt), 33.0
    fs = 30.72e6
    cps = np.append(op.cp_lens_slot(mu, 0, nfft), op.cp_lens_slot(mu, 1, nfft)[0])
    slot_len = int((cps[:-1] + nfft).sum())
    times = np.stack([op.sym_gain_times(cps, nfft, s * slot_len) / fs for s in (3, 4)])
    gains = ops.cdl_gains(T(A, dev), T(nu, dev), T(times, dev), A_los=T(Alos, dev), nu_los=nulos)
    ph = np.exp(2j * np.pi * times[:, :, None, None] * nu[None, None])
    ref_nlos = np.einsum('rtnm,bcnm->bcrtn', A, ph)
    ref_los = Alos[None, None] * np.exp(2j * np.pi * times * nulos)[:, :, None, None]
    ref_g = np.concatenate([ref_los[..., None], ref_nlos], axis=-1)
    assert rel(gains.cpu().numpy(), ref_g) < 1e-11
    delays = np.sort(rng.uniform(0, 900, N + 1))
    delays[0] = 0
    coeff, _ = op.coeff_matrix(delays, fs, op.build_firs())
    nc = 14
    cir, off = ops.cir(gains, T(coeff, dev), nc)
    for b in range(n):
        c_ref, o_ref = op.cir_from_gains(ref_g[b][:nc], coeff)
        assert rel(cir[b, :nc].cpu().numpy(), c_ref) < 1e-11 and int(off[b]) == o_ref
    H = ops.channel_matrix(cir, off, nc, K, nfft).cpu().numpy()
    for b in range(n):
        c_ref, o_ref = op.cir_from_gains(ref_g[b][:nc], coeff)
        assert rel(H[b], op.channel_matrix(c_ref, o_ref, nfft, K)) < 1e-11
    ns = slot_len + coeff.shape[1] + 5
    x = crandn(rng, n, nt, ns)
    x[..., slot_len:] = 0
    y = ops.apply_td(T(x, dev), cir, list(cps + nfft)).cpu().numpy()
    taps, offs = ops.path_taps(coeff)
    y2 = ops.apply_td_paths(T(x, dev), gains, T(taps, dev), offs, list(cps + nfft)).cpu().numpy()   # path form
    for b in range(n):
        ref_y = op.apply_td(x[b], ref_g[b], coeff, cps + nfft)
        assert rel(y[b], ref_y) < 1e-11 and rel(y2[b], ref_y) < 1e-11
    # the filter that also leaves the noise level of its output over the CP-stripped samples (nrx_apply_td_paths_pow_f64 +
    # nrx_noise_level_finish_f64) against the separate pass (nrx_noise_level_f64 with the gather table) and against NumPy
    # (Waveform.getRePower, waveform.py:107-117; grid.py:1040-1046)
    snr = np.array([3.0, 40.0])
    got = ops.apply_td_paths(T(x, dev), gains, T(taps, dev), offs, list(cps + nfft), power=(nfft, snr, nfft / (12.0 * 25), float(nfft)))
    assert got is not None, "16-tap filter, Nr = 2: the register-tiled kernel applies"
    y5, sg5, nv5 = got
    assert np.array_equal(y5.cpu().numpy(), y2)
    starts = np.concatenate([[0], np.cumsum((cps + nfft)[:-1])])
    o = np.int64(np.round(cps[:-1] * 0.5))
    idx = (starts[:-1, None] + o[:, None] + np.arange(nfft)[None]).reshape(-1)
    var_ref = np.array([np.var(y2[b][:, idx]) for b in range(n)])
    sg_ref = np.sqrt(var_ref * nfft / (12.0 * 25) / snr)
    assert rel(sg5.cpu().numpy(), sg_ref) < 1e-12 and rel(nv5.cpu().numpy(), sg_ref ** 2 * nfft) < 1e-12
    g_idx = (np.arange(nr)[:, None] * ns + idx[None]).reshape(-1)
    _, sg6, _ = ops.noise_level(T(y2, dev), snr_lin=snr, mult=nfft / (12.0 * 25), nv_mult=float(nfft), gather=np.int32(g_idx))
    assert rel(sg5.cpu().numpy(), sg6.cpu().numpy()) < 1e-13
    # the float32 waveform chain's filter (nrx_apply_td_paths_pow_f32, packed float32 arithmetic; opt-in fast mode, not the parity
    # path): float32 accuracy against the float64 result, noise level from its own output
    import torch
    got32 = ops.apply_td_paths(T(x, dev).to(torch.complex64), gains, T(taps, dev), offs, list(cps + nfft),
                               power=(nfft, snr, nfft / (12.0 * 25), float(nfft)))
    y7, sg7, _ = got32
    assert y7.dtype == torch.complex64 and rel(y7.cpu().numpy(), y2) < 2e-6 and rel(sg7.cpu().numpy(), sg_ref) < 1e-6
    y8 = ops.apply_td_paths(T(x, dev).to(torch.complex64), gains, T(taps, dev), offs, list(cps + nfft))
    assert torch.equal(y8, y7)
    # a wideband precoder folded into the gains (nrx_fold_precoder_f64): filtering the Nl layer signals with the folded gains
    # == filtering the Nt precoded signals (the time-domain link modulates layers, grid.py:505-516 + channelmodel.py:431-447)
    nl = 3
    F = crandn(rng, n, nt, nl)
    s_l = crandn(rng, n, nl, ns)
    s_l[..., slot_len:] = 0
    gf = ops.fold_precoder(gains, T(F, dev))
    assert rel(gf.cpu().numpy(), np.einsum('bcrtp,btl->bcrlp', ref_g, F)) < 1e-13
    y3 = ops.apply_td_paths(T(s_l, dev), gf, T(taps, dev), offs, list(cps + nfft)).cpu().numpy()
    xp = np.einsum('btl,bls->bts', F, s_l)
    y4 = ops.apply_td_paths(T(xp, dev), gains, T(taps, dev), offs, list(cps + nfft)).cpu().numpy()
    assert rel(y3, y4) < 1e-12
    gs = ops.fold_precoder(gains, T(F[0], dev))                                             # one precoder for all items
    assert rel(gs.cpu().numpy(), np.einsum('bcrtp,tl->bcrlp', ref_g, F[0])) < 1e-13


@pytest.mark.parametrize("nr,nl,nfft,K", [(4, 4, 4096, 3276), (2, 2, 1024, 612), (4, 3, 512, 300), (1, 1, 256, 144)])
def test_perfect_csi_equaliser_from_path_spectra(dev, nr, nl, nfft, K):
    """nrx_mmse_equalize_paths_f64 (Hest = exp(2 pi i k' o / N) sum_p g_p S_p[k] formed per RE from the folded path gains, no channel
    matrix in memory) against the matrix route nrx_cir -> nrx_channel_matrix (FFT of the CIR advanced by chanOffset,
    channelmodel.py:362-400) -> nrx_mmse_equalize: equalised symbols and LLR scales; the path spectra against NumPy's FFT of the
    coefficient matrix; a symbol mask leaves the other symbols' outputs alone."""
    import torch
    from neoradium_amd import ops
    rng = np.random.default_rng(nr * 100 + nl * 10 + nfft)
    n, N, L = 3, 9, 14
    fs = 30.72e6 * nfft / 1024
    g = crandn(rng, n, L + 1, nr, nl, N + 1)
    delays = np.sort(rng.uniform(0, 2500, N + 1))
    delays[0] = 0
    coeff, _ = op.coeff_matrix(delays, fs, op.build_firs())
    taps, offs = ops.path_taps(coeff)
    spec = ops.td_path_spectra_bins(T(taps, dev), offs, K, nfft)
    c = np.zeros((N + 1, nfft))
    c[:, :coeff.shape[1]] = coeff
    pick = np.append(np.arange(K // 2) + nfft - K // 2, np.arange(K // 2))
    assert rel(spec.cpu().numpy(), np.fft.fft(c, axis=1)[:, pick]) < 1e-13
    cir, off = ops.cir(T(g, dev), T(coeff, dev), L)
    assert int(off.max()) > 0                                      # (the circular advance is exercised)
    H = ops.channel_matrix(cir, off, L, K, nfft)                  # (n, L, K, nr, nl): the gains play the folded ones
    rx = crandn(rng, n, nr, L, K)
    nv = np.array([0.01, 0.2, 1e-12])
    eq0, sc0 = ops.mmse_equalize(T(rx, dev), H, T(nv, dev))
    eq1, sc1 = ops.mmse_equalize_paths(T(rx, dev), T(g, dev), spec, off, T(nv, dev), nfft)
    assert rel(eq1.cpu().numpy(), eq0.cpu().numpy()) < 1e-10 and rel(sc1.cpu().numpy(), sc0.cpu().numpy()) < 1e-10
    mask = 0b01010110011101
    eq2, sc2 = ops.mmse_equalize_paths(T(rx, dev), T(g, dev), spec, off, T(nv, dev), nfft, sym_mask=mask)
    keep = [l for l in range(L) if (mask >> l) & 1]
    assert torch.equal(eq2[:, :, keep], eq1[:, :, keep]) and torch.equal(sc2[:, :, keep], sc1[:, :, keep])
    assert ops.mmse_equalize_paths(T(crandn(rng, n, 3, L, K), dev), T(crandn(rng, n, L + 1, 3, nl, N + 1), dev), spec, off, T(nv, dev), nfft) is None


@pytest.mark.parametrize("nx,max_delay_ns,nfft,mu", [(4, 900, 1024, 1), (4, 19000, 1024, 1), (2, 5000, 512, 2), (1, 300, 256, 3), (4, 2600, 4096, 1)])
def test_overlap_save_filter_equals_the_path_form(dev, nx, max_delay_ns, nfft, mu):
    """nrx_apply_td_os_f64 (1024-point overlap-save, the Nr x Nt spectra of a gain set in registers) against the oracle's
    channelmodel.py:403-448 restatement and against the path-form kernel: waveform and noise level.  Cases: short and long paths
    (hist 43 ... 600: one to many blocks per gain set, a tail set shorter than a block), symbols shorter than a block, Nr = Nt in
    {1, 2, 4}, non-zero input right up to the end of the buffer."""
    from neoradium_amd import ops
    rng = np.random.default_rng(nx * 1000 + nfft)
    n, N = 3, 7
    fs = 30.72e6 * nfft / 1024
    cps = np.append(op.cp_lens_slot(mu, 0, nfft), op.cp_lens_slot(mu, 1, nfft)[0])
    slot_len = int((cps[:-1] + nfft).sum())
    g = crandn(rng, n, 15, nx, nx, N + 1)
    delays = np.sort(rng.uniform(0, max_delay_ns, N + 1))
    delays[0] = 0
    coeff, _ = op.coeff_matrix(delays, fs, op.build_firs())
    taps, offs = ops.path_taps(coeff)
    hist = int(offs.max()) + taps.shape[1] - 1
    ns = slot_len + coeff.shape[1] + 5
    x = crandn(rng, n, nx, ns)
    lens = list(cps + nfft)
    spec = ops.td_path_spectra(T(taps, dev), offs)
    # the path spectra: FFT_1024 of every row of the coefficient matrix / 1024, in decimation-in-frequency position order
    c = np.zeros((N + 1, 1024))
    c[:, :coeff.shape[1]] = coeff
    pos = np.array([int(format(k, '010b')[::-1], 2) for k in range(1024)])
    assert rel(spec.cpu().numpy()[:, pos], np.fft.fft(c, axis=1) / 1024) < 1e-13
    snr = np.array([3.0, 40.0, 17.0])
    y_os, sg, nv = ops.apply_td_os(T(x, dev), T(g, dev), spec, hist, lens, power=(nfft, snr, nfft / (12.0 * 25), float(nfft)))
    y_pf = ops.apply_td_paths(T(x, dev), T(g, dev), T(taps, dev), offs, lens).cpu().numpy()
    y_os = y_os.cpu().numpy()
    for b in range(n):
        ref_y = op.apply_td(x[b], g[b], coeff, cps + nfft)
        assert rel(y_os[b], ref_y) < 1e-12 and rel(y_pf[b], ref_y) < 1e-12
    assert np.abs(y_os - y_pf).max() < 1e-12 * np.abs(y_pf).max()       # sample by sample, tail beyond the slot included
    assert np.array_equal(ops.apply_td_os(T(x, dev), T(g, dev), spec, hist, lens).cpu().numpy(), y_os)
    starts = np.concatenate([[0], np.cumsum((cps + nfft)[:-1])])
    o = np.int64(np.round(cps[:-1] * 0.5))
    idx = (starts[:-1, None] + o[:, None] + np.arange(nfft)[None]).reshape(-1)
    var_ref = np.array([np.var(y_pf[b][:, idx]) for b in range(n)])
    sg_ref = np.sqrt(var_ref * nfft / (12.0 * 25) / snr)
    assert rel(sg.cpu().numpy(), sg_ref) < 1e-12 and rel(nv.cpu().numpy(), sg_ref ** 2 * nfft) < 1e-12
    # no instantiation: Nr != Nt -> None (the caller takes the path form)
    assert nx == 1 or ops.apply_td_os(T(x, dev), T(g[:, :, :1], dev), spec, hist, lens) is None


@pytest.mark.parametrize("P,l_cdm,ds,ctype", [(1, 1, [2], 1), (2, 1, [2, 11], 1), (4, 1, [2, 7, 11], 1), (4, 2, [2, 3, 10, 11], 1),
                                             (3, 1, [3, 9], 2)])
def test_chest_ls(dev, P, l_cdm, ds, ctype):
    from neoradium_amd import ops
    rng = np.random.default_rng(P * 7 + l_cdm)
    n, nr, L, nrb = 2, 2, 14, 6
    K = 12 * nrb
    base = np.arange(0, 11, 2) if ctype == 1 else np.array([0, 1, 6, 7])
    delta = [(p // 2) % 2 if ctype == 1 else 2 * ((p // 2) % 3) for p in range(P)]
    ks = np.stack([np.concatenate([12 * rb + base + delta[p] for rb in range(nrb)]) for p in range(P)])
    nk = ks.shape[1]
    sets = 3
    pil = np.exp(1j * rng.uniform(0, 2 * np.pi, (sets, P, len(ds), nk)))
    rx = crandn(rng, n, nr, L, K)
    pset = np.array([2, 0], dtype=np.int32)
    h = ops.chest_ls(T(rx, dev), T(pil, dev), ks, ds, l_cdm=l_cdm, k_cdm=2, pil_set=pset).cpu().numpy()
    for b in range(n):
        ref = op.estimate_channel_ls(rx[b], pil[pset[b]], ds, ks, l_cdm=l_cdm, k_cdm=2)
        assert rel(h[b], ref) < 1e-11


@pytest.mark.parametrize("name", ['a', 'b', 'c', 'd'])
def test_chest_polar_and_noise_vs_reference(dev, name):
    """Grid.estimateChannelLS(polarInt, kernel='linear') and its noise estimate against outputs of the reference
    (tests/golden/chest.npz), through the class surface and through the kernels on a batch (vs the oracle).
    Tolerance: 1e-10 relative on the complex estimate (atan2/hypot/sincos differ from libm in the last bits; the
    unwrap itself follows NumPy's operation order), 1e-8 relative on the noise variance (direct DFT vs pocketfft)."""
    import os
    import neoradium_amd as nr
    from neoradium_amd import ops
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'chest.npz'))
    cfg = eval(str(g[name + '_cfg']))
    rx, pil, ks, ds = g[name + '_rx'], g[name + '_pilots'], g[name + '_port_ks'], [int(v) for v in g[name + '_dmrs_syms']]
    l_cdm, k_cdm, nfft, cp_min, spacing = (int(v) for v in g[name + '_geom'])
    # ---- class surface: the same carrier / PDSCH / DMRS as the fixture's generator, the received grid as data
    car = nr.Carrier(numRbs=cfg['numRbs'], spacing=cfg['spacing'])
    p = nr.PDSCH(car.curBwp, numLayers=cfg['layers'], nID=car.cellId, modulation='16QAM')
    p.setDMRS(**cfg['dmrs'])
    pil2, ks2, ds2 = p.dmrs.getPilots()
    assert np.array_equal(ks2, ks) and list(ds2) == ds and np.abs(pil2 - pil).max() < 1e-14
    grid = car.curBwp.createGrid(rx.shape[0])
    grid.grid = rx.copy()
    for pi, polar in enumerate((False, True)):
        h, nv = grid.estimateChannelLS(p.dmrs, polarInt=polar, kernel='linear')
        ref = g[name + ('_h_pol' if polar else '_h_lin')]
        assert rel(h[::3], ref) < 1e-10, (name, polar)
        assert abs(nv - g[name + '_nv'][pi]) <= 1e-8 * g[name + '_nv'][pi], (name, polar, nv)
    # ---- kernels on a batch with per-item pilot sets, vs the oracle
    rng = np.random.default_rng(5)
    n = 3
    rxb = np.stack([rx, rx * np.exp(1j * 0.7) + 0.05 * crandn(rng, *rx.shape), 0.5 * crandn(rng, *rx.shape)])
    pils = np.stack([pil, pil * np.exp(1j * rng.uniform(0, 2 * np.pi, pil.shape))])
    pset = np.int32([0, 1, 0])
    h, hk = ops.chest_ls_ex(T(rxb, dev), T(pils, dev), ks, ds, l_cdm=l_cdm, k_cdm=k_cdm, pil_set=pset, polar=True, want_hk=True)
    raw, num = ops.chest_noise_var(T(rxb, dev), T(pils, dev), ks, ds, hk, nfft, cp_min, l_cdm=l_cdm, k_cdm=k_cdm, pil_set=pset)
    h, raw = h.cpu().numpy(), raw.cpu().numpy()
    for b in range(n):
        ref, at_p, at_s = op.estimate_channel_ls(rxb[b], pils[pset[b]], ds, ks, l_cdm=l_cdm, k_cdm=k_cdm, polar=True, parts=True)
        assert rel(h[b], ref) < 1e-10
        _, raw_ref = op.estimate_noise_var(at_p, at_s, ks, l_cdm, k_cdm, rx.shape[2], nfft, cp_min, spacing)
        assert abs(raw[b] - raw_ref) <= 1e-8 * raw_ref and num == sum(a.size for a in at_p)
    with pytest.raises(ValueError):
        grid.estimateChannelLS(p.dmrs, kernel='cubic')          # not one of the reference's kinds (tests/test_gpu_csirs.py)


@pytest.mark.parametrize("profile,nr_,nt_,prbs,first", [('C', [1, 2], [1, 2], 273, 0), ('D', [1, 1], [1, 2], 51, 3), ('A', [1, 1], [1, 1], 24, 0)])
def test_fused_channel_setup_equals_the_separate_entries(dev, profile, nr_, nt_, prbs, first):
    """nrx_chan_setup_f64 (chanOffset + first-PRB channel matrix from the path gains in one launch, no CIR in memory) against
    nrx_cir_f64 + nrx_channel_matrix_sub_f64: bit-identical offsets and matrices -- NLOS and LOS profiles, 4x4 / 2x4 / 2x2."""
    import torch
    import neoradium_amd as nr
    from neoradium_amd import ops
    from neoradium_amd._dev import D
    nr.random.setSeed(11)
    car = nr.Carrier(numRbs=prbs, spacing=30)
    bwp = car.curBwp
    ch = nr.CdlChannel(bwp, profile, delaySpread=300, carrierFreq=4e9, dopplerShift=5,
                       txAntenna=nr.AntennaPanel(nt_, polarization="x"), rxAntenna=nr.AntennaPanel(nr_, polarization="x"))
    p = nr.PDSCH(bwp, numLayers=2, nID=car.cellId, modulation='QPSK')
    p.setDMRS(configType=1, additionalPos=1)
    link = nr.PdschLink(p, ch, 0.5, numIter=2, decoder="f64")
    n = 9
    times = D(link.gain_times(np.arange(40, 40 + n)))
    gains = ops.cdl_gains(link.A, link.nu, times, A_los=link.Alos, nu_los=link.nulos)
    cir1, off = ops.cir(gains, link.coeff, link.L)
    want = ops.channel_matrix_sub(cir1, off, link.L, link.K, link.nfft, 12 * first, 12)
    got = ops.chan_setup(gains, link.coeff, link.L, link.K, link.nfft, 12 * first, 12)
    assert got is not None
    assert torch.equal(got[1], off) and torch.equal(got[0], want)
    assert ops.chan_setup(gains, link.coeff, link.L, link.K, link.nfft, 12 * first, 24) is None      # two PRBs: the separate entries
    # ... and from the paths' spectra (nrx_chan_setup_paths_f64: the sums over the taps taken out of the per-row work): the same offsets, the
    # same matrix up to the order of the sums
    spec = ops.td_path_spectra_bins(link.taps, link.tap_off, link.K, link.nfft)
    hp, op_ = ops.chan_setup_paths(gains, link.coeff, spec, link.L, link.K, link.nfft, 12 * first, 12)
    assert torch.equal(op_, off) and rel(hp.cpu().numpy(), want.cpu().numpy()) < 1e-13
    h24, o24 = ops.chan_setup_paths(gains, link.coeff, spec, link.L, link.K, link.nfft, 12 * first, 24)
    assert torch.equal(o24, off) and rel(h24.cpu().numpy(), ops.channel_matrix_sub(cir1, off, link.L, link.K, link.nfft, 12 * first, 24).cpu().numpy()) < 1e-13
