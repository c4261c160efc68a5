"""Oracle (CPU, NumPy complex128) for the modem / grid / waveform / channel stages.  TEST INFRASTRUCTURE.

See oracle/__init__.py.  Each function cites the reference lines (NeoRadium v0.4.0) it restates.  Carrier
numerology is passed explicitly (nfft, cp lengths) so the same code covers the reference's range
(nFFT = 2048 >> mu at 30.72 MHz) and the 273-PRB extension (nFFT 4096), where no reference exists.
"""
import numpy as np

# ----------------------------------------------------------------------------------------------- sequences / modem
QM = {'BPSK': 1, 'QPSK': 2, '16QAM': 4, '64QAM': 6, '256QAM': 8, '1024QAM': 10}
_NORM = {1: 2, 2: 2, 4: 10, 6: 42, 8: 170, 10: 682}


def gold(c_init, n):
    """utils.py:70-94 goldSequence: TS 38.211 5.2.1 length-31 Gold sequence c(0..n-1), Nc = 1600.
    Direct LFSR statement: x1(0)=1, x2 from c_init, both advanced Nc steps (the reference folds the advance into
    word-parallel updates; the emitted bits are the same sequence)."""
    nc = 1600
    tot = nc + n
    x1 = np.zeros(tot + 31, dtype=np.uint8)
    x2 = np.zeros(tot + 31, dtype=np.uint8)
    x1[0] = 1
    x2[:31] = [(c_init >> i) & 1 for i in range(31)]
    for i in range(tot):
        x1[i + 31] = x1[i + 3] ^ x1[i]
        x2[i + 31] = x2[i + 3] ^ x2[i + 2] ^ x2[i + 1] ^ x2[i]
    return (x1[nc:nc + n] ^ x2[nc:nc + n]).astype(np.int8)


def constellation(qm):
    """modulation.py:60-74: TS 38.211 5.1 mapping, index = MSB-first integer of the qm bits."""
    pts = np.zeros(1 << qm, dtype=np.complex128)
    for v in range(1 << qm):
        b = [(v >> (qm - 1 - i)) & 1 for i in range(qm)]
        if qm == 1:
            pts[v] = (1 - 2 * b[0]) * (1 + 1j)
            continue
        re = im = 1
        for q in range(2, qm, 2):                      # innermost amplitude bit first (modulation.py:66-68)
            re = (1 << (q // 2)) - (1 - 2 * b[qm - q]) * re
            im = (1 << (q // 2)) - (1 - 2 * b[qm + 1 - q]) * im
        pts[v] = (1 - 2 * b[0]) * re + 1j * (1 - 2 * b[1]) * im
    return pts / np.sqrt(_NORM[qm])


def modulate(bits, qm):
    """modulation.py:127-156."""
    bits = np.asarray(bits).astype(np.int64).reshape(-1, qm)
    idx = (bits << np.arange(qm - 1, -1, -1)).sum(1)
    return constellation(qm)[idx]


def demap_maxlog(sym, noise_var, qm):
    """modulation.py:159-204 getLLRsFromSymbols(useMax=True): exhaustive max-log, positive = bit 0."""
    c = constellation(qm)
    d = np.abs(np.asarray(sym).reshape(-1, 1) - c[None, :])
    e = -d ** 2 / noise_var
    bitsel = ((np.arange(1 << qm)[:, None] >> np.arange(qm - 1, -1, -1)[None, :]) & 1).astype(bool)   # (2^qm, qm)
    out = np.empty((e.shape[0], qm))
    for i in range(qm):
        out[:, i] = e[:, ~bitsel[:, i]].max(1) - e[:, bitsel[:, i]].max(1)
    return out.reshape(-1)


def demap_exact(sym, noise_var, qm):
    """modulation.py:198-201 getLLRsFromSymbols(useMax=False): log-sum-exp with the +-700 exponent clip."""
    c = constellation(qm)
    d = np.abs(np.asarray(sym).reshape(-1, 1) - c[None, :])
    e = np.exp(np.clip(-d ** 2 / noise_var, -700, 700))
    bitsel = ((np.arange(1 << qm)[:, None] >> np.arange(qm - 1, -1, -1)[None, :]) & 1).astype(bool)
    out = np.empty((e.shape[0], qm))
    for i in range(qm):
        out[:, i] = np.log(e[:, ~bitsel[:, i]].sum(1)) - np.log(e[:, bitsel[:, i]].sum(1))
    return out.reshape(-1)


def pdsch_scramble_cinit(rnti, q, nid):
    """pdsch.py:603-605."""
    return rnti * (1 << 15) + q * (1 << 14) + nid


def pdsch_llrs(eq_syms, scales, noise_var, qm, c_init, scr=None):
    """pdsch.py:935-1005 getLLRsFromGrid on already layer-demapped symbols/scales (one codeword).
    ``scr``: the scrambling bits if already generated (else derived from c_init)."""
    nv = max(noise_var, 1e-10)
    llr = demap_maxlog(eq_syms, nv, qm)
    c = gold(c_init, len(llr)) if scr is None else np.asarray(scr)[:len(llr)]
    llr = llr * (1 - 2 * c.astype(np.float64))
    if scales is not None:
        llr = llr * np.repeat(scales, qm)
    return llr


# ----------------------------------------------------------------------------------------------------- grid stages
def precode(grid, f):
    """grid.py:456-518 (wideband matrix form): (Nl,L,K) x (Nt,Nl) -> (Nt,L,K)."""
    return np.einsum('tn,nlk->tlk', f, grid)


def apply_channel_fd(grid, h):
    """grid.py:978-1018 applyChannel: h (L,K,Nr,Nt), grid (Nt,L,K) -> (Nr,L,K)."""
    return np.einsum('lkrt,tlk->rlk', h, grid)


def equalize_mmse(rx, hf, noise_var):
    """grid.py:626-694 equalize: rx (Nr,L,K), hf (L,K,Nr,P) -> (P,L,K) equalised, (P,L,K) LLR scales.
    Direct inverse of (H^H H + s2 I); the reference goes through pinv / SVD, same matrix (<=1e-11)."""
    nv = max(1e-8, noise_var)
    hh = np.conj(np.swapaxes(hf, -1, -2))
    a = hh @ hf + nv * np.eye(hf.shape[-1])
    ainv = np.linalg.inv(a)
    w = ainv @ hh                                                  # (L,K,P,Nr)
    eq = np.einsum('lkpr,rlk->plk', w, rx)
    scale = (1 / np.real(np.diagonal(ainv, 0, -2, -1)))            # (L,K,P)
    return eq, np.transpose(scale, (2, 0, 1))


def noise_std_grid(grid, snr_db):
    """grid.py:1040-1046 getNoiseStd (useRxPower=True branch of addNoise grid.py:1166)."""
    return np.sqrt(np.var(grid) / 10 ** (snr_db / 10))


# ------------------------------------------------------------------------------------------------------- OFDM
def cp_lens_slot(mu, slot_in_subframe, nfft, ext=False):
    """carrier.py:245-270 getCpLen for the 14 (12) symbols of one slot, scaled to an nfft-point grid
    (reference: nfft = 2048 >> mu, i.e. scale 1)."""
    base = 2048 >> mu
    sc = nfft // base
    nsym = 12 if ext else 14
    out = []
    for l in range(nsym):
        s = slot_in_subframe * nsym + l
        if ext:
            cp = 512 >> mu
        else:
            cp = 144 >> mu
            if s in (0, 7 * (1 << mu)):
                cp += 16
        out.append(cp * sc)
    return np.array(out, dtype=np.int64)


def window_len_std(cps):
    """waveform.py:99-109,120-122 ("STD", normal CP): half the CP, minimum over the slot."""
    return int(min((c + 1) // 2 for c in cps))


def ofdm_modulate(grid, nfft, cps, window=True):
    """grid.py:521-582 ofdmModulate (f0=0) + waveform.py:380-470 applyWindowing("STD")."""
    p, L, K = grid.shape
    pad = ((nfft - K + 1) // 2, (nfft - K) // 2)
    x = np.fft.ifft(np.fft.ifftshift(np.pad(grid, ((0, 0), (0, 0), pad)), axes=2), axis=2)
    syms = [np.concatenate([x[:, l, nfft - cps[l]:], x[:, l]], axis=1) for l in range(L)]
    wave = np.concatenate(syms, axis=1)
    if not window:
        return wave
    w = window_len_std(cps)
    rc = 0.5 * (1 - np.sin(np.pi * np.arange(w - 1, -w, -2) / (2 * w)))
    out = np.zeros_like(wave)
    start = 0
    for l in range(L):
        n = cps[l] + nfft
        s = wave[:, start:start + n]
        ex = np.concatenate([s[:, nfft - w:nfft], s], axis=1).copy()      # prefix extended by w samples
        ex[:, :w] *= rc
        ex[:, -w:] *= rc[::-1]
        if l < L - 1:
            out[:, start:start + n + w] += ex
        else:
            out[:, start:start + n] += ex[:, :n]
            out[:, :w] += ex[:, -w:]                                       # tail wraps to the slot start
        start += n
    return np.roll(out, -w, axis=1)


def ofdm_demodulate(wave, nfft, cps, K):
    """waveform.py:473-527 ofdmDemodulate (f0=0, cpOffsetRatio=0.5): FFT window starts half-way into the CP."""
    L = len(cps)
    sym = cps + nfft
    starts = np.concatenate([[0], np.cumsum(sym[:-1])])
    off = np.int32(np.round(cps * 0.5))
    idx = (cps[:, None] - off[:, None] + np.arange(nfft)) % nfft + off[:, None] + starts[:, None]
    g = np.fft.fftshift(np.fft.fft(wave[:, idx], axis=2), axes=2)
    k0 = nfft // 2 - K // 2
    return g[:, :, k0:k0 + K]


def re_power_waveform(wave, nfft, cps, n_rb):
    """waveform.py:107-117 getRePower."""
    sym = cps + nfft
    starts = np.concatenate([[0], np.cumsum(sym[:-1])])
    off = np.int32(np.round(cps * 0.5))
    idx = (cps[:, None] - off[:, None] + np.arange(nfft)) % nfft + off[:, None] + starts[:, None]
    return np.var(wave[:, idx]) / (12 * n_rb)


def noise_std_waveform(wave, nfft, cps, n_rb, snr_db):
    """waveform.py:119-142 getNoiseStd."""
    return np.sqrt(re_power_waveform(wave, nfft, cps, n_rb) * nfft / 10 ** (snr_db / 10))


# ---------------------------------------------------------------------------------------------- tapped delay line
def build_firs(filter_len=16, stop_band_atten=80, quant=64):
    """channelmodel.py:249-289 buildFirs: (quant+1) fractional-delay Kaiser-windowed sinc filters."""
    a = stop_band_atten
    beta = 0.1102 * (a - 8.7) if a > 50 else (0 if a < 21 else 0.5842 * (a - 21) ** 0.4 + 0.07886 * (a - 21))
    nn = quant * filter_len
    fir = np.kaiser(nn + 1, beta) * np.sinc(np.arange(-nn // 2, nn // 2 + 1) / quant)
    fir[0:nn + 1:quant] = 0
    fir[nn // 2] = 1
    firs = fir[:-1].reshape(filter_len, quant).T
    return np.concatenate([firs, np.roll(firs[:1], -1)])


def coeff_matrix(path_delays_ns, sample_rate, firs, filter_len=16, quant=64):
    """channelmodel.py:292-318 getCoeffMatrix -> (paths, coeffLen), filterDelay."""
    d = np.asarray(path_delays_ns) * 1e-9 * sample_rate
    di = np.int32(d)
    frac = d - di
    fdelay = int(np.clip(filter_len // 2 - 1 - di.min(), 0, None))
    di = di + fdelay
    q = np.int32(np.round(quant * (1 - frac)))
    clen = int(di.max() + filter_len // 2 + 1)
    m = np.zeros((len(d), clen))
    for p in range(len(d)):
        s = di[p] - filter_len // 2 + 1
        m[p, s:s + filter_len] = firs[q[p]]
    return m, fdelay


def sym_gain_times(cps, nfft, slot_start):
    """channelmodel.py:328-334: sample index of the start of each symbol's useful part (+ the first symbol of the
    next slot: nc+1 instants).  ``cps`` must hold nc+1 CP lengths."""
    sl = cps + nfft
    sl = sl.copy()
    sl[0] -= nfft
    return slot_start + np.cumsum(sl)


def cir_from_gains(gains, coeff):
    """channelmodel.py:343-346: gains (nc,Nr,Nt,P) x coeff (P,cl) -> cir (nc,Nr,Nt,cl), chanOffset."""
    cir = np.einsum('crtp,pl->crtl', gains, coeff)
    off = int(np.abs(cir.sum((0, 2))).sum(0).argmax())
    return cir, off


def channel_matrix(cir, off, nfft, K):
    """channelmodel.py:362-400 getChannelMatrix: (nc,Nr,Nt,cl) -> (nc,K,Nr,Nt)."""
    nc, nr, nt, cl = cir.shape
    buf = np.zeros((nc, nfft, nr * nt), dtype=np.complex128)
    idx = np.append(np.arange(-off, 0), np.arange(cl - off))
    buf[:, idx, :] = np.transpose(cir.reshape(nc, -1, cl), (0, 2, 1))
    hf = np.fft.fft(buf, axis=1)
    pick = np.append(np.arange(K // 2) + nfft - K // 2, np.arange(K // 2))
    return hf[:, pick, :].reshape(nc, K, nr, nt)


def apply_td(x, gains1, coeff, sym_lens):
    """channelmodel.py:403-448 applyToSignal: x (Nt, ns) -> (Nr, ns).  gains1 (nc+1,Nr,Nt,P); sym_lens (nc+1,) are the
    whole-symbol lengths used to pick the gain set of each OUTPUT sample; zero initial filter state."""
    nt, ns = x.shape
    P, cl = coeff.shape
    idx = np.concatenate([np.full(int(n), i) for i, n in enumerate(sym_lens)])[:ns]
    if ns > len(idx):
        idx = np.append(idx, np.full(ns - len(idx), len(sym_lens) - 1))
    y = np.zeros((gains1.shape[1], ns), dtype=np.complex128)
    for p in range(P):
        taps = np.nonzero(coeff[p])[0]
        xf = np.zeros((nt, ns), dtype=np.complex128)
        for k in taps:
            xf[:, k:] += coeff[p, k] * x[:, :ns - k]
        y += np.einsum('nrt,tn->rn', gains1[idx, :, :, p], xf)
    return y


# -------------------------------------------------------------------------------------------- LS channel estimate
def _lin_interp(x, y, xn):
    """utils.py:26-35 interpolate('linear') = scipy interp1d(kind='linear', fill_value='extrapolate') on axis 0."""
    x = np.asarray(x, dtype=np.float64)
    xn = np.asarray(xn, dtype=np.float64)
    j = np.clip(np.searchsorted(x, xn, side='left'), 1, len(x) - 1)
    x0, x1 = x[j - 1], x[j]
    sl = (y[j] - y[j - 1]) / (x1 - x0).reshape((-1,) + (1,) * (y.ndim - 1))
    return sl * (xn - x0).reshape((-1,) + (1,) * (y.ndim - 1)) + y[j - 1]


def _polar_interp(x, y, xn):
    """utils.py:38-42 polarInterpolate(..., 'linear'): unwrapped angle and magnitude interpolated separately."""
    theta, r = np.unwrap(np.angle(y), axis=0), np.abs(y)
    tn, rn = _lin_interp(x, theta, xn), _lin_interp(x, r, xn)
    return rn * (np.cos(tn) + 1j * np.sin(tn))


def estimate_channel_ls(rx, pilots, dmrs_syms, port_ks, l_cdm=1, k_cdm=2, polar=False, parts=False):
    """grid.py:874-975 estimateChannelLS(polarInt, kernel='linear') -> (L,K,Nr,P) (channel part).

    rx (Nr,L,K); pilots (P, nDmrsSym, nK) pilot values of each port at its own subcarriers port_ks[p] (nK,);
    dmrs_syms (nDmrsSym,).  CDM averaging over k_cdm adjacent pilots (x l_cdm symbols), linear inter/extrapolation
    over subcarriers (of the complex values, or with ``polar`` of unwrapped angle and magnitude, utils.py:38-42) then
    over symbols (always on the complex values, grid.py:866; repeat when a single estimate remains).
    ``parts``: also return the raw LS values at the pilots and the estimates at the DMRS time groups per port
    (hEstAtPilots / hEstAtPilotSyms of grid.py:767-806), which the noise estimate works on."""
    nr, L, K = rx.shape
    P = pilots.shape[0]
    out = np.zeros((L, K, nr, P), dtype=np.complex128)
    ls = np.asarray(dmrs_syms)
    at_pilots, at_syms = [], []
    for p in range(P):
        ks = np.asarray(port_ks[p])
        h = np.transpose(rx[:, ls][:, :, ks] / pilots[p][None], (1, 2, 0))          # (nL, nK, Nr)
        at_pilots.append(h)
        nL, nK = h.shape[:2]
        h = np.transpose(h.reshape(nL, -1, k_cdm, nr), (0, 2, 1, 3)).reshape(nL // l_cdm, l_cdm * k_cdm, -1, nr).mean(1)
        kc = ks.reshape(-1, k_cdm).mean(1)
        interp = _polar_interp if polar else _lin_interp
        hk = np.transpose(interp(kc, np.transpose(h, (1, 0, 2)), np.arange(K)), (1, 0, 2))   # (nL', K, Nr)
        at_syms.append(hk)
        if hk.shape[0] == 1:
            full = np.repeat(hk, L, axis=0)
        else:
            lc = ls.reshape(-1, l_cdm).mean(1)
            full = _lin_interp(lc, hk, np.arange(L))
        out[..., p] = full
    return (out, at_pilots, at_syms) if parts else out


# the small 8-8-4-1 ReLU network of grid.py:697-737 (scaleNoiseVar): raw pilot-residual variance -> noise variance
_NV_W1 = np.float64([[6.25861, -0.22737, -8.51406, -0.25593, 0.08617, 0.54746, -10.5016, -0.0075],
                     [0.05773, -0.08806, 0.03222, 0.65573, -1.05669, -0.00781, 0.01074, -0.02898],
                     [-11.48739, -18.84534, 9.54569, -0.02089, 9.92439, 0.07408, 11.41916, -34.07344],
                     [0.71498, 4.52607, -0.35023, 0.05907, 2.24553, 0.06049, 0.47961, 0.44182],
                     [0.84015, 0.14097, 0.20389, -0.45147, 0.12305, -0.51977, 0.37225, 0.12104],
                     [0.41917, 10.52318, 3.35156, 0.58207, -24.37617, 0.33745, -1.11957, 1.07133],
                     [-0.12522, -1.82239, 0.90271, -0.06134, 10.43859, 0.37885, 1.36096, -0.70045],
                     [0.00109, -0.00328, -0.00657, -0.16279, -0.00351, -0.28476, 0.00053, -0.00117]])
_NV_B1 = np.float64([0.60641, 0.06111, 0.24848, 0., 0.32098, 0., -0.21224, 0.007])
_NV_W2 = np.float64([[0.10102, 0.22608, 0.32803, -0.11752], [-0.01549, 0.39246, -0.30703, 0.12527],
                     [-0.02698, 0.09462, -0.31409, 0.03994], [-0.08645, -0.00781, 0.52137, 0.45963],
                     [0.07151, -0.27656, 0.23206, -0.06437], [-0.0154, 0.07408, -0.15198, -0.4007],
                     [-0.17055, -0.06038, -0.8417, 0.43372], [-3.12708, 2.03716, -3.90529, 1.21203]])
_NV_B2 = np.float64([0.54406, 0.36443, -0.21105, 0.35659])
_NV_W3 = np.float64([[0.04271], [0.07268], [0.0702], [-0.16217]])
_NV_B3 = np.float64([0.72121])


def scale_noise_var(raw, spacing_khz, n_tx, nr, K, l_cdm, k_cdm, n_var):
    """grid.py:697-737: raw variance kept when the raw SNR exceeds 20 dB, otherwise mapped by the network."""
    raw_snr_db = 10.0 * np.log10(1 / (raw * nr))
    if raw_snr_db > 20:
        return raw
    x = np.float64([raw_snr_db, spacing_khz, n_tx, nr, K, l_cdm, k_cdm, n_var])
    snr_db = (np.maximum(np.maximum(x.dot(_NV_W1) + _NV_B1, 0).dot(_NV_W2) + _NV_B2, 0).dot(_NV_W3) + _NV_B3)[0]
    return 1 / (10.0 ** (snr_db / 10.0) * nr)


def estimate_noise_var(at_pilots, at_syms, port_ks, l_cdm, k_cdm, K, nfft, cp_min, spacing_khz):
    """Noise side output of estimateChannelLsEx (grid.py:808-837): each port's frequency-interpolated estimate goes
    to the delay domain (IFFT over the K subcarriers), everything outside a raised-cosine window of (cp_min*K//nfft)
    taps at either end is dropped, back to frequency, and the variance of (raw LS at the pilots - denoised) is
    rescaled by `scale_noise_var`.  QUIRK kept (grid.py:823): the pilot subcarriers used for EVERY port are those of
    the LAST port (`portKs` survives from the previous loop), wrong when the ports span two CDM groups."""
    P = len(at_pilots)
    nr = at_pilots[0].shape[2]
    rise = cp_min * K // nfft
    rc = .5 * (1 - np.sin(np.pi * np.arange(rise - 1, -rise, -2) / (2 * rise)))
    win = np.concatenate([rc[::-1], np.float64((K - 2 * rise) * [0]), rc])
    ks = np.asarray(port_ks[P - 1])
    deltas = []
    for p in range(P):
        den = np.fft.fft(np.fft.ifft(at_syms[p], axis=1) * win[None, :, None], axis=1)
        if l_cdm > 1:
            den = np.repeat(den, l_cdm, axis=0)
        deltas.append((at_pilots[p] - den[:, ks, :]).flatten())
    deltas = np.concatenate(deltas)
    return scale_noise_var(deltas.var(), spacing_khz, P, nr, K, l_cdm, k_cdm, len(deltas)), deltas.var()
