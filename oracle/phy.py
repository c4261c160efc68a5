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


def estimate_channel_ls(rx, pilots, dmrs_syms, port_ks, l_cdm=1, k_cdm=2):
    """grid.py:874-975 estimateChannelLS(polarInt=False, kernel='linear') -> (L,K,Nr,P) (channel part only).

    rx (Nr,L,K); pilots (P, nDmrsSym, nK) pilot values of each port at its own subcarriers port_ks[p] (nK,);
    dmrs_syms (nDmrsSym,).  CDM averaging over k_cdm adjacent pilots (x l_cdm symbols), linear inter/extrapolation
    over subcarriers then over symbols (repeat when a single estimate remains)."""
    nr, L, K = rx.shape
    P = pilots.shape[0]
    out = np.zeros((L, K, nr, P), dtype=np.complex128)
    ls = np.asarray(dmrs_syms)
    for p in range(P):
        ks = np.asarray(port_ks[p])
        h = np.transpose(rx[:, ls][:, :, ks] / pilots[p][None], (1, 2, 0))          # (nL, nK, Nr)
        nL, nK = h.shape[:2]
        h = np.transpose(h.reshape(nL, -1, k_cdm, nr), (0, 2, 1, 3)).reshape(nL // l_cdm, l_cdm * k_cdm, -1, nr).mean(1)
        kc = ks.reshape(-1, k_cdm).mean(1)
        hk = np.transpose(_lin_interp(kc, np.transpose(h, (1, 0, 2)), np.arange(K)), (1, 0, 2))   # (nL', K, Nr)
        if hk.shape[0] == 1:
            full = np.repeat(hk, L, axis=0)
        else:
            lc = ls.reshape(-1, l_cdm).mean(1)
            full = _lin_interp(lc, hk, np.arange(L))
        out[..., p] = full
    return out
