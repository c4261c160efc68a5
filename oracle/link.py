"""Oracle (CPU, NumPy float64) for one whole PDSCH slot: the loop body of the reference's BLER harness
(Playground/PDSCH/PDSCH-BLER.ipynb cell 2), restated on top of oracle/coding.py and oracle/phy.py.
TEST INFRASTRUCTURE -- see oracle/__init__.py: used by tests, smoke() and bench.py's cpu_baseline leg only.

Configuration-only data (DMRS template grid, RE index order, scrambling sequence, pilot table, static ray
coefficients of the channel, tap matrix, CP lengths) is passed in as plain arrays in ``static``; every arithmetic
stage of the slot is done here in NumPy in the order the reference performs it.
"""
import numpy as np

from . import coding as oc
from . import phy as op


class _GainTimes:
    """Picklable slot -> (L+1,) gain instants table (the sharded CPU baseline ships `static` to worker processes)."""

    def __init__(self, table):
        self.table = table

    def __call__(self, slot):
        return self.table[int(slot)]


def static_from_tables(tb, num_iter, freq_domain, perfect, window, slots=None, gain_times=None):
    """Plain-array description of a link from the HOST tables of neoradium_amd.engine.host_tables (NumPy data computed on
    the host by the class surface's index/DMRS/TBS/channel-setup code, which tests/test_host_logic.py pins against the
    reference) plus the harness settings.  No GPU result enters: this is what tests/test_oracle_e2e.py feeds run_slot
    with on a host without a GPU.  ``gain_times``: callable slot -> (L+1,) seconds; ``slots``: tabulate it instead."""
    c = tb['cw'][0]
    if gain_times is None:
        from neoradium_amd.engine import gain_times as _gt           # host-only arithmetic on the tables
        gain_times = _GainTimes({int(s): _gt(tb, [int(s)])[0] for s in slots}) if slots is not None else \
            (lambda slot: _gt(tb, [slot])[0])
    return dict(
        templates=tb['templates'], pilots=tb['pilots'], port_ks=np.asarray(tb['port_ks']), dmrs_syms=list(tb['dmrs_syms']),
        l_cdm=tb['l_cdm'], k_cdm=tb['k_cdm'], re_index=c['re_index'], scr=c['scr'], tbs=c['tbs'], G=c['G'], nl=tb['nl'],
        qm=c['qm'], bg=c['cfg'].bg, nr=tb['nr'], nt=tb['nt'], K=tb['K'], L=tb['L'], nfft=tb['nfft'], n_rb=tb['n_rb'],
        slots_per_frame=tb['slots_per_frame'], slots_per_subframe=tb['slots_per_subframe'],
        sym_lens=[np.asarray(v) for v in tb['sym_lens']], fs=tb['fs'], A=tb['A'], nu=tb['nu'], Alos=tb['Alos'], nulos=tb['nulos'],
        coeff=tb['coeff'], max_delay=tb['max_delay'], first_prb=tb['first_prb'], num_iter=num_iter, freq_domain=freq_domain,
        perfect=perfect, window=window, gain_times=gain_times)


def static_from_link(link, slots=None):
    """:func:`static_from_tables` for a configured neoradium_amd.engine.PdschLink (its host tables, no GPU results).
    ``slots``: tabulate the gain instants of these slots, so that the result holds no reference to the link."""
    return static_from_tables(link.tables, link.numIter, link.freqDomain, link.chanEst == "Perfect", link.window != "NONE",
                              slots=slots)


def _job(args):
    st, slot, snr_db, tb, z, F = args
    return run_slot(st, slot, snr_db, tb, z, F=F)


def run_slots_parallel(jobs, n_procs):
    """bench.py's sharded CPU baseline (SURVEY 8d leg b): one job = the argument tuple of :func:`run_slot`, one
    single-threaded process per host core.  The workers are started with "spawn" and import NumPy + oracle/ only."""
    import multiprocessing as mp
    import os
    saved = {k: os.environ.get(k) for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS')}
    os.environ.update({k: '1' for k in saved})            # inherited by the children: one thread per process
    try:
        with mp.get_context('spawn').Pool(n_procs) as pool:
            return pool.map(_job, jobs, chunksize=1)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def run_slot(st, slot, snr_db, tb, z, F=None, chan_slot=None, keep=None):
    """One slot.  tb: (TBS,) bits; z: standard-normal complex array shaped like the noisy signal
    ((Nr,L,K) in frequency-domain mode, (Nr, slotLen+maxDelay) in time-domain mode).  Returns a dict with
    the per-code-block CRC verdicts, the decoded transport block and the LLRs.
    ``chan_slot``: slot whose start is the channel's time origin when it differs from the carrier's slot number (the
    reference advances the channel clock only when the channel has been applied since the last goNext,
    channelmodel.py:180-193, 326, 349: n consecutive goNext() calls leave it at slot min(n, 1)).
    ``keep``: a dict that receives every intermediate (rate-matched bits, grids, waveforms, estimate, equalised symbols) for
    stage-by-stage comparisons (tools/r6/stage_diff.py)."""
    nl, qm, K, L, nfft = st['nl'], st['qm'], st['K'], st['L'], st['nfft']
    # ---- Tx: CRC24A, segmentation, LDPC encode, rate match (ldpc.py:1167-1204)
    rm, d = oc.encode_chain(tb, st['bg'], st['G'], nl, qm)
    p = d['p']
    # ---- scramble, modulate, layer/RE map onto the DMRS-filled grid (pdsch.py:855-932)
    grid = st['templates'][slot % st['slots_per_frame']].copy()
    syms = op.modulate(rm ^ st['scr'][:len(rm)].astype(np.int8), qm)
    grid.reshape(-1)[st['re_index']] = syms
    # ---- channel state of the slot (cdl.py:641-645, channelmodel.py:321-354)
    t = st['gain_times'](slot if chan_slot is None else chan_slot)
    gains = np.einsum('rtnm,cnm->crtn', st['A'], np.exp(2j * np.pi * t[:, None, None] * st['nu'][None]))
    if st['Alos'] is not None:
        los = st['Alos'][None] * np.exp(2j * np.pi * t * st['nulos'])[:, None, None]
        gains = np.concatenate([los[..., None], gains], axis=3)
    cir, off = op.cir_from_gains(gains[:-1], st['coeff'])
    H = op.channel_matrix(cir, off, nfft, K)
    # ---- wideband SVD precoder over the first PRB (pdsch.py:1080-1165 incl. its grouping quirk)
    if F is None:
        k0 = 12 * st['first_prb']
        _, _, vh = np.linalg.svd(H[:, k0:k0 + 12].mean(axis=(0, 1)))
        F = np.conj(vh).T[:, :nl] / np.sqrt(nl)
    pg = op.precode(grid, F)
    snr = 10 ** (snr_db / 10)
    sis = slot % st['slots_per_subframe']
    if st['freq_domain']:
        rx = op.apply_channel_fd(pg, H)
        sigma = np.sqrt(np.var(rx) / snr)                                          # grid.py:1040-1046
        rxg = rx + (sigma / np.sqrt(2)) * z
        nv = sigma * sigma
    else:
        sl = st['sym_lens'][sis]
        cps = sl[:-1] - nfft
        w = op.ofdm_modulate(pg, nfft, cps, window=st['window'])
        w = np.concatenate([w, np.zeros((w.shape[0], st['max_delay']))], axis=1)   # Waveform.pad
        cir1, _ = op.cir_from_gains(gains, st['coeff'])
        y = _apply_cir(w, cir1, sl)
        sigma = op.noise_std_waveform(y, nfft, cps, st['n_rb'], snr_db)            # waveform.py:119-142
        if keep is not None:
            keep.update(tx=w, ry=y)
        y = y + (sigma / np.sqrt(2)) * z
        rxg = op.ofdm_demodulate(y[:, off:], nfft, cps, K)                         # sync + demodulate
        nv = sigma * sigma * nfft                                                  # waveform.py:523
    # ---- Rx: channel estimate, MMSE, demap, rate recovery, decode, CRC
    if st['perfect']:
        hest = H @ F[None, None]
    else:
        hest = op.estimate_channel_ls(rxg, st['pilots'][slot % st['slots_per_frame']], st['dmrs_syms'], st['port_ks'],
                                      l_cdm=st['l_cdm'], k_cdm=st['k_cdm'])
    eq, sc = op.equalize_mmse(rxg, hest, nv)
    llr = op.pdsch_llrs(eq.reshape(-1)[st['re_index']], sc.reshape(-1)[st['re_index']], nv, qm, None, scr=st['scr'])
    rr, _ = oc.rate_recover(llr, p, nl, qm)
    dec = oc.decode(rr, st['bg'], p.iLS, p.Zc, st['num_iter'])
    out, crc = oc.crc_check_and_merge(dec, p)
    if keep is not None:
        keep.update(bits=rm, grid=grid, pgrid=pg, sigma=sigma, rxg=rxg, hest=hest, eq=eq, sc=sc, nv=nv, llr=llr, dec=dec)
    return dict(crc=crc, tb_out=out, llr=llr, F=F, off=off, nv=nv, p=p)


def _apply_cir(x, cir1, sym_lens):
    """channelmodel.py:403-448 applyToSignal in its per-(rx,tx) FIR form (identical to the reference's
    per-path lfilter + gain mix to 1e-15): y[r,n] = sum_t sum_l cir1[sym(n),r,t,l] x[t,n-l]."""
    nt, ns = x.shape
    nsets, nr, _, cl = cir1.shape
    idx = np.concatenate([np.full(int(n), i) for i, n in enumerate(sym_lens)])[:ns]
    if ns > len(idx):
        idx = np.append(idx, np.full(ns - len(idx), nsets - 1))
    y = np.zeros((nr, ns), dtype=np.complex128)
    bounds = np.concatenate([[0], np.nonzero(np.diff(idx))[0] + 1, [ns]])
    for a, b in zip(bounds[:-1], bounds[1:]):
        c = cir1[idx[a]]
        lo = max(0, a - cl + 1)
        seg = x[:, lo:b]
        for r in range(nr):
            acc = np.zeros(b - a, dtype=np.complex128)
            for t_ in range(nt):
                acc += np.convolve(seg[t_], c[r, t_])[a - lo:a - lo + (b - a)]
            y[r, a:b] = acc
    return y
