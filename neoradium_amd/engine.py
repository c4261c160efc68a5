"""Batched PDSCH Monte-Carlo link engine: the loop body of the reference's BLER harness
(Playground/PDSCH/PDSCH-BLER.ipynb cell 2) for a whole batch of slots resident on one GPU.

One call of :meth:`PdschLink.run` = `n_slots` independent PDSCH slots through

    random TB -> CRC24A/segmentation/CB-CRC -> LDPC encode -> rate match -> scramble -> QAM -> layer/RE map (+DMRS)
    -> SVD precoder -> [ OFDM mod (+window) -> tapped-delay-line channel -> AWGN -> timing sync -> OFDM demod ]
                     | [ frequency-domain channel -> AWGN ]
    -> DMRS LS channel estimate (or perfect CSI) -> MMSE equalise -> max-log demap + descramble
    -> rate recovery -> layered min-sum LDPC decode -> CB/TB CRC -> error counters

every stage being one libnrx kernel launch over the whole batch (no host round trips, no per-slot Python).
Configuration comes from the same host objects the notebooks build (PDSCH + DMRS, CdlChannel/TdlChannel);
everything that depends only on (configuration, slot number in frame) is precomputed once here.

Slots are independent given their absolute slot index (channel time, DMRS/scrambling by slotNoInFrame), so a
sweep shards over GPUs by slot range with no data-path communication; only the 4 error counters are reduced.
"""
import os
import numpy as np
import torch

from . import _lib, ops
from ._dev import D, device as _device
from .waveform import Waveform


def host_tables(pdsch, channel, codeRate, baseGraphNo=1):
    """Everything of a link that depends only on (configuration, slot number in frame), computed ON THE HOST with the
    class surface's own index/DMRS/TBS/channel-setup code (NumPy only, no device call): the DMRS-filled grid template and
    pilot table of every slot of the frame, the layer-mapped RE index, scrambling sequence and LDPC configuration of
    every codeword, the channel's static ray coefficients and tap matrix, the slot geometry.  PdschLink uploads these;
    the CPU oracle harness (oracle/link.py, tests/test_oracle_e2e.py) consumes the same dictionary."""
    bwp = pdsch.bwp
    car = bwp.carrier
    dmrs = pdsch.dmrs
    L, K = bwp.symbolsPerSlot, 12 * bwp.numRbs
    saved = car.slotNo
    templ, pil = [], []
    idx0 = None
    for s in range(bwp.slotsPerFrame):
        car.slotNo = s
        g = pdsch.getGrid()
        templ.append(g.grid.copy())
        p, ks, ds = dmrs.getPilots()
        pil.append(p)
        if idx0 is None:
            idx0 = tuple(i.copy() for i in pdsch.dataIndices)
            tbs_all = [int(v) for v in pdsch.getTxBlockSize(codeRate)]
            port_ks, dmrs_syms = ks, [int(v) for v in ds]
        elif not all(np.array_equal(a, b) for a, b in zip(idx0, pdsch.dataIndices)):
            raise ValueError("PdschLink: the data RE indices must be the same in every slot of the frame")
    car.slotNo = saved
    if not (0 <= int(np.min(port_ks)) and int(np.max(port_ks)) < K):
        raise ValueError("PdschLink: DMRS subcarrier outside the bandwidth part")
    nl = pdsch.numLayers
    n_res = pdsch.getNumREsFromIndexes(idx0)
    lms = pdsch.getLayerMapIndexes(idx0, n_res)
    cw_layers = [nl] if pdsch.numCW == 1 else [nl // 2, nl - nl // 2]
    cws = []
    for q in range(pdsch.numCW):
        qm = pdsch.modems[q].qm
        G = n_res[q] * qm
        lm = lms[q]
        ccfg = _lib.ldpc_config(baseGraphNo, tbs_all[q] + 24)
        e_max = max(_lib.ldpc_cb_lens(G, ccfg.C, cw_layers[q], qm))
        cws.append(dict(tbs=tbs_all[q], qm=qm, nl=cw_layers[q], G=G, cfg=ccfg, e_max=e_max, lm=lm,
                        re_index=np.int32((np.int64(lm[0]) * L + lm[1]) * K + lm[2]), scr=pdsch._scrambling(q, G)))
    # A statistical model that redraws its ray coefficients for every slot (TDL sosType='Xiao', tdl.py:1043-1067): `static_at`
    # gives the coefficients of the s-th slot after the channel's last restart() -- the draws the slot-by-slot class surface
    # makes, reproduced from a copy of the generator -- and A / nu hold slot 0's (shapes; the oracle harness runs slot 0 only).
    static_at = None
    if getattr(channel, '_static_per_slot', False):
        def static_at(s, _ch=channel):
            a, v, _, _ = _ch.staticCoefficientsAt(s)
            return np.complex128(a * _ch._normalisation()), np.float64(v)
        A, nu, Alos, nulos = channel.staticCoefficientsAt(0)
    else:
        A, nu, Alos, nulos = channel.staticCoefficients()
    sc = channel._normalisation()
    spsf = bwp.slotsPerSubFrame
    sym_lens = [bwp.symbolLens[s * L:s * L + L + 1].astype(np.int64) for s in range(spsf)]
    return dict(templates=np.stack(templ), pilots=np.stack(pil), port_ks=port_ks, dmrs_syms=dmrs_syms, idx0=idx0, lms=lms,
                n_res=n_res, cw=cws, l_cdm=dmrs.symbols, k_cdm=(4 if dmrs.enhanced else 2), first_prb=int(pdsch.prbSet[0]),
                A=np.complex128(A * sc), nu=np.float64(nu), Alos=None if Alos is None else np.complex128(Alos * sc),
                nulos=float(nulos), coeff=channel.getCoeffMatrix(), max_delay=channel.getMaxDelay(), fs=bwp.sampleRate,
                sym_lens=sym_lens, nr=channel.nrNt[0], nt=channel.nrNt[1], nl=nl, K=K, L=L, nfft=bwp.nFFT, n_rb=bwp.numRbs,
                slots_per_frame=bwp.slotsPerFrame, slots_per_subframe=spsf, static_at=static_at)


def gain_times(tables, slots):
    """(n, L+1) seconds: starts of the useful part of each symbol (+ first symbol of the next slot) of the absolute
    slots ``slots`` on the channel's time axis (channelmodel.py:173-193, 328-334)."""
    spsf, nfft = tables['slots_per_subframe'], tables['nfft']
    slot_len = [int(v[:-1].sum()) for v in tables['sym_lens']]
    sub = int(sum(slot_len))
    tab = np.empty((spsf, tables['L'] + 1), dtype=np.int64)       # sample offsets inside the subframe, per slot-in-subframe
    for r in range(spsf):
        sl = np.int64(tables['sym_lens'][r]).copy()
        sl[0] -= nfft
        tab[r] = int(sum(slot_len[:r])) + np.cumsum(sl)
    n = np.asarray(slots, dtype=np.int64)
    return (((n // spsf) * sub)[:, None] + tab[n % spsf]) / tables['fs']


class PdschLink:
    def __init__(self, pdsch, channel, codeRate, baseGraphNo=1, numIter=20, freqDomain=False, chanEst="LS",
                 decoder="f64", windowing="STD", dev=None, firstPassIter=None, polarInt=False, useMax=True,
                 skipPuncturedRows=True, waveform="f64", certifiedExit=None, certFlags=0, certSweeps=4, certInKernel=True, certPersistent=True):
        if waveform not in ("f32", "f64"):
            raise ValueError("waveform must be 'f64' (the reference's complex128 waveforms, default) or 'f32' (time-domain link only: "
                             "Tx grid, OFDM, channel filter and received grid in complex64 -- not the parity path)")
        if waveform == "f32" and (freqDomain or pdsch.numCW != 1):
            raise ValueError("waveform='f32' is built for the one-codeword time-domain link")
        self.waveform = waveform
        if pdsch.dmrs is None:
            raise ValueError("PdschLink: the PDSCH needs a DMRS configuration (pdsch.setDMRS)")
        if chanEst not in ("LS", "Perfect"):
            raise ValueError("chanEst must be 'LS' or 'Perfect'")
        if decoder not in ("f32", "f64"):
            raise ValueError("decoder must be 'f64' (the reference's arithmetic, default) or 'f32' (float32 LLRs and decoder: faster, "
                             "CRC verdicts differ from the float64 chain on about 1 block in 1e3 at the waterfall)")
        self.dev = _device() if dev is None else dev
        self.pdsch, self.channel = pdsch, channel
        self.bwp = bwp = pdsch.bwp
        self.carrier = bwp.carrier
        self.freqDomain, self.chanEst, self.decoder = freqDomain, chanEst, decoder
        self.numIter = int(numIter)
        self.polarInt = bool(polarInt)        # estimateChannelLS(polarInt=True, kernel='linear') of PDSCH-endToEnd.ipynb
        self.useMax = bool(useMax)            # getLLRsFromGrid(useMax=...): max-log (default) or exact log-sum-exp LLRs
        # Opt-in multi-pass decoding (NOT the reference's schedule, off by default): `firstPassIter` = n or an ascending
        # list (n1, n2, ...).  Every code block is decoded with n1 iterations; the blocks whose CRC fails go on to n2, ...,
        # `numIter` iterations in total, so a block that never passes gets exactly the reference's result and a passing
        # block is assumed to be the code word the full run ends on as well.  Where the fused float64 entry exists the
        # failing blocks CONTINUE from their parked decoder state (no iteration is done twice) and the list of failing
        # blocks stays on the device; elsewhere they are decoded again from scratch after one host read per batch.
        # Opt-in CERTIFIED early exit (off by default; the reference has no early stop, ldpc.py:1545): `certifiedExit` = an ascending
        # list of iteration counts.  At each of them a block stops only if its CRC passes AND the stability certificate holds on
        # its frozen decoder state (ops.ldpc_recover_decode_merge_certified): its bits are then provably those of the
        # full `numIter` run.  Needs the fused float64 entry (else ValueError at the first batch).  `certFlags` != 0 breaks the
        # certificate on purpose (tests).
        if certifiedExit is not None:
            marks = sorted({int(v) for v in (certifiedExit if isinstance(certifiedExit, (list, tuple)) else [certifiedExit])})
            if firstPassIter is not None or decoder != "f64" or not marks or marks[0] < 1 or marks[-1] >= int(numIter):
                raise ValueError("certifiedExit: ascending iteration counts in [1, numIter-1], float64 decoder, without firstPassIter")
            self.certStages = tuple(marks)
        else:
            self.certStages = None
        self.certFlags = int(certFlags)
        self.certSweeps = int(certSweeps)      # relaxation sweeps the certificate may take before it refuses
        self.certInKernel = bool(certInKernel) # the certificate in the stage kernel's tail (default) or as its own launch on the parked states
        # the whole certified schedule as ONE launch with independent code-block slots (round 6; the default where the certificate sits in
        # the kernel and there are at most three checks): False = the staged launches (stage -> select -> stage ... -> resume)
        self.certPersistent = bool(certPersistent) and self.certInKernel and (certifiedExit is None or len(tuple(np.atleast_1d(certifiedExit))) <= 3)
        self.last_exit_iter = None          # (n_cb,) uint8 of the last batch that ran the certified schedule
        if firstPassIter is None:
            self.firstPassIter, self.passStages = None, ()
        else:
            marks = [int(v) for v in (firstPassIter if isinstance(firstPassIter, (list, tuple)) else [firstPassIter])]
            if not marks or marks != sorted(set(marks)) or not 0 < marks[0] or not marks[-1] < self.numIter:
                raise ValueError("firstPassIter must be between 1 and numIter-1 (an ascending list for several checks)")
            self.firstPassIter, self.passStages = marks[0], tuple(marks[1:])
        self.codeRate = codeRate
        self.nl = pdsch.numLayers
        self.numCW = pdsch.numCW
        self.qm = pdsch.modems[0].qm
        self.nr, self.nt = channel.nrNt
        self.K, self.L, self.nfft = 12 * bwp.numRbs, bwp.symbolsPerSlot, bwp.nFFT
        dmrs = pdsch.dmrs

        # ---- host tables (DMRS-filled grid template + pilot table per slotNoInFrame, per-codeword LDPC configuration, layer-
        # mapped RE index and scrambling, channel static coefficients): computed on the host, uploaded here
        self.tables = tb_ = host_tables(pdsch, channel, codeRate, baseGraphNo)
        idx0, lms = tb_['idx0'], tb_['lms']
        self.tbs = tb_['cw'][0]['tbs']
        self.port_ks, self.dmrs_syms = tb_['port_ks'], tb_['dmrs_syms']
        self.templates = D(tb_['templates'])                   # (S, Nl, L, K) complex128
        self.templates32 = None                                # complex64 copy for waveform="f32", made on first use
        self.pilots = D(tb_['pilots'])                         # (S, P, nDs, nK)
        self.l_cdm, self.k_cdm = tb_['l_cdm'], tb_['k_cdm']
        self.port_ks_d = D(np.ascontiguousarray(np.int32(self.port_ks)))
        self.n_tg = len(self.dmrs_syms) // self.l_cdm                   # DMRS time groups
        # ---- per codeword (TS 38.211 7.3.1.3: one codeword up to 4 layers, two above: floor(v/2) + the rest): transport
        # block size, LDPC configuration, modulation order, layers, coded bits, layer-mapped RE index, scrambling
        self.cw = []
        for c in tb_['cw']:
            # rows of the base graph whose extension parity is actually transmitted (first transmission, rv 0): the
            # others are exact no-ops for the information bits and are not run (ops.ldpc_active_rows)
            self.cw.append(dict(tbs=c['tbs'], qm=c['qm'], nl=c['nl'], G=c['G'], cfg=c['cfg'],
                                rows=ops.ldpc_active_rows(c['cfg'], c['e_max']) if skipPuncturedRows else None,
                                re_index=D(c['re_index']), scr=D(c['scr'])))
        if self.numCW == 1:     # inverse RE map (grid element -> symbol number, -1 = no data): getGrid + populateGrid in one pass
            ri = np.int64(lms[0][0]) * self.L * self.K + np.int64(lms[0][1]) * self.K + np.int64(lms[0][2])
            inv = np.full(self.nl * self.L * self.K, -1, dtype=np.int32)
            if len(np.unique(ri)) != len(ri) or int(ri.min()) < 0 or int(ri.max()) >= len(inv) or len(ri) * self.cw[0]['qm'] != self.cw[0]['G']:
                raise ValueError("PdschLink: the layer/RE map must address distinct grid elements, one per modulated symbol "
                                 "(layers with different numbers of data REs -- e.g. PTRS on a subset of the ports -- make the "
                                 "reference's layer mapping, pdsch.py:619-639, write some REs twice)")
            inv[ri] = np.arange(len(ri), dtype=np.int32)
            self.re_inv = D(inv)
            self.re_planes = None         # ops.layer_planes(self.re_inv): asked once, on first use (a device reduction)
        # OFDM symbols that hold at least one data RE of any codeword: the only ones the throughput path equalises
        self.data_sym_mask = 0
        for lm in lms:
            for l in np.unique(np.asarray(lm[1])):
                self.data_sym_mask |= 1 << int(l)
        c0 = self.cw[0]     # (single-codeword attribute names kept: bench.py, the oracle harness and the tests use them)
        self.G, self.re_index, self.scr, self.cfg = c0['G'], c0['re_index'], c0['scr'], c0['cfg']
        self.first_prb = int(pdsch.prbSet[0])
        # ---- precoder groups exactly as PDSCH.getPrecodingMatrix forms them (pdsch.py:1142-1163: a group is closed when
        # the FIRST PRB of the next group arrives and the last one is never closed).  One group covering a full-band
        # wideband allocation = the single-matrix path; otherwise the per-PRG path (prgSize 2/4, or a partial
        # allocation whose "wideband" precoder the reference applies to its first PRB only).
        groups, cur, rbs = [], -1, []
        for prb in pdsch.prbSet:
            grp = 0 if pdsch.prgSize == 0 else (int(prb) + bwp.startRb) // pdsch.prgSize
            rbs.append(int(prb))
            if grp != cur:
                groups.append(rbs)
                cur, rbs = grp, []
        self.prg = not (len(pdsch.prbSet) == bwp.numRbs and pdsch.prgSize == 0)
        if self.prg:
            k2g = np.full(self.K, -1, dtype=np.int32)
            k0, nk = [], []
            for gi, g_rbs in enumerate(groups):
                if g_rbs != list(range(g_rbs[0], g_rbs[0] + len(g_rbs))):
                    raise NotImplementedError("PdschLink: a precoding group must be a run of consecutive PRBs")
                k0.append(12 * g_rbs[0])
                nk.append(12 * len(g_rbs))
                k2g[12 * g_rbs[0]:12 * (g_rbs[0] + len(g_rbs))] = gi
            self.prg_k0, self.prg_nk, self.prg_k2g = D(np.int32(k0)), D(np.int32(nk)), D(k2g)
            self.prg_groups = groups

        # ---- channel: static ray coefficients + tap matrix on the device
        self.A, self.nu = D(tb_['A']), D(tb_['nu'])
        self.Alos, self.nulos = (None if tb_['Alos'] is None else D(tb_['Alos'])), tb_['nulos']
        self.static_at = tb_.get('static_at')
        coeff = tb_['coeff']
        self.coeff = D(coeff)
        taps, offs = ops.path_taps(coeff, channel.filterLen)
        self.taps, self.tap_off = D(taps), D(offs)
        self.td_hist = int(np.max(offs)) + int(np.asarray(taps).shape[1]) - 1
        # overlap-save form of the channel filter (nrx_apply_td_os_f64): the path spectra are a constant of the channel.
        # NRX_TD_PATHS=1 keeps the path-form kernel (what the reference literally does, channelmodel.py:431-447).
        self.td_spec = None
        if not freqDomain and waveform == "f64" and not bool(int(os.environ.get('NRX_TD_PATHS', '0'))):
            self.td_spec = ops.td_path_spectra(self.taps, self.tap_off)
        # perfect CSI on the time-domain link with a wideband precoder: the equaliser forms Hest = channelMatrix @ precoder from the
        # folded path gains and the paths' spectra at the K subcarriers (ops.mmse_equalize_paths): no channel matrix is computed at all.
        # NRX_PERFECT_MATRIX=1 keeps the matrix route (getChannelMatrix -> H @ F -> equalize, what details=True always takes).
        self.bin_spec = None
        if not freqDomain and chanEst == "Perfect" and self.nr in (1, 2, 4) and self.nl <= 4 and self.nfft <= 8192 and \
                not bool(int(os.environ.get('NRX_PERFECT_MATRIX', '0'))):      # (nfft: the spectra kernel's twiddle table, FFT_TW_N)
            self.bin_spec = ops.td_path_spectra_bins(self.taps, self.tap_off, self.K, self.nfft)
        # chanOffset + the first-PRB channel matrix (for the wideband precoder) from the same spectra instead of a DFT over the CIR's
        # taps (ops.chan_setup_paths); NRX_CHAN_SETUP_DFT=1 keeps nrx_chan_setup_f64 (bit-identical to cir + channel_matrix_sub)
        self.setup_spec = None
        if not freqDomain and self.nfft <= 8192 and not bool(int(os.environ.get('NRX_CHAN_SETUP_DFT', '0'))):
            self.setup_spec = self.bin_spec if self.bin_spec is not None else ops.td_path_spectra_bins(self.taps, self.tap_off, self.K, self.nfft)
        self.max_delay = tb_['max_delay']
        self.fs = bwp.sampleRate
        self.window = windowing

        # ---- slot geometry by slot number in subframe
        self.sym_lens = tb_['sym_lens']
        self.slot_len = [int(v[:-1].sum()) for v in self.sym_lens]
        self.subframe_len = int(sum(self.slot_len))
        self._gather = {}
        self._enc_buf = {}        # codeword -> ((batch size, rows), coded-bit buffer) reused from batch to batch (ops.ldpc_encode _reuse=)
        # run_harq: the decoder launches of a round's new blocks and of its retransmissions on two streams (NRX_HARQ_ONE_STREAM=1: one after
        # the other on the caller's stream; read at construction)
        self.harq_two_streams = not bool(int(os.environ.get('NRX_HARQ_ONE_STREAM', '0')))
        self._side_stream = None
        self._sep_rr = bool(int(os.environ.get('NRX_SEPARATE_RATE_RECOVERY', '0')))    # developer switch: demap, then rate recovery
        # developer switch: the reference's operation order for the wideband precoder (precode the grid, then modulate the ports and
        # filter with the plain gains) instead of folding it into the filter's gains (same arithmetic up to reassociation of the precoder)
        self._sep_prec = bool(int(os.environ.get('NRX_SEPARATE_PRECODER', '0')))
        self._poison = bool(int(os.environ.get('NRX_DEBUG_POISON', '0')))             # test hook: the rate-recovering demapper's buffer starts as NaN
        self._sep_power = bool(int(os.environ.get('NRX_SEPARATE_POWER', '0')))     # developer switch: noise level in its own pass
        # gain instants on the device (no host -> device copy per batch: such a copy from pageable memory waits for the
        # stream to drain, i.e. for the previous batch's decoder, and the GPU then idles while the host prepares the next one)
        spsf = bwp.slotsPerSubFrame
        tab = np.empty((spsf, self.L + 1), dtype=np.int64)
        for r in range(spsf):
            sl = np.int64(self.sym_lens[r]).copy()
            sl[0] -= self.nfft
            tab[r] = int(sum(self.slot_len[:r])) + np.cumsum(sl)
        self._gt_tab = D(tab)
        self._one_geometry = all(tuple(v) == tuple(self.sym_lens[0]) for v in self.sym_lens)

    def gain_times_dev(self, slots_dev):
        """gain_times() for a device tensor of absolute slot numbers, computed on the device with the same integer and
        floating-point operations (sample index as int64 -> float64, divided by the sample rate)."""
        spsf = self.bwp.slotsPerSubFrame
        q = torch.div(slots_dev, spsf, rounding_mode='floor')
        samples = (q * self.subframe_len)[:, None] + self._gt_tab.index_select(0, slots_dev - q * spsf)
        return samples.to(torch.float64) / float(self.fs)

    # ------------------------------------------------------------------------------------------- geometry
    def slot_start(self, n):
        """First sample of absolute slot n on the channel's time axis (channelmodel.py:173-193 bookkeeping)."""
        spsf = self.bwp.slotsPerSubFrame
        return (n // spsf) * self.subframe_len + int(sum(self.slot_len[:n % spsf]))

    def gain_times(self, slots):
        """(n, L+1) seconds: starts of the useful part of each symbol (+ first symbol of the next slot)."""
        return gain_times(self.tables, slots)

    def _cp_gather(self, sis, width):
        """Element offsets of the CP-stripped samples of all Nr rows (Waveform.getRePower, waveform.py:107-117)."""
        key = (sis, width)
        if key not in self._gather:
            cps = (self.sym_lens[sis][:-1] - self.nfft)
            sym = cps + self.nfft
            starts = np.concatenate([[0], np.cumsum(sym[:-1])])
            off = np.int64(np.round(cps * 0.5))
            idx = ((cps[:, None] - off[:, None] + np.arange(self.nfft)) % self.nfft + off[:, None] + starts[:, None]).reshape(-1)
            g = (np.arange(self.nr)[:, None] * width + idx[None, :]).reshape(-1)
            self._gather[key] = D(np.int32(g))
        return self._gather[key]

    # ------------------------------------------------------------------------------------------------ run
    def run(self, slot0, n_slots, snr_db, seed=0, tb_bits=None, noise=None, counters=None, details=False, precoder=None):
        """Simulate absolute slots [slot0, slot0+n_slots).  Returns the int64[4] device counters
        (blockErrors, totalBlocks, bitErrors, totalBits) -- accumulated into ``counters`` when given.
        ``details=True`` also returns every intermediate of each geometry group, ``details="verdicts"`` only the
        per-code-block CRC verdicts.

        Throughput mode (default): transport blocks and noise come from the counter-based device generator keyed by
        (seed, slot index), so results do not depend on batch size or on how slots are sharded over GPUs.
        Parity mode: pass ``tb_bits`` (n_slots, TBS) -- a list of two such tensors for a two-codeword PDSCH -- and
        ``noise`` (standard-normal complex pairs, shape of the noisy signal) to reproduce a host NumPy PCG64 stream; ``precoder``
        (n_slots, Nt, Nl) complex: the wideband precoding matrices to use instead of the device SVD (pdsch.py:1125-1131 leaves each
        singular vector's phase to LAPACK: replaying a reference run slot for slot takes its precoders as data)."""
        dev = self.dev
        if not (details is False or details is True or details is None or (isinstance(details, str) and details == "verdicts")):
            raise ValueError('details must be False, True or "verdicts"')
        details = details or False
        if counters is None:
            counters = torch.zeros(4, dtype=torch.int64, device=dev)
        slots = np.arange(slot0, slot0 + n_slots)
        spsf = self.bwp.slotsPerSubFrame
        # one sub-batch per slot geometry (identical for mu <= 1)
        geoms = {}
        if self._one_geometry:
            geoms[0] = np.arange(n_slots)
        else:
            for i, n in enumerate(slots):
                geoms.setdefault(tuple(self.sym_lens[n % spsf]), []).append(i)
        det = []
        for _, sel in geoms.items():
            sel = np.asarray(sel)
            if tb_bits is None:
                tbs_sel = None
            elif isinstance(tb_bits, (list, tuple)):            # two codewords: one (n_slots, TBS_q) tensor per codeword
                tbs_sel = [t[sel] for t in tb_bits]
            else:
                tbs_sel = tb_bits[sel]
            d = self._run_group(slots[sel], snr_db, seed, tbs_sel, None if noise is None else noise[sel], counters, details,
                                precoder=None if precoder is None else precoder[sel])
            if details:
                det.append((sel, d))
        return (counters, det) if details else counters

    def _harq_decode(self, rr, ccfg, rows_new, new_flags):
        """Hard bits of one HARQ round's code blocks: processes that start a new transport block decode with the rows rv 0
        reaches (exact no-ops beyond), retransmissions with all rows.  Where the kernels take a work list (float64, Zc 384)
        the two groups are listed on the device and each is one launch over the whole batch -- no host read, no gathered
        copies of the LLRs; elsewhere one small host read (n_proc flags) splits the batch."""
        C = ccfg.C
        if rr.dtype == torch.float64:
            dec = torch.empty((rr.shape[0], ccfg.K), dtype=torch.uint8, device=rr.device)
            is_new = new_flags.to(torch.uint8).reshape(-1, 1).expand(-1, C).reshape(-1).contiguous()
            s_re, n_re = ops.select_zero(is_new)                              # retransmissions
            s_nw, n_nw = ops.select_zero(1 - is_new)                          # new blocks
            # the two launches side by side on two streams: each is a few rounds of whole-CU workgroups (2 304 code blocks = 4.5 rounds of
            # 256), so one's last round leaves CUs to the other; they read rr and write disjoint rows of dec
            if self.harq_two_streams and rr.is_cuda:
                main = torch.cuda.current_stream()
                if self._side_stream is None:
                    self._side_stream = torch.cuda.Stream(device=rr.device)
                side = self._side_stream
                ready, done = torch.cuda.Event(), torch.cuda.Event()
                ready.record(main)
                ok_nw = ok_re = False
                try:        # whatever either launch raises, the caller's stream waits for the side stream before rr / dec can be reused
                    with torch.cuda.stream(side):
                        side.wait_event(ready)
                        try:
                            ok_nw = ops.ldpc_decode_selected(rr, ccfg, self.numIter, rows_new, s_nw, n_nw, dec)
                        finally:
                            done.record(side)
                    ok_re = ops.ldpc_decode_selected(rr, ccfg, self.numIter, 46 if ccfg.bg == 1 else 42, s_re, n_re, dec)
                finally:
                    main.wait_event(done)
                if ok_re and ok_nw:
                    return dec
            elif ops.ldpc_decode_selected(rr, ccfg, self.numIter, 46 if ccfg.bg == 1 else 42, s_re, n_re, dec) and \
                    ops.ldpc_decode_selected(rr, ccfg, self.numIter, rows_new, s_nw, n_nw, dec):
                return dec
        fresh = new_flags.to(torch.bool).cpu()
        cbs = torch.arange(rr.shape[0], device=rr.device).reshape(-1, C)
        dec = None
        for idx, rws in ((cbs[fresh].reshape(-1), rows_new), (cbs[~fresh].reshape(-1), None)):
            if idx.numel() == 0:
                continue
            part = ops.ldpc_decode(rr if idx.numel() == rr.shape[0] else rr.index_select(0, idx), ccfg, self.numIter, rows=rws)
            if idx.numel() == rr.shape[0]:
                return part
            dec = torch.empty((rr.shape[0],) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device) if dec is None else dec
            dec.index_copy_(0, idx, part)
        return dec

    def _channel_chain(self, slots, n, slots_dev, precoder=None, no_matrix=False):
        """Path gains, timing offset, channel matrix (where the link needs it), precoder(s) and -- time-domain link with a
        wideband precoder -- the gains with the precoder folded in, of the slots of one batch."""
        times = self.gain_times_dev(slots_dev)
        if self.static_at is None:
            gains1 = ops.cdl_gains(self.A, self.nu, times, A_los=self.Alos, nu_los=self.nulos)
        else:                       # per-slot ray coefficients (host draws in the class surface's order, see host_tables)
            per = [self.static_at(int(s)) for s in slots]
            gains1 = ops.cdl_gains(D(np.stack([a for a, _ in per])), D(np.stack([v for _, v in per])), times,
                                   A_los=self.Alos, nu_los=self.nulos)
        H = hsub = None
        need_h = self.freqDomain or (self.chanEst == "Perfect" and not no_matrix) or self.prg
        if need_h:
            fusedcs = None
        elif self.setup_spec is not None:
            fusedcs = ops.chan_setup_paths(gains1, self.coeff, self.setup_spec, self.L, self.K, self.nfft, 12 * self.first_prb, 12)
        else:
            fusedcs = ops.chan_setup(gains1, self.coeff, self.L, self.K, self.nfft, 12 * self.first_prb, 12)
        if fusedcs is not None:     # time-domain link, estimated channel, wideband precoder: the CIR is needed for nothing else
            hsub, off = fusedcs
        else:
            cir1, off = ops.cir(gains1, self.coeff, self.L)
        if need_h:
            H = ops.channel_matrix(cir1, off, self.L, self.K, self.nfft)
        if precoder is not None:
            if self.prg:
                raise ValueError("precoder=: one wideband matrix per slot (this link precodes per PRG)")
            F = torch.as_tensor(precoder).to(device=self.dev, dtype=torch.complex128).contiguous()
            if tuple(F.shape) != (n, self.nt, self.nl):
                raise ValueError(f"precoder must be (n_slots, Nt={self.nt}, Nl={self.nl}), got {tuple(F.shape)}")
        elif self.prg:        # one SVD precoder per PRG: mean channel of the group -> right singular vectors
            hm = ops.group_mean(H, self.prg_k0, self.prg_nk)                    # (n, G, Nr, Nt)
            G = hm.shape[1]
            F = ops.svd_precoder(hm.reshape(n * G, 1, self.nr, self.nt), self.nl).reshape(n, G, self.nt, self.nl)
        else:
            if hsub is None:
                hsub = ops.channel_matrix_sub(cir1, off, self.L, self.K, self.nfft, 12 * self.first_prb, 12)
            F = ops.svd_precoder(hsub, self.nl)                                 # wideband SVD precoder (first PRB)
        # A wideband precoder is the same Nt x Nl matrix on every subcarrier, so it commutes with the modulator: the Nl LAYER
        # grids are modulated (one read of each row) and the precoder goes into the path gains of the channel filter.
        # Per-PRG precoders depend on the subcarrier and are applied to the grid.
        gfold = None if (self.freqDomain or self.prg or self._sep_prec) else ops.fold_precoder(gains1, F)
        return gains1, off, H, F, gfold

    def _run_group(self, slots, snr_db, seed, tb_bits, noise, counters, details, harq=None, precoder=None):
        """One batch of slots with identical geometry.  ``harq`` = one (rv, circ, reset) per codeword: per-slot redundancy
        versions (int32 device tensor), the resident soft buffers and the restart flags of a batched HARQ round."""
        dev, cfg = self.dev, self.cfg
        n = len(slots)
        sis = int(slots[0]) % self.bwp.slotsPerSubFrame
        # device generator keys = the ABSOLUTE slot numbers (a geometry group at mu >= 2 is not a contiguous slot range)
        contiguous = bool(np.all(np.diff(slots) == 1))
        slots_dev = torch.arange(int(slots[0]), int(slots[0]) + n, dtype=torch.int64, device=dev) if contiguous else \
            torch.as_tensor(np.asarray(slots, dtype=np.int64), device=dev)
        sif = slots_dev % self.bwp.slotsPerFrame
        ids = None if contiguous else slots_dev
        snr_lin = torch.full((n,), 10.0 ** (float(snr_db) / 10.0), dtype=torch.float64, device=dev) \
            if np.isscalar(snr_db) else D(10.0 ** (np.float64(snr_db) / 10.0))

        # ---- channel state of each slot.  (It depends on nothing of the Tx bit chain, and since the wideband precoder went into
        # the channel filter's gains the modulator does not wait for it either.  Running it on a second stream beside the Tx
        # chain was tried: its one-workgroup-per-slot kernels and the bit-chain kernels then share the CUs and each takes as
        # much longer as the overlap saves -- 36.04 against 36.04 ms per step -- so it stays on the one stream.)
        # (perfect CSI without a channel matrix: see bin_spec; the precoder must be folded into the gains, details=True wants the matrix)
        pp = self.bin_spec is not None and details is not True and not self.prg and not self._sep_prec
        ch = self._channel_chain(slots, n, slots_dev, precoder=precoder, no_matrix=pp)

        # ---- Tx
        grid = None if self.numCW == 1 else self.templates.index_select(0, sif)   # DMRS-filled (n, Nl, L, K)
        tbs_in, tx_bits = [], []
        for q, cw in enumerate(self.cw):
            if tb_bits is None:     # stream ids: 1 = first codeword (as before), 3 = second; 2 is the noise
                tb = ops.random_bits(n, cw['tbs'], seed, dev, stream_id=1 + 2 * q, batch_offset=int(slots[0]), item_ids=ids)
            else:
                tb = (tb_bits[q] if self.numCW > 1 else tb_bits).to(dev).to(torch.uint8).contiguous()
            tbs_in.append(tb)
            # (first transmissions send nothing beyond the columns of the active rows: their parity is not computed)
            # (the coded-bit buffer of a (codeword, batch size, row count) is kept: its never-written parity columns are cleared once)
            enc_rows = cw['rows'] if harq is None else None
            ekey = (n, enc_rows)
            held = self._enc_buf.get(q)
            coded = ops.ldpc_encode(ops.ldpc_segment(tb, cw['cfg']), cw['cfg'], rows=enc_rows,
                                    _reuse=held[1] if held and held[0] == ekey else None)
            self._enc_buf[q] = (ekey, coded)        # (this link's own buffer: `coded` is not stable from one batch to the next)
            bits = ops.ldpc_rate_match(coded, cw['cfg'], cw['G'], cw['nl'], cw['qm'], rv=0 if harq is None else harq[q][0])
            tx_bits.append(bits)
            if grid is None:    # one codeword: template + scramble + modulate + layer/RE map in one pass over the grid
                if self.re_planes is None:
                    self.re_planes = ops.layer_planes(self.re_inv, self.templates.shape[1])
                if self.waveform == "f32" and self.templates32 is None:
                    self.templates32 = self.templates.to(torch.complex64)
                grid = ops.pdsch_populate(bits, cw['qm'], cw['scr'], self.re_inv,
                                          self.templates32 if self.waveform == "f32" else self.templates, sif, planes=self.re_planes)
            else:
                ops.qam_map(bits, cw['qm'], scr=cw['scr'], re_index=cw['re_index'], out=grid)
        tb = tbs_in[0]

        gains1, off, H, F, gfold = ch
        if self.prg:
            grid = ops.precode_prg(grid, F, self.prg_k2g)                       # (n, Nt, L, K); F is applied from here on
        elif self._sep_prec and not self.freqDomain:
            grid = ops.precode(grid, F)                                         # (NRX_SEPARATE_PRECODER: pdsch.py / grid.py order)

        if self.freqDomain:
            rx = ops.apply_channel_fd(grid if self.prg else ops.precode(grid, F), H)
            _, sigma, nv = ops.noise_level(rx, snr_lin=snr_lin)                 # grid.py:1040-1046
            rxg = ops.add_noise(rx, noise.to(dev), sigma) if noise is not None else \
                ops.awgn(rx, sigma, seed, stream_id=2, batch_offset=int(slots[0]), item_ids=ids)
        else:
            cps = [int(v) for v in (self.sym_lens[sis][:-1] - self.nfft)]
            w = Waveform.windowLength(cps, self.window, self.bwp)
            tx = ops.ofdm_modulate(grid, self.nfft, cps, window_len=w, pad=self.max_delay)      # layers (wideband) | ports (PRG)
            lens = [int(v) for v in self.sym_lens[sis]]
            mult = self.nfft / (12.0 * self.bwp.numRbs)
            # the filter leaves the power sums of its output (getRePower) where its kernel supports that ...
            gmix = gains1 if gfold is None else gfold
            got = None
            if self.td_spec is not None and not self._sep_power:       # overlap-save; None where it has no instantiation
                got = ops.apply_td_os(tx, gmix, self.td_spec, self.td_hist, lens, power=(self.nfft, snr_lin, mult, float(self.nfft)))
            if got is None and not self._sep_power:
                got = ops.apply_td_paths(tx, gmix, self.taps, self.tap_off, lens, hist=self.td_hist,
                                         power=(self.nfft, snr_lin, mult, float(self.nfft)))
            if got is not None:
                ry, sigma, nv = got
            else:           # ... else a second pass over the waveform
                ry = ops.apply_td_paths(tx, gains1 if gfold is None else gfold, self.taps, self.tap_off, lens, hist=self.td_hist)
                _, sigma, nv = ops.noise_level(ry, snr_lin=snr_lin, mult=mult, nv_mult=float(self.nfft),
                                               gather=self._cp_gather(sis, ry.shape[-1]))
            if noise is not None:
                rxg = ops.ofdm_demodulate(ops.add_noise(ry, noise.to(dev), sigma), self.nfft, cps, self.K, t_off=off, grid64=True)
            else:       # the noise is generated while the demodulator loads its samples (= ops.awgn, then demodulate)
                rxg = ops.ofdm_demodulate(ry, self.nfft, cps, self.K, t_off=off, awgn=(sigma, seed, 2, int(slots[0]), ids),
                                          grid64=True)     # (waveform="f32": the estimator / equaliser / demapper run in float64)

        # ---- Rx
        hest = None
        got = None
        if self.chanEst == "Perfect" and pp:
            got = ops.mmse_equalize_paths(rxg, gfold, self.bin_spec, off, nv, self.nfft, sym_mask=self.data_sym_mask)
            if got is None:     # the kernel has no instantiation for this shape: the matrix route (getChannelMatrix -> H @ F -> equalize)
                _, _, H, F, _ = self._channel_chain(slots, n, slots_dev, precoder=precoder, no_matrix=False)
        if got is not None:
            eq, sc = got
        elif self.chanEst == "Perfect":
            hest = ops.effective_channel_prg(H, F, self.prg_k2g) if self.prg else ops.effective_channel(H, F)
            eq, sc = ops.mmse_equalize(rxg, hest, nv)
        elif self.polarInt:
            hest = ops.chest_ls_ex(rxg, self.pilots, self.port_ks, self.dmrs_syms, l_cdm=self.l_cdm, k_cdm=self.k_cdm,
                                   pil_set=sif.to(torch.int32), polar=True)
            eq, sc = ops.mmse_equalize(rxg, hest, nv)
        elif details is True or self.n_tg > 2 or self.nl > 4:
            hest = ops.chest_ls(rxg, self.pilots, self.port_ks, self.dmrs_syms, l_cdm=self.l_cdm, k_cdm=self.k_cdm,
                                pil_set=sif.to(torch.int32))
            eq, sc = ops.mmse_equalize(rxg, hest, nv)
        else:       # estimate + equalise fused: the (L, K, Nr, Nl) estimate is never written out
            eq, sc = ops.chest_ls_mmse(rxg, self.pilots, self.port_ks_d, self.dmrs_syms, nv, l_cdm=self.l_cdm,
                                       k_cdm=self.k_cdm, pil_set=sif.to(torch.int32), sym_mask=self.data_sym_mask)
        per_cw = []
        for q, cw in enumerate(self.cw):
            ccfg = cw['cfg']
            # rate recovery + decode + CRC/merge in one launch where an instantiation exists (same bits): the demapper then
            # writes every code block's LLRs de-interleaved.  With details=True the LLRs are wanted in the reference's order.
            fuse = (harq is None and self.decoder == "f64" and self.useMax and details is not True
                    and ops.ldpc_fused_supported(ccfg, cw['nl'], cw['qm'], cw['G'], cw['rows']))
            # ... and where no fused entry applies, a first transmission without repetition needs no rate-recovery pass either: the
            # demapper's stores put every LLR where the decoder reads it (nrx_qam_demap_rr_*; None when E_r wraps around)
            ldt = torch.float32 if self.decoder == "f32" else torch.float64
            rr = None
            if not fuse and harq is None and self.useMax and details is not True and not self._sep_rr:
                n_cols = min(ccfg.N // ccfg.Zc, ccfg.K // ccfg.Zc - 2 + ops.ldpc_rows_read(ccfg, cw['rows'], self.decoder == "f32"))
                rr = ops.qam_demap(eq, nv, cw['qm'], scr=cw['scr'], re_index=cw['re_index'], scales=sc, nv_floor=1e-10,
                                   llr_dtype=ldt, rate_recovered=(ccfg, cw['nl'], n_cols) if not self._poison else
                                   (ccfg, cw['nl'], n_cols, torch.full((n * ccfg.C, ccfg.N), float('nan'), dtype=ldt, device=dev)))
            llr = None if rr is not None else \
                ops.qam_demap(eq, nv, cw['qm'], scr=cw['scr'], re_index=cw['re_index'], scales=sc, nv_floor=1e-10,
                              exact=not self.useMax, llr_dtype=ldt, code_blocks=(ccfg.C, cw['nl']) if fuse else None)
            if self.certStages is not None:
                if not fuse:
                    raise ValueError("certifiedExit needs the fused float64 decoder entry (BG1, Zc 384, first transmission, <= 15 rows, max-log LLRs)")
                got = ops.ldpc_recover_decode_merge_certified(llr, ccfg, cw['nl'], cw['qm'], self.certStages, self.numIter, rows=cw['rows'],
                                                              flags=self.certFlags, max_sweeps=self.certSweeps, in_kernel=self.certInKernel,
                                                              persistent=self.certPersistent)
                fused = None if got is None else got[:2]
                self.last_exit_iter = None if got is None else got[2]
            elif fuse and self.firstPassIter is not None:     # two passes, both on the fused entry, the failing blocks' list on the device
                fused = ops.ldpc_recover_decode_merge_two_pass(llr, ccfg, cw['nl'], cw['qm'], self.firstPassIter, self.numIter,
                                                               rows=cw['rows'], stages=self.passStages)
            else:
                fused = ops.ldpc_recover_decode_merge(llr, ccfg, cw['nl'], cw['qm'], self.numIter, rows=cw['rows']) if fuse else None
            if fuse and fused is None:
                raise RuntimeError("nrx_ldpc_recover_decode_merge_f64 declined a configuration ops.ldpc_fused_supported accepted")
            if fused is not None:
                tb_out, cb_ok = fused
                if counters is not None:
                    ops.count_errors(cb_ok, tb_out, tbs_in[q], counters)
                per_cw.append(dict(tb=tbs_in[q], cb_ok=cb_ok, tb_out=tb_out, llr=llr))
                continue
            if rr is not None:
                pass
            elif harq is None:
                rr = ops.ldpc_rate_recover(llr, ccfg, cw['nl'], cw['qm'])
            else:
                rr = ops.ldpc_rate_recover(llr, ccfg, cw['nl'], cw['qm'], rv=harq[q][0], circ=harq[q][1], reset=harq[q][2])
            rows = cw['rows'] if harq is None else None        # HARQ soft buffers fill other columns: all rows
            if harq is not None and self.firstPassIter is None and cw['rows'] is not None:
                # ... of the retransmissions.  A process that starts a new block holds rv 0 alone in a fresh buffer, which is
                # the single-shot case: the rows whose parity is punctured are exact no-ops there as well, and the truncated
                # graph has the on-chip float64 instantiation.  One small host read per round (n_proc flags) splits the batch.
                dec = self._harq_decode(rr, ccfg, cw['rows'], harq[q][2])
                tb_out, cb_ok, _ = ops.ldpc_crc_merge(dec, ccfg, want_tb_crc=False)
            elif self.firstPassIter is None:
                dec = ops.ldpc_decode(rr, ccfg, self.numIter, rows=rows)
                tb_out, cb_ok, _ = ops.ldpc_crc_merge(dec, ccfg, want_tb_crc=False)
            else:
                dec = ops.ldpc_decode(rr, ccfg, self.firstPassIter, rows=rows)
                _, cb_ok, _ = ops.ldpc_crc_merge(dec, ccfg, want_tb=False)
                fail = (cb_ok.reshape(-1) == 0).nonzero().reshape(-1)          # host read: how many blocks go on
                if fail.numel():
                    dec.index_copy_(0, fail, ops.ldpc_decode(rr.index_select(0, fail), ccfg, self.numIter, rows=rows))
                tb_out, cb_ok, _ = ops.ldpc_crc_merge(dec, ccfg, want_tb_crc=False)
            if counters is not None:
                ops.count_errors(cb_ok, tb_out, tbs_in[q], counters)
            per_cw.append(dict(tb=tbs_in[q], cb_ok=cb_ok, tb_out=tb_out, llr=llr))
        cb_ok = per_cw[0]['cb_ok']
        if details == "verdicts":       # per-code-block CRC verdicts + decoded transport blocks (nothing extra is materialised)
            d = dict(cb_ok=cb_ok, tb_out=per_cw[0]['tb_out'])
            if self.numCW > 1:
                d['cw'] = [dict(cb_ok=c['cb_ok'], tb_out=c['tb_out']) for c in per_cw]
            return d
        if details:
            d = dict(per_cw[0], eq=eq, sc=sc, hest=hest, rxg=rxg, F=F, off=off, nv=nv, sigma=sigma, grid=grid, bits=tx_bits[0])
            if not self.freqDomain:     # the waveforms either side of the channel filter (tools/r6/stage_diff.py)
                d.update(tx=tx, ry=ry)
            if self.numCW > 1:
                d['cw'] = per_cw
            return d
        if harq is not None:
            return dict(cw=[dict(cb_ok=c['cb_ok'], tb_out=c['tb_out'], llr=c['llr'], bits=tx_bits[q]) for q, c in enumerate(per_cw)])
        return None

    # ----------------------------------------------------------------------------------------------- HARQ
    def run_harq(self, n_proc, n_rounds, snr_db, seed=0, rvSequence=(0, 2, 3, 1), maxTries=4, harqType="IR", slot0=0,
                 state=None, tb_bits=None, noise=None, trace=None, proc_offset=0, n_proc_total=None):
        """Batched HARQ (reference harq.py + Playground/HARQ/Harq.ipynb loop) for ``n_proc`` HARQ processes.

        Round k transmits slots slot0 + k*n_proc + p, p = 0..n_proc-1, one per process -- the reference's round-robin
        (harq.py:626-631) -- in one batch.  Per process and codeword the state is what ``HarqCW`` keeps (harq.py:145-202):
        the transport block being sent, the try counter, and the soft buffer the decoder accumulates into; all of it stays
        on the GPU (soft buffers: n_proc*C x (Ncb-F) LLRs resident in HBM), and the retransmission decision is taken
        on the device; the one host read per round is the n_proc "new block" flags that let the decoder run first
        transmissions on the truncated graph (they are the single-shot case).  A codeword whose block decodes (every
        code-block CRC passes) or times out after ``maxTries`` starts a new block in its next round.  A PDSCH with more
        than 4 layers carries two codewords per process, each with its own try counter, redundancy version and buffer
        (harq.py:477); the statistics add them up like ``HarqEntity`` does.

        Parity mode: ``tb_bits`` (n_rounds, n_proc, TBS) -- a list of two such tensors for two codewords -- are the NEW
        transport blocks offered to the processes in each round (taken by the codewords that start a new block, like
        ``random.bits`` in the notebook loop) and ``noise`` (n_rounds, n_proc, Nr, samples) the standard-normal complex noise
        of every slot; ``trace`` (a list) receives one dict per round: rv / new flags / LLRs / CRC verdicts per codeword and a
        copy of the soft buffers after the round.

        Sharding by process streams (``run_harq_sharded``): this call simulates the processes ``proc_offset .. proc_offset + n_proc - 1``
        of ``n_proc_total`` -- process p of round k still transmits slot slot0 + k*n_proc_total + p, so every process sees the transport
        blocks, channel and noise it sees in a single-process run of all n_proc_total of them, and the statistics simply add up.

        Returns (stats, state): stats with the fields of ``HarqEntity`` (txBlocks/rxBlocks/txBits/rxBits per try,
        numTimeouts, throughput and BLER in percent, meanTries); pass ``state`` back in to continue the run."""
        dev = self.dev
        n_proc_total = int(n_proc if n_proc_total is None else n_proc_total)
        proc_offset = int(proc_offset)
        if proc_offset < 0 or proc_offset + n_proc > n_proc_total:
            raise ValueError("run_harq: the processes [proc_offset, proc_offset + n_proc) must lie inside [0, n_proc_total)")
        if harqType not in ("IR", "CC"):
            raise ValueError("harqType must be 'IR' or 'CC'")
        rvs = torch.tensor(list(rvSequence) if harqType == "IR" else [0], dtype=torch.int32, device=dev)
        ncw = self.numCW
        if state is None:
            ft = torch.float32 if self.decoder == "f32" else torch.float64
            state = dict(tb=[torch.zeros((n_proc, c['tbs']), dtype=torch.uint8, device=dev) for c in self.cw],
                         tries=[torch.zeros(n_proc, dtype=torch.int64, device=dev) for _ in self.cw],
                         circ=[torch.zeros((n_proc * c['cfg'].C, c['cfg'].N - c['cfg'].F), dtype=ft, device=dev) for c in self.cw],
                         tx=torch.zeros(maxTries, dtype=torch.int64, device=dev), rx=torch.zeros(maxTries, dtype=torch.int64, device=dev),
                         tx_bits=torch.zeros(maxTries, dtype=torch.int64, device=dev),
                         rx_bits=torch.zeros(maxTries, dtype=torch.int64, device=dev),
                         timeouts=torch.zeros(1, dtype=torch.int64, device=dev), next_slot=int(slot0))
        elif state['tb'][0].shape[0] != n_proc or state['tx'].numel() != maxTries or len(state['tb']) != ncw:
            raise ValueError("state does not belong to this (n_proc, maxTries, codewords) configuration")
        if tb_bits is not None and not isinstance(tb_bits, (list, tuple)):
            tb_bits = [tb_bits]
        if tb_bits is not None and (len(tb_bits) != ncw or any(t.shape[0] < n_rounds or t.shape[1] != n_proc for t in tb_bits)):
            raise ValueError("tb_bits: one (n_rounds, n_proc, TBS) tensor per codeword")
        if noise is not None and (noise.shape[0] < n_rounds or noise.shape[1] != n_proc):
            raise ValueError("noise: (n_rounds, n_proc, Nr, samples)")
        ones = torch.ones(n_proc, dtype=torch.int64, device=dev)
        spsf = self.bwp.slotsPerSubFrame
        for k in range(n_rounds):
            s0 = state['next_slot'] + proc_offset          # this shard's first slot of the round
            slots = np.arange(s0, s0 + n_proc)
            new, rv = [], []
            for q, c in enumerate(self.cw):
                tries = state['tries'][q]
                nq = tries == 0
                if tb_bits is None:     # (stream ids as in run(): 1 = first codeword, 3 = second)
                    fresh = ops.random_bits(n_proc, c['tbs'], seed, dev, stream_id=1 + 2 * q, batch_offset=s0)
                else:
                    fresh = tb_bits[q][k].to(dev).to(torch.uint8)
                state['tb'][q] = torch.where(nq[:, None], fresh, state['tb'][q])
                # a new block always starts at rv 0 (HarqCW.reset, harq.py:119); rvSequence is consulted for retransmissions only
                # (harq.py:199, 580-583)
                rv.append(torch.where(nq, torch.zeros_like(rvs[0]), rvs[tries % rvs.numel()]).contiguous())
                new.append(nq)
            # one sub-batch per slot geometry (a single one for mu <= 1): the soft buffers of a sub-batch's processes are
            # gathered, combined into, and written back
            geoms = {}
            for i, n in enumerate(slots):
                geoms.setdefault(tuple(self.sym_lens[n % spsf]), []).append(i)
            oks = [torch.zeros(n_proc, dtype=torch.bool, device=dev) for _ in self.cw]
            outs = {}
            for _, sel in geoms.items():
                whole = len(sel) == n_proc
                # (a host -> device copy waits for the stream to drain, i.e. for the previous round's decoder: only where a sub-batch needs it)
                sel_t = None if whole else torch.as_tensor(sel, dtype=torch.int64, device=dev)
                hq, rows_of = [], []
                for q, c in enumerate(self.cw):
                    C = c['cfg'].C
                    if whole:
                        circ_q, rws = state['circ'][q], None
                    else:
                        rws = (sel_t[:, None] * C + torch.arange(C, device=dev)[None, :]).reshape(-1)
                        circ_q = state['circ'][q].index_select(0, rws)
                    rows_of.append(rws)
                    hq.append((rv[q] if whole else rv[q][sel_t].contiguous(), circ_q,
                               (new[q] if whole else new[q][sel_t]).to(torch.uint8).contiguous()))
                tbs_sel = [t if whole else t[sel_t] for t in state['tb']]
                out = self._run_group(slots[np.asarray(sel)], snr_db, seed, tbs_sel if ncw > 1 else tbs_sel[0],
                                      None if noise is None else (noise[k] if whole else noise[k][sel_t.to(noise.device)]), None, False, harq=hq)
                for q, c in enumerate(self.cw):
                    if not whole:
                        state['circ'][q].index_copy_(0, rows_of[q], hq[q][1])
                    okq = out['cw'][q]['cb_ok'].reshape(len(sel), c['cfg'].C).to(torch.bool).all(1)
                    oks[q] = okq if whole else oks[q].index_copy(0, sel_t, okq)
                outs[tuple(sel)] = out
            for q, c in enumerate(self.cw):
                tries, ok = state['tries'][q], oks[q]
                state['tx'].index_add_(0, tries, ones)                                  # harq.py:185-186
                state['rx'].index_add_(0, tries, ok.to(torch.int64))                    # harq.py:187-190
                state['tx_bits'].index_add_(0, tries, ones * c['tbs'])
                state['rx_bits'].index_add_(0, tries, ok.to(torch.int64) * c['tbs'])
                nxt = tries + 1
                timeout = (~ok) & (nxt == maxTries)                                     # harq.py:196-199
                state['timeouts'] += timeout.sum()
                state['tries'][q] = torch.where(ok | timeout, torch.zeros_like(nxt), nxt)
            if trace is not None:
                trace.append(dict(slots=slots.copy(), rv=[r.clone() for r in rv], new=[n.clone() for n in new], ok=[o.clone() for o in oks],
                                  groups={g: [dict(llr=c['llr'], cb_ok=c['cb_ok'], tb_out=c['tb_out'], bits=c['bits']) for c in o['cw']] for g, o in outs.items()},
                                  circ=[c.clone() for c in state['circ']], tb=[t.clone() for t in state['tb']]))
            state['next_slot'] = s0 - proc_offset + n_proc_total
        return harq_stats(state['tx'].cpu().numpy(), state['rx'].cpu().numpy(), state['tx_bits'].cpu().numpy(), state['rx_bits'].cpu().numpy(),
                          int(state['timeouts'].item())), state


def harq_stats(tx, rx, txb, rxb, nto):
    """The statistics ``HarqEntity`` prints (harq.py:185-199, 337-395) from the per-try counters: transmitted / received blocks and bits
    by number of earlier tries, time-outs."""
    tx, rx, txb, rxb = (np.asarray(v, dtype=np.int64) for v in (tx, rx, txb, rxb))
    max_tries = len(tx)
    return dict(txBlocks=tx, rxBlocks=rx, txBits=txb, rxBits=rxb, numTimeouts=int(nto),
                throughput=100.0 * rxb.sum() / max(txb.sum(), 1), bler=100.0 * (tx.sum() - rx.sum()) / max(tx.sum(), 1),
                meanTries=float(((rx * np.arange(max_tries)).sum() + nto * max_tries) / max(rx.sum() + nto, 1)))


# ------------------------------------------------------------------------------------------------------ sweeps
def shard_slots(slot0, n_slots, world, rank):
    """Contiguous share of [slot0, slot0 + n_slots) owned by ``rank`` (slots are independent given their
    absolute index, so a sweep shards by slot range with no data-path communication)."""
    q, r = divmod(int(n_slots), int(world))                # balanced: the first r ranks take one more, none is left empty while another holds two
    lo = rank * q + min(rank, r)
    return slot0 + lo, q + (1 if rank < r else 0)


def run_harq_sharded(link, n_proc, n_rounds, snr_db, state=None, **kw):
    """``PdschLink.run_harq`` sharded by independent process streams over the ranks of the default process group (harq.py:626-631: the
    HARQ processes of an entity take turns and share nothing).  Rank r simulates a contiguous share of the ``n_proc``
    processes -- each with the slots, transport blocks, channel and noise it has in a single-process run -- and ONE all-reduce(SUM)
    of the int64 per-try counters (txBlocks, rxBlocks, txBits, rxBits, time-outs: 4 maxTries + 1 numbers) gives every rank the
    statistics of the whole entity.  Returns (stats, state) like run_harq; ``state`` is this rank's shard (pass it back in)."""
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if on else (1, 0)
    lo, cnt = shard_slots(0, n_proc, world, rank)
    if cnt > 0:
        _, state = link.run_harq(cnt, n_rounds, snr_db, state=state, proc_offset=lo, n_proc_total=n_proc, **kw)
        vec = torch.cat([state['tx'], state['rx'], state['tx_bits'], state['rx_bits'], state['timeouts'].reshape(-1)]).to(torch.int64)
    else:       # (more ranks than processes: an empty shard still joins the collective)
        vec = torch.zeros(4 * int(kw.get('maxTries', 4)) + 1, dtype=torch.int64, device=link.dev)
    vec = vec.clone()
    if on:
        if dist.get_backend() == 'gloo':        # (host-side collectives: the CPU tests, several ranks on one GPU)
            vec = vec.cpu()
        dist.all_reduce(vec)
    v = vec.cpu().numpy()
    m = (len(v) - 1) // 4
    return harq_stats(v[:m], v[m:2 * m], v[2 * m:3 * m], v[3 * m:4 * m], int(v[4 * m])), state


def run_sweep(link, snrs_db, n_slots, seed=0, batch=64, slot0=0):
    """BLER sweep over a fixed SNR grid, sharded over the ranks of the default process group (if initialised).

    Every rank simulates its own slot range of every SNR point (``link.run`` in throughput mode: the device generator is
    keyed by (seed, absolute slot), so the union of the ranks' results does not depend on the number of ranks), then
    ONE all-reduce (RCCL over xGMI with backend 'nccl') sums the int64 [nSnr, 4] counter table
    (blockErrors, totalBlocks, bitErrors, totalBits) -- the path's only collective.  Returns the table as NumPy."""
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if on else (1, 0)
    lo, cnt = shard_slots(slot0, n_slots, world, rank)
    table = torch.zeros((len(snrs_db), 4), dtype=torch.int64, device=link.dev)
    for i, snr in enumerate(snrs_db):
        done = 0
        while done < cnt:
            nb = min(int(batch), cnt - done)
            link.run(lo + done, nb, float(snr), seed=seed, counters=table[i])
            done += nb
    if on:      # (also with a single rank: the collective is the same code path whatever the world size)
        dist.all_reduce(table)
    return table.cpu().numpy()
