"""Resource grid (reference grid.py:47-1246) -- NumPy-in/NumPy-out surface over the HIP kernels.

The per-RE arithmetic (precode, applyChannel, equalize, estimateChannelLS, ofdmModulate, addNoise) runs in
libnrx on the GPU; RE bookkeeping (type map, indexing) is host logic.  For throughput use
``neoradium_amd.engine`` which keeps whole slot batches resident on the device.
"""
import numpy as np

from . import ops
from ._dev import D, N
from .random import random
from .utils import toLinear

_RE_TYPES = ["UNASSIGNED", "RESERVED", "NO_DATA", "DMRS", "PTRS", "CSIRS_NZP", "CSIRS_ZP", "DATA", "PDSCH", "PDCCH",
             "PUSCH", "PUCCH", "PRECODED_MIX", "RX_DATA"]
_RE_COLORS = ["white", "gray", "lightgray", "pink", "yellow", "red", "orange", "cyan", "cornflowerblue", "lime",
              "lightblue", "peachpuff", "violet", "sienna"]
_DATA_TYPES = ("DATA", "PDSCH", "PDCCH", "PUSCH", "PUCCH")


class Grid:
    retMaxPredefine, retMaxCustom = 50, 20
    retIdToName = _RE_TYPES + [None] * (70 - len(_RE_TYPES))
    retColors = _RE_COLORS + ["white"] * (70 - len(_RE_COLORS))
    retNameToId = {n: i for i, n in enumerate(_RE_TYPES)}
    retNumCustom = 0

    def __init__(self, bwp, numPlanes=1, contents="DATA", useReDesc=False, numSlots=1):
        self.bwp = bwp
        if isinstance(contents, str):
            if contents not in _DATA_TYPES:
                raise ValueError("Unsupported grid content type \"%s\"!" % (contents))
            self.defaultReType = self.retNameToId[contents]
        else:
            if not self.retValid(contents):
                raise ValueError("Unsupported grid content type \"%d\"!" % (contents))
            if self.retIdToName[contents] not in _DATA_TYPES:
                raise ValueError("Unsupported grid content type \"%s\"!" % (self.retIdToName[contents]))
            self.defaultReType = contents
        self.numSlots = numSlots
        shape = (numPlanes, numSlots * bwp.symbolsPerSlot, 12 * bwp.numRbs)
        self.grid = np.zeros(shape, dtype=np.complex128)
        self.reTypeIds = np.zeros(shape, dtype=np.uint8)          # UNASSIGNED = 0
        self.reDesc = np.full(shape, "UNASSIGNED", dtype='<U20') if useReDesc else None
        self.noiseVar = 0

    # ------------------------------------------------------------------ RE-type registry
    @classmethod
    def retValid(cls, key):
        return key in cls.retNameToId if isinstance(key, str) else key in cls.retNameToId.values()

    @classmethod
    def retRegister(cls, name, color):
        if name in cls.retNameToId:
            return cls.retNameToId[name]
        if color in cls.retColors:
            raise ValueError("RE Color \"%s\" is already taken!" % (color))
        if cls.retNumCustom >= cls.retMaxCustom:
            raise ValueError("Too many Custom RE types!")
        new = cls.retMaxPredefine + cls.retNumCustom
        cls.retNumCustom += 1
        cls.retNameToId[name], cls.retIdToName[new], cls.retColors[new] = new, name, color
        return new

    # ------------------------------------------------------------------ shape helpers
    @property
    def shape(self): return self.grid.shape
    @property
    def numPlanes(self): return self.grid.shape[0]
    numPorts = numPlanes
    numLayers = numPlanes
    @property
    def numSymbols(self): return self.grid.shape[1]
    @property
    def numSubcarriers(self): return self.grid.shape[2]
    @property
    def numRBs(self): return self.grid.shape[2] // 12
    @property
    def size(self): return self.grid.size

    def __getattr__(self, name):
        if name in ("startRb", "numRbs", "nFFT", "symbolsPerSlot", "slotsPerSubFrame", "slotsPerFrame",
                    "symbolsPerSubFrame"):
            return getattr(self.__dict__['bwp'], name)
        raise AttributeError("Class '%s' does not have any property named '%s'!" % (self.__class__.__name__, name))

    def __getitem__(self, key):
        return self.grid[key]

    def __setitem__(self, key, values):
        """grid[idx] = value | (value, "TYPE") | "TYPE"  (reference grid.py 'Resource Grid Indexing')."""
        if isinstance(values, tuple):
            values, name = values
        elif isinstance(values, str):
            values, name = 0, values
        else:
            name = None
        if name is None:
            ret = self.defaultReType
        else:
            if not self.retValid(name):
                raise ValueError("Unknown content type \"%s\"!" % (name))
            ret = self.retNameToId[name]
        if self.reDesc is not None:
            self.reDesc[key] = self.retIdToName[ret]
        self.grid[key] = values
        self.reTypeIds[key] = ret

    def reTypeAt(self, p, l, k):
        return self.retIdToName[self.reTypeIds[p, l, k]]

    def getReIndexes(self, reTypeStr=None):
        if reTypeStr is None:
            reTypeStr = self.retIdToName[self.defaultReType]
        if not self.retValid(reTypeStr):
            raise ValueError("Unknown RE Content type \"%s\"!" % (reTypeStr))
        return np.where(self.reTypeIds == self.retNameToId[reTypeStr])

    def getReValues(self, reTypeStr=None):
        return self.grid[self.getReIndexes(reTypeStr)]

    def getStats(self):
        stats = {"GridSize": self.grid.size}
        for name, rid in self.retNameToId.items():
            n = int(np.count_nonzero(self.reTypeIds == rid))
            if n:
                stats[name] = n
        return stats

    def clone(self):
        g = Grid(self.bwp, self.numPlanes, self.defaultReType, self.reDesc is not None, self.numSlots)
        g.grid, g.reTypeIds, g.noiseVar = self.grid.copy(), self.reTypeIds.copy(), self.noiseVar
        return g

    def _like(self, data, retName="RX_DATA", noiseVar=0):
        g = Grid(self.bwp, numPlanes=data.shape[0], numSlots=self.numSlots)
        g.grid = data
        g.reTypeIds = np.full(data.shape, self.retNameToId[retName], dtype=np.uint8)
        g.noiseVar = noiseVar
        return g

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        title = "Resource Grid Properties:" if title is None else title
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + title + "\n"
        s += pad + "  startRb: %d\n" % (self.startRb) + pad + "  numRbs: %d\n" % (self.numRbs)
        s += pad + "  numSlots: %d\n" % (self.numSlots)
        s += pad + "  Data Contents: %s\n" % (self.retIdToName[self.defaultReType])
        s += pad + "  Size: %d\n" % (self.size) + pad + "  Shape: %s\n" % (str(self.shape))
        if self.noiseVar > 0:
            s += pad + "  Noise Var.: %s\n" % (str(self.noiseVar))
        s += self.bwp.print(indent + 2, "Bandwidth Part:", True)
        if getStr:
            return s
        print(s)

    # ------------------------------------------------------------------ GPU stages
    def precode(self, f):
        """grid.py:456-518.  ``f``: (Nt,Nl) matrix, or a list of (rbList, (Nt,Nl)) per PRG."""
        if isinstance(f, list):
            nt, nl = f[0][1].shape
            perk = np.zeros((self.numSubcarriers, nt, nl), dtype=np.complex128)
            for rbs, fg in f:
                for rb in rbs:
                    perk[rb * 12:rb * 12 + 12] = fg
            # per-subcarrier precoder: one device call per distinct PRG matrix is wasteful; do it per group
            out = np.zeros((nt,) + self.shape[1:], dtype=np.complex128)
            for rbs, fg in f:
                ks = np.concatenate([np.arange(rb * 12, rb * 12 + 12) for rb in rbs]) if len(rbs) else np.arange(0)
                if len(ks) == 0:
                    continue
                sub = np.ascontiguousarray(self.grid[:, :, ks])
                out[:, :, ks] = N(ops.precode(D(sub[None]), D(np.complex128(fg))))[0]
            data = out
        else:
            if not isinstance(f, np.ndarray):
                raise ValueError("'f' must be a 2D NumPy array or a list of tuples.")
            if f.shape[1] != self.numLayers:
                raise ValueError("The last dimension of 'f' (%d) must match the first dimension of the grid (%d)" %
                                 (f.shape[-1], self.shape[0]))
            data = N(ops.precode(D(self.grid[None]), D(np.complex128(f))))[0]
        pg = Grid(self.bwp, data.shape[0], self.defaultReType, numSlots=self.numSlots)
        pg.grid = data
        types = self.reTypeIds[0].copy()
        mixed = (self.reTypeIds != self.reTypeIds[0:1]).any(0)
        types[mixed] = self.retNameToId["PRECODED_MIX"]
        pg.reTypeIds[:] = types
        return pg

    def applyChannel(self, channelMatrix):
        """grid.py:978-1018: per-RE H x (frequency-domain channel)."""
        ll, kk, nr, nt = channelMatrix.shape
        if nt != self.numPorts:
            raise ValueError("Mismatch in the number of transmitter antennas (%d vs %d)!" % (nt, self.numPorts))
        rx = N(ops.apply_channel_fd(D(self.grid[None]), D(np.complex128(channelMatrix))))[0]
        return self._like(rx)

    def equalize(self, hf, noiseVar=None):
        """grid.py:626-694: MMSE equalisation -> (eqGrid, llrScales)."""
        if (self.shape[0] != hf.shape[2]) or (self.shape[1] != hf.shape[0]) or (self.shape[2] != hf.shape[1]):
            raise ValueError("Mismatch in the number of receiver antennas, OFDM symbols, or subcarriers!")
        nv = self.noiseVar if noiseVar is None else noiseVar
        nv = max(1e-8, nv)
        eq, sc = ops.mmse_equalize(D(self.grid[None]), D(np.complex128(hf)[None]), D(np.float64([nv])))
        # like the reference the equalised grid has hf.shape[2] planes allocated but carries the layer estimates
        g = self._like(N(eq)[0], noiseVar=nv)
        return g, N(sc)[0]

    # the 8-8-4-1 ReLU network of grid.py:697-737 that maps the raw pilot-residual variance to a noise variance
    _NV = (np.float64([[6.25861, -0.22737, -8.51406, -0.25593, 0.08617, 0.54746, -10.5016, -0.0075],
                       [0.05773, -0.08806, 0.03222, 0.65573, -1.05669, -0.00781, 0.01074, -0.02898],
                       [-11.48739, -18.84534, 9.54569, -0.02089, 9.92439, 0.07408, 11.41916, -34.07344],
                       [0.71498, 4.52607, -0.35023, 0.05907, 2.24553, 0.06049, 0.47961, 0.44182],
                       [0.84015, 0.14097, 0.20389, -0.45147, 0.12305, -0.51977, 0.37225, 0.12104],
                       [0.41917, 10.52318, 3.35156, 0.58207, -24.37617, 0.33745, -1.11957, 1.07133],
                       [-0.12522, -1.82239, 0.90271, -0.06134, 10.43859, 0.37885, 1.36096, -0.70045],
                       [0.00109, -0.00328, -0.00657, -0.16279, -0.00351, -0.28476, 0.00053, -0.00117]]),
           np.float64([0.60641, 0.06111, 0.24848, 0., 0.32098, 0., -0.21224, 0.007]),
           np.float64([[0.10102, 0.22608, 0.32803, -0.11752], [-0.01549, 0.39246, -0.30703, 0.12527],
                       [-0.02698, 0.09462, -0.31409, 0.03994], [-0.08645, -0.00781, 0.52137, 0.45963],
                       [0.07151, -0.27656, 0.23206, -0.06437], [-0.0154, 0.07408, -0.15198, -0.4007],
                       [-0.17055, -0.06038, -0.8417, 0.43372], [-3.12708, 2.03716, -3.90529, 1.21203]]),
           np.float64([0.54406, 0.36443, -0.21105, 0.35659]),
           np.float64([[0.04271], [0.07268], [0.0702], [-0.16217]]), np.float64([0.72121]))

    def scaleNoiseVar(self, rawNoiseVar, numTx, lCdm, kCdm, numVar):
        """grid.py:697-737 (host scalar arithmetic): the raw variance is returned above 20 dB raw SNR, otherwise the
        small network maps (raw SNR, spacing, ports, Nr, K, lCdm, kCdm, number of residuals) to an SNR."""
        rr, kk = self.shape[0], self.shape[2]
        rawSnrDb = 10.0 * np.log10(1 / (rawNoiseVar * rr))
        if rawSnrDb > 20:
            return rawNoiseVar
        w1, b1, w2, b2, w3, b3 = self._NV
        x = np.float64([rawSnrDb, self.bwp.spacing, numTx, rr, kk, lCdm, kCdm, numVar])
        snrDb = (np.maximum(np.maximum(x.dot(w1) + b1, 0).dot(w2) + b2, 0).dot(w3) + b3)[0]
        return 1 / (10.0 ** (snrDb / 10.0) * rr)

    def _rsTables(self, rsInfo):
        """Pilot tables of the reference signal for the current slot: (lCdm, kCdm, number of ports, groups) where a group
        is (port indices, pilot symbols, pilots (Pg, nLs, nKs), subcarriers (Pg, nKs)) of ports that share their symbols
        (all DMRS ports do; CSI-RS CDM groups can sit on different symbols, grid.py:746-760)."""
        from .csirs import CsiRsConfig
        from .dmrs import DMRS
        if isinstance(rsInfo, DMRS):
            pil, ks, ds = rsInfo.getPilots()
            return rsInfo.symbols, (4 if rsInfo.enhanced else 2), pil.shape[0], [(list(range(pil.shape[0])), list(ds), pil, ks)]
        if not isinstance(rsInfo, CsiRsConfig):
            raise ValueError("'rsInfo' must be a 'CsiRsConfig' or a 'DMRS' object")
        lCdm, kCdm = {1: (1, 1), 2: (1, 2), 4: (2, 2), 8: (4, 2)}[rsInfo.csiRsSetList[0].csiRsList[0].cdmSize]
        rsGrid = self.bwp.createGrid(rsInfo.numPorts)
        rsInfo.populateGrid(rsGrid)
        nzp = rsGrid.reTypeIds == rsGrid.retNameToId["CSIRS_NZP"]
        groups = {}
        for p in range(rsGrid.shape[0]):
            ls = np.flatnonzero(nzp[p].any(1))
            if len(ls) == 0:
                raise ValueError("CSI-RS port %d has no resource elements in this slot" % (p))
            ks = np.flatnonzero(nzp[p, ls[0]])
            groups.setdefault((tuple(ls), len(ks)), []).append((p, ks, rsGrid.grid[p][np.ix_(ls, ks)]))
        out = [([p for p, _, _ in g], list(key[0]), np.stack([v for _, _, v in g]), np.int32([k for _, k, _ in g]))
               for key, g in groups.items()]
        return lCdm, kCdm, rsGrid.shape[0], out

    _tapCache = {}

    @classmethod
    def _taps(cls, key, build):
        if key not in cls._tapCache:
            if len(cls._tapCache) > 64:
                cls._tapCache.clear()
            cls._tapCache[key] = build()
        return cls._tapCache[key]

    def estimateChannelLsEx(self, rsInfo, meanCdm=True, polarInt=True, int2d=True, kernel='thin_plate_spline', neighbors=12,
                            smoothing=0.0, degree=None):
        """grid.py:740-871: LS estimates at the pilots, CDM averaging, interpolation along the subcarriers (cartesian or
        polar), the delay-domain noise estimate, interpolation along the symbols (1-D, or 2-D radial basis functions)
        -> (H (L,K,Nr,P), noise variance, [per port: estimates at the pilot symbols (nTg,K,Nr)]).

        ``rsInfo``: a DMRS (the estimate includes the precoder) or a CsiRsConfig.  kernel='linear' with int2d=False is one
        fused kernel; every other kind goes through tap tables (neoradium_amd.interp) and nrx_interp_taps_f64.  Kept
        from the reference: every port's denoised estimate is sampled at the LAST port's pilot subcarriers in the noise
        estimate (grid.py:823); meanCdm=False fails for time-domain CDM (lCdm > 1), there as a NumPy broadcast error."""
        import torch
        from . import interp
        lCdm, kCdm, pp, groups = self._rsTables(rsInfo)
        rr, ll, kk = self.shape
        if (ll, kk) != (self.bwp.symbolsPerSlot, 12 * self.bwp.numRbs):
            raise ValueError("The Grid size (%dx%d) does not match Reference Signals (%dx%d)." %
                             (ll, kk, self.bwp.symbolsPerSlot, 12 * self.bwp.numRbs))
        if not meanCdm and lCdm > 1:
            raise ValueError("operands could not be broadcast together: meanCdm=False with time-domain CDM (grid.py:829-832)")
        lc, kc = (lCdm, kCdm) if meanCdm else (1, 1)
        ks_last = [ks[ports.index(pp - 1)] for ports, _, _, ks in groups if pp - 1 in ports][0]
        rx = D(self.grid[None])
        cp_min = int(min(self.bwp.symbolLens)) - self.bwp.nFFT
        rp_all = torch.empty((1, ll, kk, rr, pp), dtype=torch.complex128, device=rx.device)
        hk_ports, deltas = [None] * pp, []
        for ports, ds, pil, ks in groups:
            pg, n_ds, n_k = len(ports), len(ds), ks.shape[1]
            if (n_k % kc) or (n_ds % lc):
                raise ValueError("Partial CDMs are not supported in this version.")
            n_g, n_j, rp = n_ds // lc, n_k // kc, rr * pg
            pd = D(pil[None])
            h = None
            if kernel == 'linear':
                h, hk = ops.chest_ls_ex(rx, pd, ks, ds, l_cdm=lc, k_cdm=kc, polar=bool(polarInt), want_hk=True)
            else:
                means = ops.chest_pilot_means(rx, pd, ks, ds, l_cdm=lc, k_cdm=kc, polar=bool(polarInt))
                centres = ks.reshape(pg, n_j, kc).mean(2)

                def build():
                    tabs = [interp.taps_1d(c, np.arange(kk), kernel, neighbors, smoothing) for c in centres]
                    return np.stack([t[0] for t in tabs]), np.stack([t[1] for t in tabs])
                idx, w = self._taps(('k', kernel, neighbors, smoothing, kk, centres.tobytes()), build)
                hk = torch.empty((1, n_g, kk, rr, pg), dtype=torch.complex128, device=rx.device)
                ops.interp_taps(means, idx, w, n_g, rp, n_j, kk, (rp * n_j, n_j, 1), (kk * rp, 1, rp), hk, polar=bool(polarInt))
            if len(ports) == pp and ports[-1] == pp - 1:
                deltas.append(ops.chest_noise_deltas(rx, pd, ks, ds, hk, self.bwp.nFFT, cp_min, l_cdm=lc, k_cdm=kc))
            else:
                deltas.append(ops.chest_noise_deltas(rx, pd, ks, ds, hk, self.bwp.nFFT, cp_min, l_cdm=lc, k_cdm=kc,
                                                     ks_sample=ks_last))
            # along the symbols (grid.py:839-868)
            if n_g == 1:
                h = hk.expand(1, ll, kk, rr, pg)
            elif int2d:
                lm = np.float64(ds).reshape(n_g, lc).mean(1) if meanCdm else np.float64(ds)

                def build2():
                    pts = np.float64(np.meshgrid(np.arange(kk), lm)).reshape(2, -1).T
                    qs = np.float64(np.meshgrid(range(kk), range(ll))).reshape(2, -1).T
                    i2, w2 = interp.rbf_taps(pts, qs, kernel, neighbors, smoothing, None, degree)
                    return i2[None], w2[None]
                idx, w = self._taps(('2d', kernel, neighbors, smoothing, degree, kk, ll, lm.tobytes()), build2)
                h = torch.empty((1, ll, kk, rr, pg), dtype=torch.complex128, device=rx.device)
                ops.interp_taps(hk, idx, w, 1, rp, n_g * kk, ll * kk, (0, 1, rp), (0, 1, rp), h)
            elif kernel != 'linear':
                lm = np.float64(ds).reshape(n_g, lc).mean(1) if meanCdm else np.float64(ds)

                def build1():
                    i1, w1 = interp.taps_1d(lm, np.arange(ll), kernel, neighbors, smoothing)
                    return i1[None], w1[None]
                idx, w = self._taps(('l', kernel, neighbors, smoothing, ll, lm.tobytes()), build1)
                h = torch.empty((1, ll, kk, rr, pg), dtype=torch.complex128, device=rx.device)
                ops.interp_taps(hk, idx, w, 1, kk * rp, n_g, ll, (0, 1, kk * rp), (0, 1, kk * rp), h)
            rp_all[..., ports] = h
            hkn = N(hk)[0]
            for i, p in enumerate(ports):
                hk_ports[p] = hkn[..., i]
        alld = torch.cat(deltas, 1)
        raw, _, _ = ops.noise_level(alld)
        est = self.scaleNoiseVar(float(N(raw)[0]), pp, lCdm, kCdm, alld.shape[1])
        return N(rp_all)[0], est, hk_ports

    def estimateChannelLS(self, rsInfo, meanCdm=True, polarInt=False, kernel='linear'):
        """grid.py:874-975: (H (L,K,Nr,P), estimated noise variance) from DMRS or CSI-RS pilots; interpolation along the
        subcarriers then the symbols with ``kernel`` in 'linear' | 'nearest' | 'quadratic' | 'thin_plate_spline' |
        'multiquadric' (utils.py:26-35), ``polarInt`` for magnitude / unwrapped angle along the subcarriers."""
        return self.estimateChannelLsEx(rsInfo, meanCdm, polarInt, False, kernel)[:2]

    def estimateTimingOffset(self, rxWaveform):
        """grid.py:592-622: this grid (the reference signals only) is OFDM-modulated without windowing and correlated
        with every received antenna; the lag with the largest summed magnitude is the timing offset."""
        ref = self.ofdmModulate(windowing="NONE").waveform
        rxw = np.complex128(rxWaveform.waveform if hasattr(rxWaveform, 'waveform') else rxWaveform)
        if ref.shape[1] > rxw.shape[1]:
            raise ValueError("The received waveform is shorter than the reference signal's (%d vs %d samples)" %
                             (rxw.shape[1], ref.shape[1]))
        nz = np.flatnonzero(np.abs(ref).max(0) > 0)
        start, length = (int(nz[0]), int(nz[-1] - nz[0] + 1)) if len(nz) else (0, 0)
        xc = ops.xcorr_abs(D(rxw), D(ref), start, length)
        return int(xc.argmax().item())

    def ofdmModulate(self, f0=0, windowing="STD"):
        """grid.py:521-582 + waveform.py:380-470."""
        from .waveform import Waveform
        pp, ll, kk = self.shape
        bwp = self.bwp
        if ll % bwp.symbolsPerSlot:
            raise ValueError("the grid must hold whole slots")
        l0 = bwp.slotNoInSubFrame * bwp.symbolsPerSlot
        if ll > bwp.symbolsPerSubFrame - l0:
            raise ValueError("Cannot modulate across subframe boundary! (At most %d symbols)" % (bwp.symbolsPerSubFrame - l0))
        cps = (bwp.symbolLens[l0:l0 + ll] - bwp.nFFT).astype(np.int32)
        w = Waveform.windowLength(cps, windowing, bwp)
        if ll > 112:
            raise NotImplementedError("ofdmModulate: more than 8 slots per call is not built")
        grid = self.grid
        if f0 > 0:
            # up-conversion, TS 38.211 5.4 (grid.py:568-572): one phase factor per symbol, exp(j 2 pi f0 (-t_start - t_cp)),
            # applied to the symbol's subcarriers before the IFFT (a scalar per symbol commutes with it)
            symLens = bwp.symbolLens[l0:l0 + ll]
            n0 = bwp.symbolLens[:l0].sum()
            starts = np.cumsum(np.append(n0, symLens[:-1]))
            phase = np.exp(2j * np.pi * f0 * (-starts - cps) / bwp.sampleRate)
            grid = grid * phase[None, :, None]
        wave = ops.ofdm_modulate(D(grid[None]), bwp.nFFT, list(cps), window_len=w)
        return Waveform(N(wave)[0])

    def getRePower(self):
        return (self.grid.var() / (self.bwp.nFFT ** 2)).item()

    def getNoiseStd(self, snr):
        """grid.py:1040-1046 (signal variance on the device)."""
        var, _, _ = ops.noise_level(D(self.grid[None]))
        return float(np.sqrt(var.item() / snr))

    def addNoise(self, **kwargs):
        """grid.py:1049-1187: noise=, noiseStd=, noiseVar= or snrDb= (+useRxPower)."""
        noise = kwargs.get('noise', None)
        if noise is not None:
            if self.shape != noise.shape:
                raise ValueError(f"Shape Mismatch: Grid: {self.shape} vs Noise: {noise.shape}")
            g = self._like(self.grid + noise, noiseVar=noise.var())
            g.reTypeIds = self.reTypeIds.copy()
            return g
        ranGen = kwargs.get('ranGen', random)
        noiseStd = kwargs.get('noiseStd', None)
        if noiseStd is None:
            noiseVar = kwargs.get('noiseVar', None)
            if noiseVar is not None:
                noiseStd = np.sqrt(noiseVar)
        if noiseStd is None:
            snrDb = kwargs.get('snrDb', None)
            if snrDb is None:
                raise ValueError("You must specify the noise power using 'snrDb', 'noiseVar', or 'noiseStd'!")
            snr = toLinear(snrDb)
            if kwargs.get('useRxPower', False):
                noiseStd = self.getNoiseStd(snr)
            else:
                noiseStd = np.sqrt(1 / (snr * self.shape[0]))       # Matlab convention (grid.py:1182-1185)
        # host PCG64 stream in the reference's draw order; scaled + added on the device
        z = ranGen.normal(0, 1, self.shape + (2,))
        zc = z[..., 0] + 1j * z[..., 1]
        out = ops.add_noise(D(self.grid[None]), D(zc[None]), D(np.float64([noiseStd])))
        g = self._like(N(out)[0], noiseVar=noiseStd * noiseStd)
        g.reTypeIds = self.reTypeIds.copy()
        return g

    def drawMap(self, *a, **k):
        raise NotImplementedError("drawMap (matplotlib plotting) is out of scope of neoradium_amd")
