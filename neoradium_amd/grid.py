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

    def estimateChannelLS(self, rsInfo, meanCdm=True, polarInt=False, kernel='linear'):
        """grid.py:874-975 with DMRS pilots, CDM averaging and linear interpolation -> (H (L,K,Nr,P), noiseVar est.).

        Built: meanCdm=True, kernel='linear', polarInt False (complex-linear along the subcarriers) or True (unwrapped
        angle and magnitude interpolated separately, utils.py:38-42 -- what PDSCH-endToEnd.ipynb asks for).  The second
        return value is the reference's noise estimate (grid.py:808-837 + scaleNoiseVar), including its habit of
        sampling every port at the last port's pilot subcarriers."""
        from .dmrs import DMRS
        if not isinstance(rsInfo, DMRS):
            raise NotImplementedError("estimateChannelLS: only DMRS-based estimation is built (CSI-RS is out of scope)")
        if not meanCdm or kernel != 'linear':
            raise NotImplementedError("estimateChannelLS: only meanCdm=True, kernel='linear' is built")
        dmrs = rsInfo
        pil, ks, ds = dmrs.getPilots()
        if self.shape[1:] != (self.bwp.symbolsPerSlot, 12 * self.bwp.numRbs):
            raise ValueError("The Grid size (%dx%d) does not match Reference Signals (%dx%d)." %
                             (self.shape[1], self.shape[2], self.bwp.symbolsPerSlot, 12 * self.bwp.numRbs))
        l_cdm, k_cdm = dmrs.symbols, (4 if dmrs.enhanced else 2)
        rx, pd = D(self.grid[None]), D(pil[None])
        h, hk = ops.chest_ls_ex(rx, pd, ks, list(ds), l_cdm=l_cdm, k_cdm=k_cdm, polar=bool(polarInt), want_hk=True)
        cp_min = int(min(self.bwp.symbolLens)) - self.bwp.nFFT
        raw, num = ops.chest_noise_var(rx, pd, ks, list(ds), hk, self.bwp.nFFT, cp_min, l_cdm=l_cdm, k_cdm=k_cdm)
        est = self.scaleNoiseVar(float(N(raw)[0]), pil.shape[0], l_cdm, k_cdm, num)
        return N(h)[0], est

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
