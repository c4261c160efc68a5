"""Time-domain waveform (reference waveform.py:24-527) over the HIP kernels."""
import numpy as np

from . import ops
from ._dev import D, N
from .random import random
from .utils import toLinear


class Waveform:
    def __init__(self, waveform, noiseVar=0):
        self.waveform = waveform
        self.noiseVar = noiseVar

    @property
    def shape(self): return self.waveform.shape
    @property
    def numPorts(self): return self.waveform.shape[0]
    @property
    def length(self): return self.waveform.shape[1]
    def __getitem__(self, key): return self.waveform[key]
    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        title = "Waveform Properties:" if title is None else title
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + title + "\n"
        s += pad + "  Number of Ports: %d\n" % (self.numPorts) + pad + "  Length: %d\n" % (self.length)
        if self.noiseVar > 0:
            s += pad + "  Noise Var.: %s\n" % (str(self.noiseVar))
        if getStr:
            return s
        print(s)

    # ---------------------------------------------------------------------------------------------- noise
    @staticmethod
    def _fftSampleIndexes(bwp):
        """Sample positions the demodulator reads (CP removed), waveform.py:107-117 / :503-510."""
        cps = bwp.getCpLens().astype(np.int64)
        sym = cps + bwp.nFFT
        starts = np.concatenate([[0], np.cumsum(sym[:-1])])
        off = np.int64(np.round(cps * 0.5))
        return ((cps[:, None] - off[:, None] + np.arange(bwp.nFFT)) % bwp.nFFT + off[:, None] + starts[:, None]).reshape(-1)

    def getRePower(self, bwp):
        idx = self._fftSampleIndexes(bwp)
        width = self.shape[1]
        gather = (np.arange(self.shape[0])[:, None] * width + idx[None, :]).reshape(-1).astype(np.int32)
        var, _, _ = ops.noise_level(D(self.waveform[None]), gather=gather)
        return var.item() / (12 * bwp.numRbs)

    def getNoiseStd(self, snr, bwp):
        """waveform.py:119-142."""
        return float(np.sqrt(self.getRePower(bwp) * bwp.nFFT / snr))

    def addNoise(self, **kwargs):
        """waveform.py:145-292."""
        noise = kwargs.get('noise', None)
        if noise is not None:
            if self.shape != noise.shape:
                raise ValueError("Shape mismatch: Waveform: %s vs Noise: %s" % (str(self.shape), str(noise.shape)))
            return Waveform(self.waveform + noise, noise.var())
        ranGen = kwargs.get('ranGen', random)
        noiseStd = kwargs.get('noiseStd', None)
        if noiseStd is None and kwargs.get('noiseVar', None) is not None:
            noiseStd = np.sqrt(kwargs['noiseVar'])
        if noiseStd is None:
            snrDb = kwargs.get('snrDb', None)
            if snrDb is None:
                raise ValueError("You must specify the noise power using 'snrDb', 'noiseVar', or 'noiseStd'!")
            snr = toLinear(snrDb)
            bwp = kwargs.get('bwp', None)
            if kwargs.get('useRxPower', False) and bwp is not None:
                noiseStd = self.getNoiseStd(snr, bwp)
            else:
                nFFT = bwp.nFFT if bwp is not None else kwargs.get('nFFT', None)
                if nFFT is None:
                    raise ValueError("When using SNR, you must also specify the FFT size!")
                noiseStd = np.sqrt(1 / (snr * self.numPorts * nFFT))
        z = ranGen.normal(0, 1, self.shape + (2,))
        zc = z[..., 0] + 1j * z[..., 1]
        out = ops.add_noise(D(np.complex128(self.waveform)[None]), D(zc[None]), D(np.float64([noiseStd])))
        return Waveform(N(out)[0], noiseStd * noiseStd)

    # ------------------------------------------------------------------------------------------ shape ops
    def pad(self, numPad):
        return Waveform(np.concatenate((self.waveform, np.zeros((self.numPorts, numPad))), axis=1), self.noiseVar)

    def sync(self, timingOffset):
        return Waveform(self.waveform[:, timingOffset:], self.noiseVar)

    def applyChannel(self, channel):
        return channel.applyToSignal(self)

    # ------------------------------------------------------------------------------------------ windowing
    @classmethod
    def getWindowingSize(cls, cpLen, bwp):
        """TS 38.101-1/-2 F.5 (waveform.py:99-109)."""
        if bwp.cpType == 'normal':
            return (cpLen + 1) // 2
        table = {64: 54, 96: 80, 128: 106, 192: 164}
        return table[cpLen] if cpLen in table else int(np.round(cpLen * 0.859))

    @classmethod
    def windowLength(cls, cpLens, windowing, bwp):
        """Overlap length chosen by the ``windowing`` string of ofdmModulate (waveform.py:113-126)."""
        w = str(windowing)
        if w.upper() == 'NONE':
            return 0
        if '%' in w:
            ratio = np.float64(w.replace('%', '')) / 100.0
            return min(int(.5 + ratio * c) for c in cpLens)
        if '.' in w:
            ratio = np.float64(w)
            if ratio < 0 or ratio > 1:
                raise ValueError("The windowing ratio must be between 0 and 1")
            return min(int(.5 + ratio * c) for c in cpLens)
        if w.upper() == 'STD':
            return min(cls.getWindowingSize(int(c), bwp) for c in cpLens)
        n = int(w)
        if n >= min(cpLens):
            raise ValueError("The windowing size must be smaller than CP size")
        return n

    # ------------------------------------------------------------------------------------------ demodulate
    def ofdmDemodulate(self, bwp, f0=0, cpOffsetRatio=0.5):
        """waveform.py:473-527: one slot; the FFT window starts cpOffsetRatio of the way into each CP; f0 > 0 undoes the
        up-conversion phase of Grid.ofdmModulate(f0) (one factor per symbol)."""
        from .grid import Grid
        if not 0.0 <= cpOffsetRatio <= 1.0:
            raise ValueError("'cpOffsetRatio' must be between 0 and 1")
        cps = bwp.getCpLens()
        kk = 12 * bwp.numRbs
        if self.shape[1] < int(cps.sum()) + len(cps) * bwp.nFFT:
            raise ValueError("The waveform is shorter than one slot")
        g = ops.ofdm_demodulate(D(np.complex128(self.waveform)[None]), bwp.nFFT, list(cps), kk, cp_offset_ratio=cpOffsetRatio)
        grid = Grid(bwp, numPlanes=self.shape[0])
        grid.grid = N(g)[0]
        if f0 > 0:                                          # waveform.py:525-526
            symLens = bwp.getSymLens()[:-1]
            symStarts = np.cumsum(np.append(0, symLens[:-1])) + np.asarray(cps)
            grid.grid = grid.grid * np.exp(2j * np.pi * f0 * symStarts / bwp.sampleRate).reshape(1, -1, 1)
        grid.reTypeIds = np.full(grid.shape, grid.retNameToId["RX_DATA"], dtype=np.uint8)
        grid.noiseVar = self.noiseVar * bwp.nFFT        # time -> frequency domain (waveform.py:523)
        return grid
