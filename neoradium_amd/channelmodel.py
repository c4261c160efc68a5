"""Tapped-delay-line channel base class (reference channelmodel.py:28-491) over the HIP kernels.

Per slot the device computes: path gains at the 15 symbol-start instants (ops.cdl_gains, from the subclass's
static ray coefficients), the discrete CIR and timing offset (ops.cir), and on request the frequency-domain
channel matrix (ops.channel_matrix) or the time-domain filtering of a waveform (ops.apply_td).  Filter design
(Kaiser-sinc fractional-delay bank, delay quantisation) is host-side setup.
"""
import numpy as np

from . import ops
from ._dev import D, N
from .random import random
from .utils import freqStr, toDb, toLinear
from .waveform import Waveform


class ChannelModel:
    def __init__(self, bwp, **kwargs):
        if bwp is None:
            raise ValueError("The bandwidth part cannot be 'None'!")
        self.bwp = bwp
        self.sampleRate = bwp.sampleRate
        self.dopplerShift = kwargs.get('dopplerShift', 40)
        self.carrierFreq = kwargs.get('carrierFreq', 3.5e9)
        self.normalizeGains = kwargs.get('normalizeGains', True)
        self.normalizeOutput = kwargs.get('normalizeOutput', True)
        self.txDir = kwargs.get('txDir', 'Downlink')
        if self.txDir not in ['Downlink', 'Uplink']:
            raise ValueError("Unsupported 'txDir' (%s). It must be one of 'Downlink' or 'Uplink'." % (self.txDir))
        self.filterLen = kwargs.get('filterLen', 16)
        self.stopBandAtten = kwargs.get('stopBandAtten', 80)
        self.delayQuantSize = kwargs.get('delayQuantSize', 64)
        self.seed = kwargs.get('seed', None)
        self.rangen = random if self.seed is None else random.getGenerator(self.seed)
        self.allFirs = self.buildFirs()
        self._static = None          # device tensors of the time-invariant ray coefficients (subclass)
        self._dev = {}               # device tensors of the current slot

    # ------------------------------------------------------------------------------------------- lifecycle
    def restart(self, restartRanGen=False, applyToBwp=True):
        if applyToBwp:
            self.bwp.restart()
        self.filterDelays = np.array([(self.filterLen - 1) // 2])
        self.curSlotStart = 0
        self.nextSlotStart = 0
        self._static = None
        self.prepareForNextSlot()

    def goNext(self, applyToBwp=True):
        self.curSlotStart = self.nextSlotStart
        if applyToBwp:
            self.bwp.goNext()

    def getMaxDelay(self):
        return int(np.ceil(self.pathDelays.max() * self.sampleRate / 1e9 + self.filterDelays.max()))

    @property
    def coherenceTime(self):
        return np.sqrt(9 / (16 * np.pi)) / self.dopplerShift

    @property
    def nrNt(self):
        raise NotImplementedError("The derived channel model classes must implement the `nrNt` property!")

    def staticCoefficients(self):
        """Subclass hook: (A (Nr,Nt,N,M) complex, nu (N,M) Hz, A_los (Nr,Nt) | None, nu_los) -- everything of the
        path gains that does not depend on time (without the output normalisations)."""
        raise NotImplementedError("The derived channel model classes must implement `staticCoefficients`!")

    # ----------------------------------------------------------------------------------------------- filters
    def buildFirs(self):
        """(delayQuantSize+1) x filterLen bank of Kaiser-windowed sinc fractional-delay filters
        (channelmodel.py:249-289): filter q delays by q/delayQuantSize of a sample."""
        a = self.stopBandAtten
        beta = 0.1102 * (a - 8.7) if a > 50 else (0 if a < 21 else 0.5842 * (a - 21) ** 0.4 + 0.07886 * (a - 21))
        q, n = self.delayQuantSize, self.delayQuantSize * self.filterLen
        fir = np.kaiser(n + 1, beta) * np.sinc(np.arange(-n // 2, n // 2 + 1, 1) / q)
        fir[0:n + 1:q] = 0                    # exact zeros at the integer offsets ...
        fir[n // 2] = 1                       # ... except the centre tap
        firs = fir[:-1].reshape(self.filterLen, q).T
        return np.concatenate([firs, np.roll(firs[:1], -1)])

    def getCoeffMatrix(self):
        """(numPaths, coeffLen) tap matrix of the integer + quantised fractional path delays (channelmodel.py:292-318)."""
        d = self.pathDelays * 1e-9 * self.sampleRate
        di = np.int32(d)
        frac = d - di
        self.filterDelay = np.clip(self.filterLen // 2 - 1 - di.min(), 0, None)
        di = di + self.filterDelay
        qi = np.int32(np.round(self.delayQuantSize * (1 - frac)))
        clen = int(di.max() + self.filterLen // 2 + 1)
        m = np.zeros((len(d), clen))
        for p in range(len(d)):
            s = di[p] - self.filterLen // 2 + 1
            m[p, s:s + self.filterLen] = self.allFirs[qi[p]]
        return m

    # ---------------------------------------------------------------------------------------------- per slot
    def _normalisation(self):
        s = 1.0
        if self.normalizeOutput:
            s /= np.sqrt(self.nrNt[0])
        if self.normalizeGains:
            s /= np.sqrt(toLinear(self.pathPowers).sum())
        return s

    def _staticOnDevice(self):
        if getattr(self, '_static_per_slot', False):     # statistical models redraw their coefficients for every slot
            self._static = None
        if self._static is None:
            A, nu, Alos, nulos = self.staticCoefficients()
            sc = self._normalisation()
            self._static = (D(np.complex128(A * sc)), D(np.float64(nu)),
                            None if Alos is None else D(np.complex128(Alos * sc)), float(nulos))
        return self._static

    def prepareForNextSlot(self):
        """Gains at the nc+1 symbol-start instants, CIR and timing offset of the current slot
        (channelmodel.py:321-354)."""
        if self.nextSlotStart > self.curSlotStart:
            return
        symLens = self.bwp.getSymLens().copy()
        slotLen = int(symLens[:-1].sum())
        starts = symLens.copy()
        starts[0] -= self.bwp.nFFT                       # a symbol's useful part starts after its CP
        self.chanGainSamples = self.curSlotStart + np.cumsum(starts)
        A, nu, Alos, nulos = self._staticOnDevice()
        times = D((self.chanGainSamples / self.sampleRate)[None])
        gains1 = ops.cdl_gains(A, nu, times, A_los=Alos, nu_los=nulos)          # (1, nc+1, nr, nt, P)
        self.nextSlotStart = self.curSlotStart + slotLen
        coeff = self.getCoeffMatrix()
        nc = len(symLens) - 1
        cir1, off = ops.cir(gains1, D(coeff), nc)
        self._dev = dict(gains1=gains1, cir1=cir1, off=off)
        self.symLens = symLens
        self.chanGains1 = N(gains1)[0]
        self.chanGains = self.chanGains1[:-1]
        self.coeffMatrix = coeff
        self.cir = N(cir1)[0, :nc]
        self.chanOffset = int(off.item())

    def getChannelGains(self):
        self.prepareForNextSlot()
        return self.chanGains1.copy()

    def getPathGains(self):
        """Un-normalised path gains at the nc+1 instants (what the subclasses' getPathGains returns in the reference)."""
        return self.getChannelGains() / self._normalisation()

    def getTimingOffset(self):
        self.prepareForNextSlot()
        return self.chanOffset

    def getChannelMatrix(self):
        """(nc, K, Nr, Nt) frequency-domain channel of the current slot (channelmodel.py:362-400)."""
        self.prepareForNextSlot()
        nc = len(self.symLens) - 1
        if self.coeffMatrix.shape[1] > self.bwp.nFFT:
            print("WARNING: The delay spread is larger than FFT size! Ignoring larger delays!")
        H = ops.channel_matrix(self._dev['cir1'], self._dev['off'], nc, 12 * self.bwp.numRbs, self.bwp.nFFT)
        return N(H)[0]

    def applyToGrid(self, grid):
        return grid.applyChannel(self.getChannelMatrix())

    def applyToSignal(self, inputSignal):
        """Time-domain filtering of a (Nt, ns) waveform by the time-varying channel (channelmodel.py:403-448)."""
        self.prepareForNextSlot()
        slotLen = int(self.symLens[:-1].sum())
        x = inputSignal if isinstance(inputSignal, np.ndarray) else inputSignal.waveform
        nt_sig, ns = x.shape
        nr, nt = self.nrNt
        if nt_sig != nt:
            raise ValueError("The number of transmit antennas in the signal does not match the channel.")
        if ns < slotLen:
            raise ValueError(f"The inputSignal is too short. It must be at least {slotLen} samples.")
        if ns > self.symLens.sum():
            print("WARNING: The delays are larger than symbol size! Extending gains to match delays!")
        taps, offs = ops.path_taps(self.coeffMatrix, self.filterLen)
        y = ops.apply_td_paths(D(np.complex128(x)[None]), self._dev['gains1'], D(taps), offs, [int(v) for v in self.symLens])
        return Waveform(N(y)[0])

    def applyKFactorScaling(self):
        """TR 38.901 7.7.6 K-factor scaling for LOS profiles (channelmodel.py:472-491)."""
        assert self.hasLos and self.kFactor is not None
        p = toLinear(self.pathPowers)
        kModel = toDb(p[0] / p[1:].sum())
        self.pathPowers[1:] = self.pathPowers[1:] - self.kFactor + kModel
        pd = p * self.pathDelays
        rms = np.sqrt(np.square(pd).sum() / p.sum() - np.square(pd.sum() / p.sum()))
        self.pathDelays /= rms

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + ("Channel Model Properties:" if title is None else title) + "\n"
        s += pad + f"  carrierFreq:     {freqStr(self.carrierFreq)}\n"
        s += pad + f"  normalizeGains:  {str(self.normalizeGains)}\n" + pad + f"  normalizeOutput: {str(self.normalizeOutput)}\n"
        s += pad + f"  txDir:           {self.txDir}\n" + pad + f"  filterLen:       {self.filterLen} samples\n"
        s += pad + f"  delayQuantSize:  {self.delayQuantSize}\n" + pad + f"  stopBandAtten:   {self.stopBandAtten} dB\n"
        s += pad + f"  dopplerShift:    {freqStr(self.dopplerShift)}\n" + pad + f"  coherenceTime:   {self.coherenceTime} sec\n"
        if getStr:
            return s
        print(s)
