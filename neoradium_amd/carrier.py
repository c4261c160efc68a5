"""Carrier / BandwidthPart numerology (host side; reference carrier.py:37-470).

All lengths are in samples of the bandwidth part's sample rate.  The reference hard-wires 30.72 MHz
(nFFT = 2048 >> mu, hence at most 169 PRB @15 kHz / 84 PRB @30 kHz, carrier.py:147-150).  EXTENSION (no oracle):
when the allocation does not fit, the FFT size and the sample rate are scaled together by a power of two
(273 PRB @30 kHz -> nFFT 4096 at 122.88 MHz, CP 288/352); inside the reference's range nothing changes.
"""
import numpy as np

MAX_CARRIER_BW = 400e6
MAX_RESOURCE_BLOCKS = 275
MIN_RESOURCE_BLOCKS = 20
BASE_SAMPLE_RATE = 30720000.0          # = 1/(Tc*kappa), carrier.py:32-34
SAMPLE_RATE = BASE_SAMPLE_RATE
_SPACINGS = [15, 30, 60, 120, 240, 480, 960]


class BandwidthPart:
    sampleRate = BASE_SAMPLE_RATE

    def __init__(self, carrier, **kwargs):
        self.carrier = carrier
        self.startRb = kwargs.get('startRb', 0)
        self.numRbs = kwargs.get('numRbs', 50)
        spacing = kwargs.get('spacing', 15)
        if spacing in _SPACINGS:
            self.u, self.spacing = _SPACINGS.index(spacing), spacing
        elif spacing in range(7):
            self.u, self.spacing = spacing, _SPACINGS[spacing]
        else:
            raise ValueError("Invalid \"spacing\" values (%s)!" % (str(spacing)))
        self.cpType = kwargs.get('cpType', 'normal').lower()
        if self.cpType not in ('normal', 'extended'):
            raise ValueError("Unsupported cpType \"%s\"! It must be one of 'Normal' or 'Extended'" % (self.cpType))
        self.bandwidth = self.numRbs * 12 * self.spacing * 1000
        self.symbolsPerSlot = 14 if self.cpType == 'normal' else 12
        self.slotsPerSubFrame = 1 << self.u

        base_fft = 2048 >> self.u                      # what carrier.py:147 derives from 30.72 MHz
        want = kwargs.get('nFFT', None)
        if want is None:
            scale = 1
            while self.numRbs >= (base_fft * scale) // 12 and base_fft * scale < 8192:
                scale *= 2                             # EXTENSION beyond the reference's fixed sample rate
        else:
            if want % base_fft or (want // base_fft) & (want // base_fft - 1):
                raise ValueError(f"'nFFT' must be {base_fft} times a power of two")
            scale = want // base_fft
        self.fftScale = scale
        self.sampleRate = BASE_SAMPLE_RATE * scale
        self.nFFT = base_fft * scale
        if self.numRbs >= self.nFFT // 12:
            raise ValueError(f"'numRbs' must be less than nFFT/12 (={self.nFFT // 12})!")
        cps = np.int32([self.getCpLen(l) for l in range(self.symbolsPerSubFrame)])
        # one extra entry (= first symbol) because callers always ask for symbolsPerSlot+1 lengths (carrier.py:153-156)
        self.symbolLens = np.append(cps + self.nFFT, cps[0] + self.nFFT)
        self.dataTimeRatio = self.nFFT / self.symbolLens.mean()

    # -- counters live on the carrier
    @property
    def slotsPerFrame(self): return 10 * self.slotsPerSubFrame
    @property
    def symbolsPerSubFrame(self): return self.symbolsPerSlot * self.slotsPerSubFrame
    @property
    def slotNoInFrame(self): return self.slotNo % self.slotsPerFrame
    @property
    def slotNoInSubFrame(self): return self.slotNo % self.slotsPerSubFrame
    @property
    def avgSlotDuration(self): return 1000. / self.slotsPerSubFrame
    @property
    def cellId(self): return self.carrier.cellId
    @property
    def slotNo(self): return self.carrier.slotNo
    @property
    def frameNo(self): return self.carrier.frameNo
    def goNext(self): self.carrier.goNext()
    def restart(self): self.carrier.restart()

    def getCpLen(self, symIdxInSubframe):
        """CP length of a symbol of the subframe (TS 38.211 5.3.1; carrier.py:245-270), in this BWP's samples."""
        if symIdxInSubframe >= self.symbolsPerSubFrame:
            raise ValueError("'symIdxInSubframe' must be less than the number of OFDM Symbols in a "
                             f"subframe ({self.symbolsPerSubFrame}).")
        if self.cpType == 'normal':
            cp = 144 >> self.u
            if symIdxInSubframe in (0, 7 * (1 << self.u)):
                cp += 16
        else:
            cp = 512 >> self.u
        return cp * self.fftScale

    def getSlotLen(self, slotIndex=None):
        if slotIndex is None:
            slotIndex = self.slotNoInSubFrame
        if slotIndex >= self.slotsPerSubFrame:
            raise ValueError(f"'slotIndex' must be less than number of slots in a subframe ({self.slotsPerSubFrame}).")
        s = slotIndex * self.symbolsPerSlot
        return int(self.symbolLens[s:s + self.symbolsPerSlot].sum())

    def getSymLens(self):
        """Lengths of the next symbolsPerSlot+1 symbols (carrier.py:296-310)."""
        s = self.symbolsPerSlot * self.slotNoInSubFrame
        return self.symbolLens[s:s + self.symbolsPerSlot + 1]

    def getCpLens(self, slotNoInSubFrame=None):
        """CP lengths of the symbols of one slot (helper used by the OFDM kernels)."""
        k = self.slotNoInSubFrame if slotNoInSubFrame is None else slotNoInSubFrame
        s = self.symbolsPerSlot * k
        return (self.symbolLens[s:s + self.symbolsPerSlot] - self.nFFT).astype(np.int32)

    def createGrid(self, numPlanes, useReDesc=False):
        from .grid import Grid
        return Grid(self, numPlanes, useReDesc=useReDesc)

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        from .utils import freqStr
        title = "Bandwidth Part Properties:" if title is None else title
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + title + "\n"
        s += pad + f"  Resource Blocks:    {self.numRbs} RBs starting at {self.startRb} ({self.numRbs * 12} subcarriers)\n"
        s += pad + f"  Subcarrier Spacing: {self.spacing} kHz\n"
        s += pad + f"  CP Type:            {self.cpType}\n"
        s += pad + f"  Bandwidth:          {freqStr(self.bandwidth)}\n"
        s += pad + f"  symbolsPerSlot:     {self.symbolsPerSlot}\n"
        s += pad + f"  slotsPerSubFrame:   {self.slotsPerSubFrame}\n"
        s += pad + f"  nFFT:               {self.nFFT}\n"
        s += pad + f"  frameNo:            {self.frameNo}\n"
        s += pad + f"  slotNo:             {self.slotNo}\n"
        if getStr:
            return s
        print(s)


class Carrier:
    sampleRate = BASE_SAMPLE_RATE

    def __init__(self, **kwargs):
        self.startRb = kwargs.get('startRb', 0)
        self.numRbs = kwargs.get('numRbs', 50)
        self.slotNo = 0
        self.frameNo = 0
        self.cellId = kwargs.get('cellId', 1)
        self.bwps = kwargs.get('bwps', None) or [BandwidthPart(self, **kwargs)]
        self.curBwpIndex = kwargs.get('curBwpIndex', 0)
        self.dcLocation = kwargs.get('dcLocation', 0)

    @property
    def curBwp(self): return self.bwps[self.curBwpIndex]
    @property
    def symbolsPerSlot(self): return self.curBwp.symbolsPerSlot
    @property
    def slotsPerSubFrame(self): return self.curBwp.slotsPerSubFrame
    @property
    def slotsPerFrame(self): return self.curBwp.slotsPerFrame
    @property
    def symbolsPerSubFrame(self): return self.curBwp.symbolsPerSubFrame
    @property
    def frameNoRel(self): return (self.frameNo + self.slotNo // self.slotsPerFrame) % 1024
    @property
    def slotNoInFrame(self): return self.slotNo % self.slotsPerFrame

    def restart(self):
        self.slotNo = 0
        self.frameNo = 0

    def goNext(self):
        """Advance one slot (carrier.py:456-462)."""
        self.slotNo += 1
        if self.slotNo % self.slotsPerFrame == 0:
            self.frameNo += 1

    def createGrid(self, numPorts, useReDesc=False):
        return self.curBwp.createGrid(numPorts, useReDesc=useReDesc)

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        title = "Carrier Properties:" if title is None else title
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + title + "\n"
        s += pad + f"  Cell Id:              {self.cellId}\n"
        s += pad + f"  Bandwidth Parts:      {len(self.bwps)}\n"
        s += pad + f"  Active BWP:           {self.curBwpIndex}\n"
        for i, bwp in enumerate(self.bwps):
            s += bwp.print(indent + 2, f"Bandwidth Part {i}:", True)
        if getStr:
            return s
        print(s)
