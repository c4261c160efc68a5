"""Modem (reference modulation.py:17-234): QAM mapping and LLR demapping on the GPU."""
import numpy as np

from . import ops
from ._dev import D, N


class Modem:
    mod2qm = {'BPSK': 1, 'QPSK': 2, '16QAM': 4, '64QAM': 6, '256QAM': 8, '1024QAM': 10}

    def __init__(self, modulation='QPSK'):
        if modulation not in self.mod2qm:
            raise ValueError("Unsupported modulation \"%s\"!" % (modulation))
        self.modulation = modulation
        self.qm = self.mod2qm[modulation]
        self._const = None

    @property
    def constellation(self):
        """TS 38.211 5.1 constellation indexed by the MSB-first bit tuple (modulation.py:60-74), from the kernel."""
        if self._const is None:
            qm = self.qm
            v = np.arange(1 << qm)
            bits = ((v[:, None] >> np.arange(qm - 1, -1, -1)[None, :]) & 1).astype(np.uint8).reshape(1, -1)
            self._const = N(ops.qam_map(D(bits), qm))[0]
        return self._const

    @property
    def symbolOrder(self):
        c = self.constellation
        return np.argsort(1000 * c.real - c.imag)

    def modulate(self, bitstreams):
        bitstreams = np.asarray(bitstreams)
        flat = bitstreams.ndim == 1
        b = np.uint8(bitstreams.reshape(1, -1) if flat else bitstreams)
        if b.shape[1] % self.qm:
            raise ValueError("The length of 'bitstream' (%d) must be a multiple of 'qm' (%d)!" % (b.shape[1], self.qm))
        out = N(ops.qam_map(D(b), self.qm))
        return out[0] if flat else out

    def getLLRsFromSymbols(self, symbols, noiseVar, useMax=True):
        symbols = np.asarray(symbols, dtype=np.complex128)
        flat = symbols.ndim == 1
        s = symbols.reshape(1, -1) if flat else symbols.reshape(symbols.shape[0], -1)
        llr = N(ops.qam_demap(D(s), D(np.float64([noiseVar])), self.qm, exact=not useMax))
        return llr[0] if flat else llr.reshape(symbols.shape[:-1] + (-1,))

    def demodulate(self, symbols, noiseVar, useMax=True):
        return np.int8(self.getLLRsFromSymbols(symbols, noiseVar, useMax) <= 0)

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        title = "Modem Properties:" if title is None else title
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + title + "\n"
        s += pad + "  Modulation Type ...........: %s\n" % (self.modulation)
        s += pad + "  Qm ........................: %d\n" % (self.qm)
        s += pad + "  Num constellation points ..: %d\n" % (1 << self.qm)
        if getStr:
            return s
        print(s)
