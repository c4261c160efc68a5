"""5G NR LDPC encoder / decoder classes (reference ldpc.py:670-1619) over the libnrx kernels.

NumPy in / NumPy out like the reference; every heavy step (CRC, segmentation, encode, rate match, rate
recovery, layered min-sum decode, CRC check) is one kernel launch on the GPU.
"""
import os

import numpy as np

from . import _lib, ops
from ._dev import D, N
from .chancodebase import ChanCodeBase
from .utils import deprecated

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'ldpc_bg.npz')
liftingSizeSets = [[a << j for j in range(8) if (a << j) <= 384] for a in (2, 3, 5, 7, 9, 11, 13, 15)]
_MOD2QM = {'BPSK': 1, 'QPSK': 2, '16QAM': 4, '64QAM': 6, '256QAM': 8, '1024QAM': 10}


class LdpcBase(ChanCodeBase):
    def __init__(self, baseGraphNo=1, modulation='QPSK', txLayers=1, nRef=0):
        super().__init__()
        self.baseGraphNo = baseGraphNo
        if self.baseGraphNo not in [1, 2]:
            raise ValueError("'baseGraphNo' must be 1 or 2!")
        self.modulation = modulation
        if self.modulation not in _MOD2QM:
            raise ValueError("Invalid 'modulation' value!")
        self.qm = _MOD2QM[modulation]
        self.maxCodeBlockSize = 8448 if baseGraphNo == 1 else 3840
        self.txBlockSize = 0
        self.numCodeBlocks = 0
        self.codeBlockSize = 0
        self.liftingSize = 0
        self.setIndex = -1
        self._baseGraph = None
        self.numFillerBits = 0
        self.txLayers = txLayers
        self.nRef = nRef
        self._cfg = None

    def initialize(self, txBlockSize):
        """Sizes of TS 38.212 5.2.2 (C, Zc, iLS, K) from the C ABI's nrx_ldpc_config (ldpc.py:859-892)."""
        if self.txBlockSize == txBlockSize and self._cfg is not None:
            return
        cfg = _lib.ldpc_config(self.baseGraphNo, int(txBlockSize))
        self._cfg = cfg
        self._baseGraph = None
        self.txBlockSize = int(txBlockSize)
        self.numCodeBlocks, self.liftingSize, self.setIndex, self.codeBlockSize = cfg.C, cfg.Zc, cfg.iLS, cfg.K

    @property
    def baseGraph(self):
        """(46x68 | 42x52) int16 base graph: -1 = no edge, else shift mod Zc (ldpc.py:776-789)."""
        if self._baseGraph is None:
            assert self.setIndex >= 0 and self.liftingSize > 0, "'Base Graph' not available. Encoder not initialized yet!"
            d = np.load(_DATA)
            r, c, s = (d[f'bg{self.baseGraphNo}_{k}'] for k in ('row', 'col', 'shift'))
            bg = -np.ones((46, 68) if self.baseGraphNo == 1 else (42, 52), dtype=np.int16)
            bg[r, c] = s[:, self.setIndex] % self.liftingSize
            self._baseGraph = bg
        return self._baseGraph

    def getRateMatchedCbLens(self, g, c):
        return np.int32(_lib.ldpc_cb_lens(int(g), int(c), self.txLayers, self.qm))

    def isValidCodedBlock(self, codedBlock):
        """All parity checks of the lifted graph hold (the reference stops after the first base-graph row,
        ldpc.py:841-843; this checks every row)."""
        z, bg = self.liftingSize, self.baseGraph
        w = np.int64(np.asarray(codedBlock)).reshape(-1, z)
        for row in bg:
            acc = np.zeros(z, dtype=np.int64)
            for j in np.nonzero(row >= 0)[0]:
                acc += np.roll(w[j], -int(row[j]))
            if (acc % 2).any():
                return False
        return True

    @deprecated(replacement="isValidCodedBlock")
    def isValidCodeword(self, codeWord):
        return self.isValidCodedBlock(codeWord)

    def print(self, indent, title, getStr):
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + title + "\n"
        s += pad + "  Base Graph:         %d\n" % (self.baseGraphNo)
        s += pad + "  Modulation:         %s\n" % (self.modulation)
        s += pad + "  Number of layers:   %d\n" % (self.txLayers)
        if getStr:
            return s
        print(s)


class LdpcEncoder(LdpcBase):
    def __init__(self, baseGraphNo=1, modulation='QPSK', txLayers=1, nRef=0, targetRate=449 / 1024):
        super().__init__(baseGraphNo, modulation, txLayers, nRef)
        self.targetRate = targetRate

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        s = super().print(indent, "LDPC Encoder Properties:" if title is None else title, True)
        s += indent * ' ' + "  Target Rate:        %s\n" % (str(self.targetRate))
        if getStr:
            return s
        print(s)

    def doSegmentation(self, txBlock, fillerBit=0):
        """TS 38.212 5.2.2: (B,) bits incl. TB CRC -> (C,K) with CB CRCs and zero fillers (ldpc.py:981-1030)."""
        txBlock = np.asarray(txBlock)
        self.initialize(len(txBlock))
        cbs = N(ops.ldpc_segment(D(np.uint8(txBlock)[None]), self._cfg, add_tb_crc=False)).astype(np.int8)
        self.numFillerBits = self._cfg.F
        return cbs

    def encode(self, codeBlocks, puncture=True):
        """(C,K) -> (C,N) LDPC coded blocks, first 2Zc bits punctured by default (ldpc.py:1033-1090)."""
        codeBlocks = np.asarray(codeBlocks)
        assert self._cfg is not None and codeBlocks.shape[1] == self._cfg.K
        return N(ops.ldpc_encode(D(np.uint8(codeBlocks)), self._cfg, puncture)).astype(np.int8)

    def rateMatch(self, codedBlocks, g=None, concatCBs=True, rv=0):
        """Bit selection + interleaving, TS 38.212 5.4.2 (ldpc.py:1093-1159)."""
        on_dev = hasattr(codedBlocks, 'is_cuda')          # HARQ keeps the coded blocks as a device tensor
        if not on_dev:
            codedBlocks = np.asarray(codedBlocks)
        c, nz = codedBlocks.shape
        assert nz in [66 * self.liftingSize, 50 * self.liftingSize]
        if rv not in [0, 1, 2, 3]:
            raise ValueError("Invalid 'rv' value! It must be one of 0, 1, 2, or 3.")
        if g is None:
            g = int(np.ceil((self.txBlockSize - 24) / self.targetRate))
        dev_in = codedBlocks if on_dev else D(np.uint8(codedBlocks))
        out = N(ops.ldpc_rate_match(dev_in, self._cfg, int(g), self.txLayers, self.qm, rv,
                                    self.nRef))[0].astype(np.int8)
        if concatCBs:
            return out
        ends = np.cumsum(self.getRateMatchedCbLens(g, c))
        return np.split(out, ends[:-1])

    @deprecated(replacement="getRateMatchedCodeBlocks")
    def getRateMatchedCodeWords(self, txBlock, g=None, concatCBs=True, addCrc=True):
        return self.getRateMatchedCodeBlocks(txBlock, g, concatCBs, addCrc)

    def getRateMatchedCodeBlocks(self, txBlock, g=None, concatCBs=True, addCrc=True):
        """CRC24A + segmentation + encode + rate match, all on the device without host round trips."""
        txBlock = np.asarray(txBlock)
        self.initialize(len(txBlock) + (24 if addCrc else 0))
        cfg = self._cfg
        self.numFillerBits = cfg.F
        if g is None:
            g = int(np.ceil((self.txBlockSize - 24) / self.targetRate))
        cbs = ops.ldpc_segment(D(np.uint8(txBlock)[None]), cfg, add_tb_crc=addCrc)
        coded = ops.ldpc_encode(cbs, cfg)
        out = N(ops.ldpc_rate_match(coded, cfg, int(g), self.txLayers, self.qm, 0, self.nRef))[0].astype(np.int8)
        if concatCBs:
            return out
        ends = np.cumsum(self.getRateMatchedCbLens(g, cfg.C))
        return np.split(out, ends[:-1])

    def getDecoder(self):
        return LdpcDecoder(self.baseGraphNo, self.modulation, self.txLayers, self.nRef)


class LdpcDecoder(LdpcBase):
    def __init__(self, baseGraphNo=1, modulation='QPSK', txLayers=1, nRef=0):
        super().__init__(baseGraphNo, modulation, txLayers, nRef)

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        s = super().print(indent, "LDPC Decoder Properties:" if title is None else title, True)
        if getStr:
            return s
        print(s)

    def recoverRate(self, rxBlock, txBlockSize, harq=None):
        """Inverse rate matching with HARQ soft combining (ldpc.py:1330-1418): (G,) LLRs -> (C,N) float64."""
        self.initialize(txBlockSize + 24)
        cfg = self._cfg
        self.numFillerBits = cfg.F
        ncb = cfg.N if self.nRef == 0 else min(cfg.N, self.nRef)
        rv, circ = 0, None
        if harq is not None:
            rv = harq.rv
            buf = harq.decBuffer
            if buf is None:
                buf = np.zeros((cfg.C, ncb - cfg.F), dtype=np.float64)
            assert tuple(buf.shape) == (cfg.C, ncb - cfg.F), \
                f"HARQ buffer shape mismatch! It must be a {cfg.C}x{ncb - cfg.F} NumPy array!"
            circ = buf if hasattr(buf, 'is_cuda') else D(np.float64(buf))     # soft buffer lives in HBM
        out = ops.ldpc_rate_recover(D(np.float64(rxBlock)[None]), cfg, self.txLayers, self.qm, rv, self.nRef, circ)
        if harq is not None:
            harq.decBuffer = circ
        # (C, Ncb) like the reference (ldpc.py:1414-1418); the library's N-wide matrix has zeros beyond a limited buffer
        return N(out)[:, :ncb]

    def decode(self, rxCodeBlock, numIter=5, onlyInfoBits=True, outputBelief=False, certifiedExit=None):
        """Layered normalised min-sum (ldpc.py:1495-1581).  float64 input -> the bit-exact float64 kernel;
        float32 input -> the single-precision throughput kernel.

        ``certifiedExit`` (not in the reference, which always runs ``numIter`` iterations; float64 hard decisions of the information
        bits only): ascending iteration counts after which a code block may stop -- only where the stability certificate proves
        that its hard decisions are already those of the full run, so the returned bits are the reference's.
        ``self.lastExitIter`` then holds the iteration every block stopped at (0 = it ran all ``numIter``)."""
        rx = np.asarray(rxCodeBlock)
        if rx.dtype != np.float32:
            rx = np.float64(rx)
        if self._cfg is not None and self.nRef and rx.ndim == 2 and rx.shape[1] == min(self._cfg.N, self.nRef) < self._cfg.N:
            # limited buffer (LBRM): the positions beyond Ncb were never transmitted -- LLR 0, like any punctured bit.  (The reference's
            # decode stops on this shape, ldpc.py:1538; TS 38.212 5.4.2.1 decodes it so.)
            rx = np.concatenate([rx, np.zeros((rx.shape[0], self._cfg.N - rx.shape[1]), dtype=rx.dtype)], axis=1)
        if self._cfg is None or rx.shape[1] != self._cfg.N:
            raise ValueError("decode: call recoverRate first (or the block length does not match the configuration)")
        if certifiedExit is not None:
            if outputBelief or not onlyInfoBits or rx.dtype != np.float64:
                raise ValueError("certifiedExit gives float64 hard decisions of the information bits (onlyInfoBits=True, outputBelief=False)")
            hard, ex = ops.ldpc_decode_certified(D(rx), self._cfg, numIter, [int(v) for v in np.atleast_1d(certifiedExit)])
            self.lastExitIter = N(ex)
            return N(hard).astype(np.int8)
        out = N(ops.ldpc_decode(D(rx), self._cfg, numIter, only_info=onlyInfoBits, belief=outputBelief))
        return out if outputBelief else out.astype(np.int8)

    def checkCrcAndMerge(self, rxCodedBlocks):
        """CRC24B per code block (CRC24A when C=1), strip, merge (ldpc.py:1584-1619)."""
        cfg = self._cfg
        tb, ok, _ = ops.ldpc_crc_merge(D(np.uint8(rxCodedBlocks)), cfg)
        ok = N(ok)[0].astype(bool)
        merged = N(tb)[0].astype(np.int8)
        if cfg.C == 1:
            return merged, [bool(ok[0])]
        return merged, ok
