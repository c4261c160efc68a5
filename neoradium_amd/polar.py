"""Polar coding for the control channels (DCI / PBCH / UCI) on the GPU -- class surface of reference polar.py.

``PolarEncoder`` / ``PolarDecoder`` keep the reference's constructor arguments, properties and method names
(polar.py:117-295, :448-603, :723-982).  The code construction -- mother code size, interleaver patterns, frozen /
message / parity-check sets (polar.py:298-408, TS 38.212 5.3.1, 5.4.1) -- is integer bookkeeping done once per
(A, E) here on the host; encoding, rate matching, rate recovery and CRC-aided successive-cancellation list decoding
run in libnrx (csrc/nrx_polar.hip), a whole batch of code blocks / blind-decode candidates per launch.

Deviations from the reference, all where the reference cannot run:
  * E >= N (repetition, e.g. aggregation level 8): ``recoverRate`` adds the LLRs of the repeated bits as TS 38.212
    5.4.1.2 prescribes; the reference raises a broadcasting error there (polar.py:914-915).
  * the candidate ranking is stable (ties keep the 0-branch / the older path first).  NumPy's default ``argsort``
    in the reference resolves exact ties differently from one CPU to the next (AVX-512 sort network), so only
    tie-free inputs have a reference-defined order.
  * ``nPCwm > 0`` raises NotImplementedError (the reference raises NameError, polar.py:384); ``sclListSize`` <= 8.
"""
import os

import numpy as np

from . import ops
from ._dev import D, N
from .chancodebase import ChanCodeBase

_TABLES = None


def _tables():
    global _TABLES
    if _TABLES is None:
        _TABLES = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'polar_tables.npz'))
    return _TABLES


class PolarBase(ChanCodeBase):
    """Parameters and code construction shared by the encoder and the decoder (reference polar.py:117-445)."""

    def __init__(self, payloadSize=0, rateMatchedLen=0, dataType=None, **kwargs):
        super().__init__()
        self.payloadSize = int(payloadSize)
        self.rateMatchedLen = int(rateMatchedLen)
        self.rateMatchedBlockLen = int(rateMatchedLen)
        self.codeBlockSize = 0
        self.polarCodeSize = 0
        self.inInterleaveIndexes = self.cbInterleaveIndexes = self.sbInterleaveIndexes = None
        self.msgBits, self.frozenBits, self.pcBits = [], [], []
        self._generator = None
        self._dev = {}
        self.dataType = None if dataType is None else dataType.lower()
        if self.dataType is None:
            self.iBIL = kwargs.get('iBIL', False)
            self.nMax = kwargs.get('nMax', 10)
            self.iIL = kwargs.get('iIL', False)
            self.nPC = kwargs.get('nPC', 0)
            self.nPCwm = kwargs.get('nPCwm', 0)
            self.iSeg = kwargs.get('iSeg', False)
            self.crcPoly = kwargs.get('crcPoly', "11")
        elif self.dataType == 'uci':                        # TS 38.212 6.3.1.3.1 / 6.3.1.4.1
            self.iBIL, self.nMax, self.iIL = True, 10, False
            self.nPC = self.nPCwm = 0
            self.iSeg, self.crcPoly = False, '11'
        elif self.dataType in ('pbch', 'dci'):              # TS 38.212 7.1.3-7.1.5 / 7.3.2-7.3.4
            self.iBIL, self.nMax, self.iIL = False, 9, True
            self.nPC = self.nPCwm = 0
            self.iSeg, self.crcPoly = False, '24C'
        else:
            raise ValueError("'dataType' value must be one of 'UCI', 'DCI', or 'PBCH'.")
        if payloadSize > 0 and rateMatchedLen > 0:
            self.initialize(payloadSize, rateMatchedLen)

    # ------------------------------------------------------------------------------------------------------
    def __repr__(self):
        return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        if title is None:
            title = "Polar Coding Properties:"
        pad = indent * ' '
        lines = ["\n" if indent == 0 else "", pad + title + "\n"]

        def row(name, val):
            lines.append(pad + "  " + name + ' ' + '.' * max(1, 31 - len(name)) + ": " + str(val) + "\n")
        row("Payload Size (payloadSize)", self.payloadSize)
        row("Rate-matched Len (rateMatchedLen)", self.rateMatchedLen)
        row("Code Block Size (codeBlockSize)", self.codeBlockSize)
        row("Polar Code Size (polarCodeSize)", self.polarCodeSize)
        row("Max Log2(N) (nMax)", self.nMax)
        row("Segmentation (iSeg)", "Enabled" if self.iSeg else "Disabled")
        row("Code Block CRC (crcPoly)", self.crcPoly)
        row("Input Interleaving (iIL)", "Enabled" if self.iIL else "Disabled")
        row("Coded bit Interleaving (iBIL)", "Enabled" if self.iBIL else "Disabled")
        row("Parity-check bits (nPC, nPCwm)", "%d,%d" % (self.nPC, self.nPCwm))
        text = ''.join(lines)
        if getStr:
            return text
        print(text)

    # ------------------------------------------------------------------------------------------------------
    @classmethod
    def intLog2(cls, num):
        return max(int(num), 1).bit_length() - 1

    @classmethod
    def ceilLog2(cls, num):
        """Reference polar.py:442-445 (works on the integer part of ``num``)."""
        v = int(num) - 1
        return max(v.bit_length(), 1)

    def setIoSizes(self, payloadSize, rateMatchedLen):
        if self.payloadSize != payloadSize or self.rateMatchedLen != rateMatchedLen or self.polarCodeSize == 0:
            self.initialize(payloadSize, rateMatchedLen)

    @property
    def generator(self):
        """Dense N x N generator (only materialised when asked for; the kernels use the XOR butterfly)."""
        if self._generator is None:
            g = np.ones((1, 1), dtype=np.int64)
            for _ in range(self.intLog2(self.polarCodeSize)):
                g = np.kron([[1, 0], [1, 1]], g)
            self._generator = g
        return self._generator

    # ------------------------------------------------------------------------------------------------------
    def initialize(self, payloadSize, rateMatchedLen):
        """Code construction, TS 38.212 5.3.1 / 5.4.1 (reference polar.py:298-408)."""
        self.payloadSize, self.rateMatchedLen = int(payloadSize), int(rateMatchedLen)
        A, Etot = self.payloadSize, self.rateMatchedLen
        tab = _tables()
        if self.dataType == 'uci':
            if A < 12:
                raise ValueError("Polar coding is not supported for UCI with payload size smaller than 12!")
            self.iSeg = (A >= 360 and Etot >= 1088) or A >= 1013
            self.crcPoly = '6' if A < 20 else '11'
            crcLen = int(self.crcPoly)
            K = ((A + 1) // 2 + crcLen) if self.iSeg else (A + crcLen)
            E = Etot // (2 if self.iSeg else 1)
            self.nPC = 3 if 17 < K < 26 else 0
            self.nPCwm = 1 if (self.nPC and (E - K + 3) > 192) else 0
        elif self.dataType is None:
            crcLen = 0 if self.crcPoly is None else self.getCrcLen(self.crcPoly)
            K = ((A + 1) // 2 + crcLen) if self.iSeg else (A + crcLen)
            E = Etot // (2 if self.iSeg else 1)
        else:
            K, E = A + 24, Etot
        self.codeBlockSize, self.rateMatchedBlockLen = K, E

        n1 = self.ceilLog2(E) - 1
        if K / E >= 9 / 16.0 or E > (9 / 8) * (1 << n1):
            n1 += 1
        n = max(min(n1, self.ceilLog2(K / (1 / 8)), self.nMax), 5)
        NN = self.polarCodeSize = 1 << n
        if K + self.nPC > NN:
            raise ValueError("Code block size %d does not fit the polar code size %d" % (K, NN))
        self._generator = None

        self.inInterleaveIndexes = None
        if self.iIL:                                        # 5.3.1.1
            if K > 164:
                raise ValueError("Input interleaving supports at most 164 bits (got %d)" % K)
            pi = tab['input_interleaver'].astype(np.int64) - (164 - K)
            self.inInterleaveIndexes = pi[pi >= 0].tolist()

        i = np.arange(NN)                                   # 5.4.1.1 sub-block interleaver
        sb = (tab['subblock_interleaver'].astype(np.int64)[(i * 32) // NN] * (NN // 32) + i % (NN // 32))
        self.sbInterleaveIndexes = sb.tolist()

        barred = np.zeros(NN, dtype=bool)                   # positions made useless by rate matching
        if E < NN:
            if K / E <= 7.0 / 16:                           # puncturing
                barred[sb[:max(NN - E - 1, 0)]] = True      # (the reference pre-freezes N-E-1 of the N-E, :359)
                cut = ((3 * NN - 2 * E + 3) // 4 - 1) if E >= 3.0 * NN / 4 else ((9 * NN - 4 * E + 15) // 16 - 1)
                barred[:max(cut, 0)] = True
            else:                                           # shortening
                barred[sb[E:]] = True
        rel = tab['reliability'].astype(np.int64)
        rel = rel[rel < NN]
        usable = rel[~barred[rel]]
        chosen = np.sort(usable[len(usable) - (K + self.nPC):])
        if len(chosen) != K + self.nPC:
            raise ValueError("Not enough usable polar sub-channels for K=%d (E=%d, N=%d)" % (K, E, NN))
        isMsg = np.zeros(NN, dtype=bool)
        isMsg[chosen] = True
        self.frozenBits = np.nonzero(~isMsg)[0].tolist()
        self.pcBits = []
        if self.nPC > 0:                                    # 5.3.1.2
            if self.nPCwm > 0:
                raise NotImplementedError("nPCwm > 0 is not supported (the reference fails there too, polar.py:384)")
            self.pcBits = chosen[:self.nPC].tolist()
            chosen = chosen[self.nPC:]
        self.msgBits = chosen.tolist()

        self.cbInterleaveIndexes = None
        if self.iBIL:                                       # 5.4.1.3 triangular interleaver, column-wise read-out
            if E > 8192:
                raise ValueError("The rate-matched output length (%d) should not be larger than 8192!" % E)
            T = int(np.floor(np.sqrt(2 * E)))
            if T * (T + 1) < 2 * E:
                T += 1
            r, c = np.meshgrid(np.arange(T), np.arange(T), indexing='ij')
            start = r * T - (r * (r - 1)) // 2              # first index of row r (row r holds T-r entries)
            k = start + c
            ok = (c < T - r) & (k < E)
            self.cbInterleaveIndexes = k.T[ok.T]
        self._dev = {}
        self._finishInit()

    def _finishInit(self):
        pass

    def _d(self, name, make):
        """Device copy of an index table, made on first use."""
        if name not in self._dev:
            self._dev[name] = D(np.ascontiguousarray(make()))
        return self._dev[name]


class PolarEncoder(PolarBase):
    """Segmentation + CRC, polar encoding and rate matching (reference polar.py:448-603)."""

    def print(self, indent=0, title=None, getStr=False):
        return super().print(indent, "Polar Encoder Properties:" if title is None else title, getStr)

    def doSegmentation(self, txBlock):
        """TS 38.212 5.2.1: one block, or two halves (zero-prepended when A is odd), each with its CRC."""
        txBlock = np.int8(np.asarray(txBlock))
        if self.iSeg:
            if len(txBlock) % 2:
                txBlock = np.concatenate([np.int8([0]), txBlock])
            codeBlocks = txBlock.reshape(2, -1)
        else:
            codeBlocks = txBlock[None, :]
        if self.crcPoly is None:
            return codeBlocks
        return self.appendCrc(codeBlocks, self.crcPoly)

    def encode(self, codeBlocks):
        """(C, K) bits -> (C, N) polar-coded bits (any number of rows; one launch)."""
        codeBlocks = np.asarray(codeBlocks)
        if codeBlocks.ndim != 2 or codeBlocks.shape[1] != self.codeBlockSize:
            raise ValueError("codeBlocks must be C x %d, got %s" % (self.codeBlockSize, codeBlocks.shape))
        return N(self.encodeDevice(D(np.uint8(codeBlocks)))).astype(np.int8)

    def encodeDevice(self, codeBlocks):
        il = None if not self.iIL else self._d('il', lambda: np.int32(self.inInterleaveIndexes))
        pc = None if not self.pcBits else self._d('pc', lambda: np.int32(self.pcBits))
        return ops.polar_encode(codeBlocks, self.polarCodeSize, self._d('msg', lambda: np.int32(self.msgBits)), il, pc)

    def _gather(self):
        NN, K, E = self.polarCodeSize, self.codeBlockSize, self.rateMatchedBlockLen
        sb = np.int64(self.sbInterleaveIndexes)
        if E >= NN:
            pick = sb[np.arange(E) % NN]                    # repetition
        elif K / E <= 7.0 / 16:
            pick = sb[NN - E:]                              # puncturing
        else:
            pick = sb[:E]                                   # shortening
        if self.iBIL:
            pick = pick[self.cbInterleaveIndexes]
        return np.int32(pick)

    def rateMatch(self, codeBlocks):
        """(C, N) -> (C, E): sub-block interleaving, bit selection, coded-bit interleaving (TS 38.212 5.4.1)."""
        codeBlocks = np.asarray(codeBlocks)
        if codeBlocks.ndim != 2 or codeBlocks.shape[1] != self.polarCodeSize:
            raise ValueError("codeBlocks must be C x %d, got %s" % (self.polarCodeSize, codeBlocks.shape))
        return N(self.rateMatchDevice(D(np.uint8(codeBlocks)))).astype(np.int8)

    def rateMatchDevice(self, coded):
        return ops.polar_rate_match(coded, self._d('gather', self._gather))


class PolarDecoder(PolarBase):
    """Rate recovery and CRC-aided SCL decoding (reference polar.py:723-982)."""

    def __init__(self, payloadSize=0, rateMatchedLen=0, dataType=None, **kwargs):
        self.sclListSize = kwargs.get('sclListSize', 8)
        self.useMinsum = kwargs.get('useMinsum', True)      # the reference always decodes with min-sum (:963)
        if not 1 <= self.sclListSize <= 8:
            raise NotImplementedError("sclListSize must be between 1 and 8")
        super().__init__(payloadSize, rateMatchedLen, dataType, **kwargs)

    def print(self, indent=0, title=None, getStr=False):
        text = super().print(indent, "Polar Decoder Properties:" if title is None else title, True)
        pad = indent * ' '
        text += pad + "  SCL List Size .................: %s\n" % (self.sclListSize)
        text += pad + "  Min-sum Approximation .........: %s\n" % ("Enabled" if self.useMinsum else "Disabled")
        if getStr:
            return text
        print(text)

    def _finishInit(self):
        # The decoder keeps the inverse permutations (reference polar.py:866-879).
        if self.inInterleaveIndexes is not None:
            self.inInterleaveIndexes = np.argsort(self.inInterleaveIndexes)
        self.sbInterleaveIndexes = np.argsort(self.sbInterleaveIndexes)
        if self.cbInterleaveIndexes is not None:
            self.cbInterleaveIndexes = np.argsort(self.cbInterleaveIndexes)

    def recoverRate(self, rxBlock):
        """(C, E) LLRs -> (C, N) LLRs of the mother code word."""
        rxBlock = np.asarray(rxBlock, dtype=np.float64)
        if rxBlock.ndim != 2 or rxBlock.shape[1] != self.rateMatchedBlockLen:
            raise ValueError("rxBlock must be C x %d, got %s" % (self.rateMatchedBlockLen, rxBlock.shape))
        return N(self.recoverRateDevice(D(rxBlock)))

    def recoverRateDevice(self, llr):
        deil = None if self.cbInterleaveIndexes is None else self._d('deil', lambda: np.int32(self.cbInterleaveIndexes))
        return ops.polar_rate_recover(llr, self.polarCodeSize, self.codeBlockSize,
                                      self._d('isb', lambda: np.int32(self.sbInterleaveIndexes)), deil)

    def _sclTables(self):
        NN = self.polarCodeSize
        mask = np.ones(NN, dtype=np.uint8)
        mask[self.frozenBits] = 0
        leafRank = np.cumsum(mask) - 1                      # position of a non-frozen leaf among the non-frozen
        nInfo = int(mask.sum())
        # rate-0 nodes: bits 1..4 of a frozen leaf's byte = log2 of the largest aligned all-frozen block starting there
        # (the kernel expands such a block in one go instead of leaf by leaf; see include/nrx.h)
        kinds = mask.copy()
        i = 0
        while i < NN:
            if mask[i]:
                i += 1
                continue
            s = 0
            while s + 1 < int(np.log2(NN)) and i % (2 << s) == 0 and i + (2 << s) <= NN and not mask[i:i + (2 << s)].any():
                s += 1
            kinds[i] = s << 1
            i += 1 << s
        msg = np.int64(self.msgBits)
        if self.inInterleaveIndexes is not None:
            msg = msg[self.inInterleaveIndexes]
        return (self._d('mask', lambda: kinds), nInfo, self._d('msrc', lambda: np.int32(leafRank[msg])))

    def decodeDevice(self, llr, wantCandidates=False, crcExpect=None):
        """(n, N) float64 device LLRs -> message bits incl. CRC (n, K), CRC flags (n,) [, candidates, path costs].
        ``crcExpect``: per-row CRC register value of a passing candidate (masked CRCs, see neoradium_amd.pdcch)."""
        mask, nInfo, msrc = self._sclTables()
        return ops.polar_scl_decode(llr, mask, nInfo, msrc, self.sclListSize, self.crcPoly, wantCandidates, crcExpect)

    def decode(self, rxLlrBlocks):
        """(C, N) LLRs of one transport block -> (payload bits, number of code blocks whose CRC failed)."""
        rxLlrBlocks = np.asarray(rxLlrBlocks, dtype=np.float64)
        if rxLlrBlocks.ndim != 2 or rxLlrBlocks.shape[1] != self.polarCodeSize:
            raise ValueError("The rxLLRs's second dimension(%d) must match the configured Polar Code Size(%d)" %
                             (rxLlrBlocks.shape[-1], self.polarCodeSize))
        msg, ok = self.decodeDevice(D(rxLlrBlocks))
        msg, ok = N(msg).astype(np.int8), N(ok)
        crcLen = 0 if self.crcPoly is None else self.getCrcLen(self.crcPoly)
        payload = msg[:, :msg.shape[1] - crcLen].reshape(-1)
        return payload[-self.payloadSize:], int((ok == 0).sum())

    def decodeCandidates(self, rxLlrBlocks):
        """Blind decoding: every row is an independent candidate of this (A, E) format.

        Returns (bits (n, A), crcOk (n,)): what ``decode`` would return for each row on its own (iSeg=False only)."""
        if self.iSeg:
            raise ValueError("decodeCandidates needs single-block code words (iSeg=False)")
        rxLlrBlocks = np.asarray(rxLlrBlocks, dtype=np.float64)
        msg, ok = self.decodeDevice(D(rxLlrBlocks))
        crcLen = 0 if self.crcPoly is None else self.getCrcLen(self.crcPoly)
        msg = N(msg).astype(np.int8)
        return msg[:, :msg.shape[1] - crcLen], N(ok).astype(bool)
