"""HARQ entity / process / codeword bookkeeping (reference harq.py:77-667).

Control logic only (which process transmits, which redundancy version, statistics).  The soft combining itself
happens inside the rate-recovery kernel; the per-codeword soft buffer ``decBuffer`` and the encoded-block cache
``encBuffer`` are kept as device tensors, so retransmissions never leave HBM.
"""
import numpy as np

from . import ops
from ._dev import D, N


class HarqCW:
    """State of one codeword of one HARQ process."""

    def __init__(self, process, cwIdx):
        self.process = process
        self.cwIdx = cwIdx
        self.reset()

    def reset(self):
        self.curTry = 0          # 0 = next transmission carries new data
        self.txBlockNo = 0
        self.rv = 0
        self.encBuffer = None    # device uint8 (C, N): LDPC coded blocks of the current transport block
        self.decBuffer = None    # device float64 (C, Ncb-F): accumulated LLRs (circular buffer without fillers)

    @property
    def needNewData(self):
        return self.curTry == 0

    def getRateMatchedCodeBlocks(self, txBlock, g=None, concatCBs=True):
        enc = self.process.entity.encoder
        if txBlock is None:                                   # retransmission: only a new rate matching
            assert self.curTry > 0 and self.encBuffer is not None
        else:                                                 # new data: CRC + segmentation + encode, cached
            assert self.curTry == 0 and self.encBuffer is None
            tb = np.asarray(txBlock)
            enc.initialize(len(tb) + 24)
            enc.numFillerBits = enc._cfg.F
            self.encBuffer = ops.ldpc_encode(ops.ldpc_segment(D(np.uint8(tb)[None]), enc._cfg, add_tb_crc=True), enc._cfg)
        return enc.rateMatch(self.encBuffer, g, concatCBs, self.rv)

    def decodeLLRs(self, llrs, txBlockSize, numIter):
        dec = self.process.entity.decoder
        rx = dec.recoverRate(llrs, txBlockSize, self)         # accumulates into self.decBuffer (device)
        bits = dec.decode(rx, numIter=numIter)
        tb, crc = dec.checkCrcAndMerge(bits)
        errors = len(crc) - int(np.sum(crc))
        self.update(errors, txBlockSize)
        return tb[:-24], errors

    def update(self, blockErrors, txBlockSize):
        """Statistics + next redundancy version / reset (harq.py:181-202)."""
        ent = self.process.entity
        k = self.curTry
        if k == 0:
            self.txBlockNo = ent.txBlocks[0]
        ent.txBits[k] += txBlockSize
        ent.txBlocks[k] += 1
        if blockErrors == 0:
            ent.rxBits[k] += txBlockSize
            ent.rxBlocks[k] += 1
            ent.handleEvent("RXSUCCESS", self)
            self.reset()
            return
        ent.handleEvent("RXFAILED", self)
        self.curTry += 1
        if self.curTry == ent.maxTries:
            ent.handleEvent("TIMEOUT", self)
            ent.numTimeouts += 1
            self.reset()
        else:
            self.rv = ent.getRV(self.curTry)

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + ("HARQ Codeword Properties:" if title is None else title) + "\n"
        s += pad + f"  curTry:             {self.curTry}\n" + pad + f"  txBlockNo:          {self.txBlockNo}\n"
        s += pad + f"  rv:                 {self.rv}\n"
        if self.encBuffer is not None:
            s += pad + f"  encBuffer Shape:    {tuple(self.encBuffer.shape)}\n"
        if self.decBuffer is not None:
            s += pad + f"  decBuffer Shape:    {tuple(self.decBuffer.shape)}\n"
        if getStr:
            return s
        print(s)


class HarqProcess:
    def __init__(self, entity, id, numCW):
        self.id = id
        self.entity = entity
        self.cws = [HarqCW(self, i) for i in range(numCW)]

    def reset(self):
        for cw in self.cws:
            cw.reset()

    @property
    def needNewData(self):
        return [cw.curTry == 0 for cw in self.cws]

    def getRateMatchedCodeBlocks(self, txBlocks, gs=None, concatCBs=True):
        return [cw.getRateMatchedCodeBlocks(txBlocks[i], None if gs is None else gs[i], concatCBs)
                for i, cw in enumerate(self.cws)]

    def decodeLLRs(self, llrs, txBlockSizes, numIter=5):
        res = [cw.decodeLLRs(llrs[i], txBlockSizes[i], numIter) for i, cw in enumerate(self.cws)]
        return tuple(list(x) for x in zip(*res))

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + ("HARQ Process Properties:" if title is None else title) + "\n"
        s += pad + f"  id:                   {self.id}\n" + pad + f"  numCW:                {len(self.cws)}\n"
        for i, cw in enumerate(self.cws):
            s += cw.print(indent + 2, f"HARQ CW {i + 1}:", True)
        if getStr:
            return s
        print(s)


class HarqEntity:
    def __init__(self, encoder, harqType="CC", numProc=8, rvSequence=[0, 2, 3, 1], maxTries=4, eventCallback=None):
        self.encoder = encoder
        self.decoder = encoder.getDecoder()
        self.numCW = 2 if encoder.txLayers > 4 else 1
        assert harqType in ["CC", "IR"]
        self.harqType = harqType
        assert 0 < numProc <= 32
        self.numProc = numProc
        self.processes = [HarqProcess(self, i, self.numCW) for i in range(numProc)]
        self.rvSequence = rvSequence
        self.maxTries = maxTries
        self.eventCallback = eventCallback
        self.reset()

    def reset(self):
        for p in self.processes:
            p.reset()
        self.curProcIdx = 0
        # per-try counters (int64 here; the reference's int32 overflows beyond 2.1e9 bits)
        self.rxBits = np.zeros(self.maxTries, dtype=np.int64)
        self.txBits = np.zeros(self.maxTries, dtype=np.int64)
        self.rxBlocks = np.zeros(self.maxTries, dtype=np.int64)
        self.txBlocks = np.zeros(self.maxTries, dtype=np.int64)
        self.numTimeouts = 0

    def handleEvent(self, event, process):
        if self.eventCallback is not None:
            self.eventCallback(event, process)

    def getRV(self, tryNum):
        return 0 if self.harqType == "CC" else self.rvSequence[tryNum % len(self.rvSequence)]

    @property
    def totalTxBlocks(self): return self.txBlocks.sum().item()
    @property
    def totalRxBlocks(self): return self.rxBlocks.sum().item()
    @property
    def totalTxBits(self): return self.txBits.sum().item()
    @property
    def totalRxBits(self): return self.rxBits.sum().item()
    @property
    def throughput(self): return self.totalRxBits * 100 / self.totalTxBits
    @property
    def bler(self): return (self.totalTxBlocks - self.totalRxBlocks) * 100 / self.totalTxBlocks
    @property
    def meanTries(self):
        return (((self.rxBlocks * np.arange(self.maxTries)).sum() + self.numTimeouts * self.maxTries) /
                max(self.totalRxBlocks + self.numTimeouts, 1)).item()
    @property
    def curProcess(self): return self.processes[self.curProcIdx]
    @property
    def needNewData(self): return self.curProcess.needNewData

    def __getitem__(self, idx): return self.processes[idx]

    def goNext(self):
        self.curProcIdx = (self.curProcIdx + 1) % self.numProc

    def getRateMatchedCodeBlocks(self, txBlocks, gs=None, concatCBs=True):
        if not isinstance(txBlocks, list):
            return self.curProcess.getRateMatchedCodeBlocks([txBlocks], [gs], concatCBs)[0]
        return self.curProcess.getRateMatchedCodeBlocks(txBlocks, gs, concatCBs)

    def decodeLLRs(self, llrs, txBlockSize, numIter=5):
        if not isinstance(llrs, list):
            r = self.curProcess.decodeLLRs([llrs], [txBlockSize], numIter)
            return r[0][0], r[1][0]
        return self.curProcess.decodeLLRs(llrs, txBlockSize, numIter)

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + ("HARQ Entity Properties:" if title is None else title) + "\n"
        s += pad + f"  HARQ Type:            {self.harqType}\n" + pad + f"  Num. Processes:       {self.numProc}\n"
        s += pad + f"  Num. Codewords:       {self.numCW}\n" + pad + f"  RV sequence:          {self.rvSequence}\n"
        s += pad + f"  maxTries:             {self.maxTries}\n"
        s += self.encoder.print(indent + 2, "Encoder:", True) + self.decoder.print(indent + 2, "Decoder:", True)
        if getStr:
            return s
        print(s)

    def printStats(self, getStr=False):
        s = "\nHARQ Entity Statistics:\n"
        for name, v in (("txBits (per try):  ", self.txBits), ("rxBits (per try):  ", self.rxBits),
                        ("txBlocks (per try):", self.txBlocks), ("rxBlocks (per try):", self.rxBlocks),
                        ("numTimeouts:       ", self.numTimeouts), ("totalTxBlocks:     ", self.totalTxBlocks),
                        ("totalRxBlocks:     ", self.totalRxBlocks), ("totalTxBits:       ", self.totalTxBits),
                        ("totalRxBits:       ", self.totalRxBits)):
            s += f"  {name}   {v}\n"
        s += f"  throughput:           {self.throughput:.2f}%\n  bler:                 {self.bler:.2f}%\n"
        s += f"  Average Num. Retries: {self.meanTries:.2f}%\n"
        if getStr:
            return s
        print(s)
