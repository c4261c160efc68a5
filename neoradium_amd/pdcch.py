"""PDCCH candidate layer: DCI encoding and batched blind decoding on top of the polar control-channel codec.

The reference has no PDCCH (neoradium/dmrs.py:199-203 implements the PDSCH only; its control-channel code is the polar codec
neoradium/polar.py).  This layer is what BASELINE cfg4 ("batched DCI blind-decode candidates") needs above
``PolarDecoder.decodeCandidates``; it follows the specifications directly:

  TS 38.212 7.3.2        CRC attachment: CRC24C over [24 ones, payload], last 16 parity bits XORed with the RNTI
  TS 38.212 7.3.3-7.3.4  polar coding / rate matching to E = 108 * L bits (L CCEs of 6 REGs x 9 data REs x 2 bits)
  TS 38.211 7.3.2.3-4    scrambling with c_init = (n_RNTI * 2^16 + n_ID) mod 2^31, QPSK
  TS 38.213 10.1         a candidate = (aggregation level L, first CCE); the CCE-to-REG mapping is the non-interleaved one,
                         i.e. the CORESET's equalised data symbols are taken in CCE order (54 symbols per CCE)

Everything runs on the device in a handful of launches for ALL candidates of ALL monitoring occasions in the batch: one
demap + descramble (``nrx_qam_demap``), one rate recovery and one SCL decode per aggregation level.  The RNTI mask and the
24 ones are folded into the decoder's CRC test through ``crc_expect`` (the CRC is linear: a mask on the parity bits moves
the end value of the CRC register from 0 to the register value of the mask alone) -- so the list entries are tested against
the RNTI INSIDE the SCL kernel, like a plain CRC.
"""
import numpy as np
import torch

from . import ops
from ._dev import D, device as _device
from .polar import PolarEncoder, PolarDecoder
from .utils import goldBits

BITS_PER_CCE = 108
_POLY24C = 0x1B2B117


def _crc24c_bits(bits):
    """CRC24C parity of a short host bit string (chancodebase.py:83-128 convention) as a 24-bit integer."""
    reg = 0
    for b in bits:
        top = ((reg >> 23) & 1) ^ int(b)
        reg = ((reg << 1) & 0xFFFFFF) ^ ((_POLY24C & 0xFFFFFF) if top else 0)
    return reg


class PDCCH:
    def __init__(self, numCces, nID=1, rnti=1, sclListSize=8):
        if numCces < 1:
            raise ValueError("numCces must be positive")
        if not 0 <= int(rnti) < 65536 or not 0 <= int(nID) < 65536:
            raise ValueError("rnti and nID are 16-bit values")
        self.numCces, self.nID, self.rnti, self.sclListSize = int(numCces), int(nID), int(rnti), int(sclListSize)
        self._codecs = {}

    # ------------------------------------------------------------------------------------------------ helpers
    def codec(self, A, aggLevel):
        key = (int(A), int(aggLevel))
        if key not in self._codecs:
            E = BITS_PER_CCE * int(aggLevel)
            self._codecs[key] = (PolarEncoder(A, E, 'dci'), PolarDecoder(A, E, 'dci', sclListSize=self.sclListSize))
        return self._codecs[key]

    def _cinit(self, rnti):
        return (int(rnti) * 65536 + self.nID) % (1 << 31)

    @staticmethod
    def parityMask(A, rnti):
        """24-bit mask XORed onto plain CRC24C(payload) by TS 38.212 7.3.2: the contribution of the 24 prepended ones
        (CRC of [ones, zeros(A)], by linearity) and the RNTI on the last 16 bits."""
        return _crc24c_bits([1] * 24 + [0] * int(A)) ^ (int(rnti) & 0xFFFF)

    @staticmethod
    def crcExpect(A, rnti):
        """CRC register value at which a correctly decoded [payload, masked parity] word ends (0 for an unmasked CRC)."""
        m = PDCCH.parityMask(A, rnti)
        return _crc24c_bits([(m >> (23 - i)) & 1 for i in range(24)])

    def candidates(self, aggLevels=(1, 2, 4, 8, 16)):
        """All aligned candidates of the CORESET: [(L, firstCce)], first CCE a multiple of L (TS 38.213 10.1 places the
        candidates of a search-space set on such positions; which of them a UE monitors is configuration)."""
        return [(int(L), c) for L in aggLevels if L <= self.numCces for c in range(0, self.numCces - L + 1, L)]

    # -------------------------------------------------------------------------------------------------- encode
    def dciEncode(self, payload, aggLevel, rnti=None):
        """(n, A) payload bits (NumPy or device tensor) -> (n, 108 * aggLevel) coded bits on the device."""
        rnti = self.rnti if rnti is None else int(rnti)
        a = payload if torch.is_tensor(payload) else D(np.uint8(np.atleast_2d(payload)))
        a = a.to(torch.uint8).contiguous()
        A = a.shape[1]
        enc, _ = self.codec(A, aggLevel)
        par = ops.crc(a, '24C')                                         # plain CRC24C(payload) ...
        mask = self.parityMask(A, rnti)                                 # ... + the ones prefix and the RNTI (linear)
        mbits = torch.tensor([(mask >> (23 - i)) & 1 for i in range(24)], dtype=torch.uint8, device=a.device)
        cbs = torch.cat([a, par ^ mbits[None, :]], dim=1).contiguous()
        return enc.rateMatchDevice(enc.encodeDevice(cbs))

    def encode(self, payload, aggLevel, rnti=None):
        """DCI payloads -> scrambled QPSK symbols (n, 54 * aggLevel) complex128 on the device."""
        rnti = self.rnti if rnti is None else int(rnti)
        coded = self.dciEncode(payload, aggLevel, rnti)
        scr = D(goldBits(self._cinit(rnti), coded.shape[1]).astype(np.uint8))
        return ops.qam_map(coded, 2, scr=scr)

    # ------------------------------------------------------------------------------------------- blind decoding
    def blindDecode(self, symbols, noiseVar, A, rnti=None, candidates=None):
        """symbols: (n, numCces * 54) equalised QPSK symbols of n monitoring occasions (device complex128 or NumPy);
        noiseVar: scalar or (n,).  Every candidate of every occasion is decoded for ``rnti``.
        Returns (found (n, nCand) bool, bits (n, nCand, A) uint8, candidates): device tensors + the candidate list."""
        rnti = self.rnti if rnti is None else int(rnti)
        sym = symbols if torch.is_tensor(symbols) else D(np.atleast_2d(np.complex128(symbols)))
        sym = sym.to(torch.complex128).contiguous()
        n = sym.shape[0]
        if sym.shape[1] != self.numCces * 54:
            raise ValueError(f"symbols must be (n, {self.numCces * 54}) for a CORESET of {self.numCces} CCEs")
        cands = self.candidates() if candidates is None else [(int(L), int(c)) for L, c in candidates]
        dev = sym.device
        nv = torch.as_tensor(noiseVar, dtype=torch.float64, device=dev).reshape(-1)
        if nv.numel() not in (1, n):
            raise ValueError("noiseVar must be a scalar or one value per monitoring occasion")
        found = torch.zeros((n, len(cands)), dtype=torch.bool, device=dev)
        bits = torch.zeros((n, len(cands), int(A)), dtype=torch.uint8, device=dev)
        expect = self.crcExpect(A, rnti)
        for L in sorted({L for L, _ in cands}):
            idx = [i for i, (l, _) in enumerate(cands) if l == L]
            starts = [cands[i][1] for i in idx]
            if any(c < 0 or c + L > self.numCces for c in starts):
                raise ValueError("candidate outside the CORESET")
            E = BITS_PER_CCE * L
            _, dec = self.codec(A, L)
            win = torch.stack([sym[:, c * 54:(c + L) * 54] for c in starts], dim=1).reshape(n * len(idx), 54 * L).contiguous()
            nvw = (nv if nv.numel() == 1 else nv.repeat_interleave(len(idx))).contiguous()
            scr = D(goldBits(self._cinit(rnti), E).astype(np.uint8))
            llr = ops.qam_demap(win, nvw, 2, scr=scr, llr_dtype=torch.float64)          # max-log LLRs, descrambled
            rr = dec.recoverRateDevice(llr)
            ce = torch.full((rr.shape[0],), expect, dtype=torch.int32, device=dev)
            msg, ok = dec.decodeDevice(rr, crcExpect=ce)
            cols = torch.as_tensor(idx, device=dev)
            found[:, cols] = ok.reshape(n, len(idx)).to(torch.bool)
            bits[:, cols] = msg[:, :int(A)].reshape(n, len(idx), int(A))
        return found, bits, cands
