"""Base of the channel-coding classes: CRC on the GPU (reference chancodebase.py:48-189)."""
import numpy as np

from . import ops
from ._dev import D, N

CRC_LENGTHS = {'6': 6, '11': 11, '16': 16, '24A': 24, '24B': 24, '24C': 24}


class ChanCodeBase:
    LARGE_LLR = 1e20        # LLR of a known (filler / frozen) bit

    def __init__(self):
        pass

    @classmethod
    def getCrcLen(cls, poly):
        return 24 if poly[:2] == "24" else int(poly)

    @classmethod
    def getCrc(cls, bits, poly):
        """CRC remainder (TS 38.212 5.1) of a bit vector or of every row of a bit matrix."""
        if poly not in CRC_LENGTHS:
            raise ValueError("Unsupported CRC polynomial '%s'" % (poly))
        bits = np.asarray(bits)
        flat = bits.ndim == 1
        b = np.uint8(bits.reshape(1, -1) if flat else bits)
        out = N(ops.crc(D(b), poly)).astype(np.int8)
        return out[0] if flat else out

    @classmethod
    def checkCrc(cls, bits, poly):
        return np.count_nonzero(cls.getCrc(bits, poly), -1) == 0

    @classmethod
    def appendCrc(cls, bits, poly):
        bits = np.asarray(bits)
        return np.concatenate([np.int8(bits), cls.getCrc(bits, poly)], axis=-1)
