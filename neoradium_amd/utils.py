"""Small host-side helpers with the names the NeoRadium notebooks import from ``neoradium.utils``."""
import functools
import warnings

import numpy as np

from . import _lib


def toRadian(angle):
    return None if angle is None else np.float64(angle) * np.pi / 180.0


def toDegrees(angle):
    return None if angle is None else np.float64(angle) * 180.0 / np.pi


def toLinear(x):
    return 10.0 ** (x / 10.0)


def toDb(x):
    return 10.0 * np.log10(x)


def herm(x):
    """Conjugate transpose of the last two axes."""
    return np.conj(np.swapaxes(x, -1, -2))


def getMse(h, hEst):
    return np.square(np.abs(hEst - h)).mean()


def getNmse(u, uEst):
    return np.square(np.abs(uEst - u)).sum() / np.square(np.abs(u.mean() - u)).sum()


def goldSequence(cInit, numBits):
    """TS 38.211 5.2.1 Gold sequence (reference utils.py:70-94); returns a list of bits like the reference."""
    return _lib.gold_sequence(cInit, numBits).tolist()


def goldBits(cInit, numBits):
    """Same sequence as an int8 NumPy array (what this package uses internally)."""
    return _lib.gold_sequence(cInit, numBits)


def intToBits(n, length=None):
    bits = [int(c) for c in bin(n)[2:]]
    if length is not None:
        bits = [0] * (length - len(bits)) + bits
    return np.uint8(bits)


def freqStr(f):
    for lim, div, unit in ((1e12, 1e12, 'THz'), (1e9, 1e9, 'GHz'), (1e6, 1e6, 'MHz'), (1e3, 1e3, 'kHz')):
        if f > lim and f <= 1e15:
            return f"{f / div:.4g} {unit}"
    return f"{f:.4g} Hz" if f > 1e15 else f"{f} Hz"


def getMultiLineStr(label, values, indent, formatStr, length, numPerLine):
    """Multi-line value list used by the ``print`` methods."""
    head = indent * ' ' + '  ' + label.rstrip() + ':' + ' ' * (len(label) - len(label.rstrip()))
    pad = indent * ' ' + '  ' + ' ' * (len(label.rstrip()) + 1 + len(label) - len(label.rstrip()))
    out = ""
    for r in range(0, len(values), numPerLine):
        row = " ".join((formatStr % v)[:length] for v in values[r:r + numPerLine])
        out += (head if r == 0 else pad) + " " + row + "\n"
    return out


def deprecated(replacement=None):
    def deco(func):
        @functools.wraps(func)
        def wrapper(*a, **k):
            msg = f"Call to deprecated function {func.__name__}."
            if replacement:
                msg += f" Use {replacement} instead."
            warnings.warn(msg, category=DeprecationWarning, stacklevel=2)
            return func(*a, **k)
        return wrapper
    return deco
