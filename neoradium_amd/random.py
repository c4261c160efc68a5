"""The package-wide random generator object ``random`` (reference random.py:190-318).

The draw ORDER is the reproducibility contract with the reference (bits -> channel phases -> noise), so this is a
thin layer over NumPy's own ``Generator(PCG64)`` / ``RandomState`` exactly like the reference: ``bits`` and ``awgn``
are the two additions.  Device-side counter-based noise (ops.awgn) is used only by the throughput engine.
"""
import numpy as np


class _LegacyGen(np.random.RandomState):
    """RandomState flavour (Matlab-compatible streams, reference random.py:190-195)."""

    def integers(self, low, high=None, size=None, dtype=np.int64):
        return self.randint(low, high, size, dtype)

    def bits(self, size):
        return self.randint(0, 2, size, dtype=np.int8)

    def awgn(self, shape, noiseStd):
        pair = self.normal(0, noiseStd / np.sqrt(2), tuple(shape) + (2,))
        return (pair * [1, 1j]).sum(-1)


class _Gen(np.random.Generator):
    """Generator flavour (default, PCG64; reference random.py:198-203)."""

    def randint(self, low, high=None, size=None, dtype=int):
        return self.integers(low, high, size, dtype)

    def bits(self, size):
        return self.integers(0, 2, size, dtype=np.int8)

    def awgn(self, shape, noiseStd):
        pair = self.normal(0, noiseStd / np.sqrt(2), tuple(shape) + (2,))
        return (pair * [1, 1j]).sum(-1)


class RanGen:
    """Holder of the active generator; attribute access falls through to it (random.py:206-316)."""

    def __init__(self, generator=None):
        self.generator = self._make(None) if generator is None else generator

    @staticmethod
    def _make(seed):
        if seed is None:
            return _Gen(np.random.PCG64())
        if isinstance(seed, np.random.BitGenerator):
            return _Gen(seed)
        if isinstance(seed, np.random.Generator):
            return _Gen(seed.bit_generator)
        if isinstance(seed, np.random.RandomState):
            return _LegacyGen(seed.get_state()[1][0])
        return _Gen(np.random.PCG64(seed))

    def getGenerator(self, seed=None):
        if isinstance(seed, RanGen):
            return seed
        return RanGen(self._make(seed))

    def setSeed(self, seed):
        self.generator = self.getGenerator(seed).generator

    def __getattr__(self, name):
        if name == 'generator':
            raise AttributeError(name)
        return getattr(self.generator, name)


random = RanGen()
