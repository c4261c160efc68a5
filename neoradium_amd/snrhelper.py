"""Adaptive SNR walk for BLER/BER curves (reference snrhelper.py:14-254) -- host-side harness logic.

The scheduler hands out SNR points one at a time; after each point the caller reports the metric (e.g. BLER %)
with ``setData``.  Starting from ``snr0`` it first finds the transition region, walks down to the "low" plateau
(metric at ``loSnrVal``), then up to the "high" plateau (metric at ``hiSnrVal``), reusing results already seen.
"""
import numpy as np


class SnrScheduler:
    def __init__(self, snr0=0, step=1, maxSnrs=500, loSnrVal=100, hiSnrVal=0):
        self.snr0 = snr0
        if not (isinstance(step, (int, float)) and step > 0):
            raise ValueError("`step` must be a positive number.")
        self.step = step
        if not (isinstance(maxSnrs, int) and maxSnrs > 0):
            raise ValueError("`maxSnrs` must be a positive integer.")
        self.maxSnrs = maxSnrs
        self.loSnrVal = loSnrVal
        self.hiSnrVal = hiSnrVal
        self.reset()

    def reset(self):
        self.curSnr = self.snr0
        self.buffers = None
        self.state = 'Start'
        self.curLo, self.curHi = -np.inf, np.inf
        self.setDataCalled = True

    def __iter__(self):
        return self

    def __next__(self):
        if self.state == 'Done':
            raise StopIteration
        if not self.setDataCalled:
            raise ValueError("The \"setData\" was not called in the last iteration!")
        self.setDataCalled = False
        return self.curSnr

    def whereAmI(self, value):
        """'LoSNR' / 'HiSNR' plateau or 'MidSNR' (the metric may grow or shrink with SNR)."""
        rising = self.loSnrVal < self.hiSnrVal
        if (value <= self.loSnrVal) if rising else (value >= self.loSnrVal):
            return 'LoSNR'
        if (value >= self.hiSnrVal) if rising else (value <= self.hiSnrVal):
            return 'HiSNR'
        return 'MidSNR'

    def setData(self, value, *otherValues):
        """Report the metric of the SNR handed out last; advances to the next SNR not evaluated yet."""
        self.setDataCalled = True
        row = (self.curSnr, value) + tuple(otherValues)
        if self.buffers is None:
            self.buffers = [[] for _ in row]
        elif len(row) != len(self.buffers):
            raise ValueError("Inconsistent number of values passed to the \"setData\" function!")
        elif len(self.buffers[0]) >= self.maxSnrs:
            raise ValueError(f"Did not converge after {self.maxSnrs} tries.")
        for buf, v in zip(self.buffers, row):
            buf.append(v)
        while self.curSnr in self.buffers[0]:                 # skip points whose result is already known
            self.updateState(self.buffers[1][self.buffers[0].index(self.curSnr)])
            if self.curSnr is None:
                break
            self.curSnr = np.round(self.curSnr, 4).item()

    def _seen_lo(self):
        self.curLo = max(self.curSnr, self.curLo)

    def _seen_hi(self):
        self.curHi = min(self.curSnr, self.curHi)

    def _enter_range(self):
        """An in-range point was found: remember where to resume upwards, walk down first."""
        self.upStart = self.curSnr + self.step
        self.curSnr -= self.step
        self.state = 'GoingDown'

    def updateState(self, value):
        zone, st = self.whereAmI(value), self.state
        if st in ('Start', 'SearchingUp', 'SearchingDown'):
            if zone == 'MidSNR':
                self._enter_range()
            elif zone == 'LoSNR':
                self._seen_lo()
                if st == 'SearchingDown':                      # overshot: bisect between the plateaus
                    self.curSnr = (self.curHi + self.curLo) / 2
                else:
                    self.curSnr += self.step * (2 if st == 'SearchingUp' else 1)
                self.state = 'SearchingUp'
            else:
                self._seen_hi()
                if st == 'SearchingUp':
                    self.curSnr = (self.curHi + self.curLo) / 2
                else:
                    self.curSnr -= self.step * (2 if st == 'SearchingDown' else 1)
                self.state = 'SearchingDown'
        elif st in ('GoingDown', 'AtLow'):
            if zone == 'HiSNR':
                how = "Going down -> HiSNR" if st == 'GoingDown' else "LoSNR -> going down -> HiSNR"
                raise RuntimeError(f"Unexpected state reached in algorithm. ({how}) SNR:{self.curSnr} Value:{value}")
            if zone == 'LoSNR' and st == 'AtLow':              # two low points in a row: the bottom is confirmed
                self.curSnr = self.upStart
                self.state = 'GoingUp'
            else:
                if zone == 'LoSNR':
                    self._seen_lo()
                self.curSnr -= self.step
                self.state = 'AtLow' if zone == 'LoSNR' else 'GoingDown'
        elif st in ('GoingUp', 'AtHigh'):
            if zone == 'LoSNR':
                how = "Going up -> LoSNR" if st == 'GoingUp' else "HiSNR -> going up - LoSNR"
                raise RuntimeError(f"Unexpected state reached in algorithm. ({how}) SNR:{self.curSnr} Value:{value}")
            if zone == 'HiSNR' and st == 'AtHigh':
                self.state = 'Done'
                self.curSnr = None
            else:
                if zone == 'HiSNR':
                    self._seen_hi()
                self.curSnr += self.step
                self.state = 'AtHigh' if zone == 'HiSNR' else 'GoingUp'

    def getSnrsAndData(self):
        """[snrs, metric, others...] sorted by SNR, restricted to the span between the two plateaus."""
        if not self.buffers:
            return [np.array([])]
        snrs = self.buffers[0]
        idx = [i for i in np.argsort(snrs) if self.curLo <= snrs[i] <= self.curHi]
        return [np.array(b)[idx] for b in self.buffers]
