"""TR 38.901 / TS 38.101-4 tapped-delay-line channels TDL-A..E, A30, B100, C60, C300, D30 (reference tdl.py).

The sum-of-sinusoids fading (GMEDS-1 by default), the MIMO correlation matrix, the LOS tap and the tap powers
are folded on the host into the same static form the CDL uses -- complex coefficients x complex exponentials --
(cos x = (e^{jx}+e^{-jx})/2), so the per-slot gains run on the same GPU kernel.
"""
import os

import numpy as np
from scipy.linalg import sqrtm

from .channelmodel import ChannelModel
from .random import random
from .utils import toDb, toLinear

_TABLES = None
_DELAY_SPREADS = {"VeryShort": 10, "Short": 30, "Nominal": 100, "Long": 300, "VeryLong": 1000}
# (alpha, beta, gamma) of TS 38.101-4 Tables B.2.3.1.2-1 / B.2.3.2.2-1 and TS 38.104 Tables G.2.3.1.2-1 / G.2.3.2.3-1
_ABG = {('Downlink', 'CoPolar'): {'High': (0.9, 0.9, 0), 'Medium': (0.3, 0.9, 0), 'MediumA': (0.3, 0.3874, 0), 'Low': (0.0, 0.0, 0)},
        ('Downlink', 'CrossPolar'): {'High': (0.9, 0.9, 0.3), 'Medium': (0.3, 0.6, 0.2)},
        ('Uplink', 'CoPolar'): {'High': (0.9, 0.9, 0), 'Medium': (0.9, 0.3, 0), 'Low': (0.0, 0.0, 0)},
        ('Uplink', 'CrossPolar'): {'Low': (0.0, 0.0, 0.0)}}
# positive-semi-definite correction factors "a" (TS 38.101-4 B.2.3.1.2 / B.2.3.2.2, TS 38.104 G.2.3.1.2)
_PSD = {('Downlink', 'CoPolar', 'High', '4x2'): 1.0e-4, ('Downlink', 'CoPolar', 'High', '4x4'): 1.2e-4,
        ('Downlink', 'CoPolar', 'Medium', '2x4'): 1.0e-4, ('Downlink', 'CoPolar', 'Medium', '4x4'): 1.2e-4,
        ('Downlink', 'CrossPolar', 'High', '8x2'): 1.0e-4,
        ('Uplink', 'CoPolar', 'High', '2x4'): 1.0e-4, ('Uplink', 'CoPolar', 'High', '4x4'): 1.2e-4,
        ('Uplink', 'CoPolar', 'Medium', '4x4'): 1.2e-4}


def _tables():
    global _TABLES
    if _TABLES is None:
        _TABLES = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'channel_tables.npz'))
    return _TABLES


class TdlChannel(ChannelModel):
    def __init__(self, bwp, profile='A', **kwargs):
        super().__init__(bwp, **kwargs)
        combos = ['A30-5', 'A30-10', 'B100-400', 'C300-100', 'C300-600', 'C300-1200',       # TS 38.101-4 Table B.2.2-1
                  'A30-35', 'A30-75', 'A30-300', 'C60-300', 'D30-75']                       # Table B.2.2-2
        if profile in combos:
            profile, dop = profile.split('-')
            self.dopplerShift = int(dop)
        if profile not in ['A', 'B', 'C', 'D', 'E', 'A30', 'B100', 'C60', 'C300', 'D30']:
            raise ValueError("Unsupported delay profile \"%s\"!" % (profile))
        self.profile = profile
        self.delaySpread = kwargs.get('delaySpread', 30)     # (the reference overrides the profile's value too)
        if isinstance(self.delaySpread, str):
            if self.delaySpread not in _DELAY_SPREADS:
                raise ValueError("'delaySpread' must be a number or one of 'VeryShort', 'Short', 'Nominal', 'Long', "
                                 "or 'VeryLong'")
            self.delaySpread = _DELAY_SPREADS[self.delaySpread]
        self.txAntennaCount = kwargs.get('txAntennaCount', 1)
        self.rxAntennaCount = kwargs.get('rxAntennaCount', 1)
        self.kFactor = kwargs.get('kFactor', None)
        tab = np.float64(_tables()['tdl_' + profile])
        self.pathDelays = np.float64(kwargs.get('pathDelays', tab[:, 0].copy()))
        self.pathPowers = np.float64(kwargs.get('pathPowers', tab[:, 1].copy()))
        self.hasLos = profile[0] in "DE"
        self.kFactorLos = kwargs.get('kFactorLos', (self.pathPowers[0] - self.pathPowers[1]) if self.hasLos else None)
        if len(self.pathDelays) != len(self.pathPowers):
            raise ValueError("Tap delays and powers must have the same size!")
        self.scaleDelays()
        if self.kFactor is not None:
            self.applyKFactorScaling()
        if self.hasLos:
            self.losDopplerShift = 0.7 * self.dopplerShift                       # TR 38.901 7.7.2
            self.pathPowers = np.concatenate(([toDb(toLinear(self.pathPowers[:2]).sum())], self.pathPowers[2:]))
            self.pathDelays = np.concatenate((self.pathDelays[0:1], self.pathDelays[2:]))
        self.numPaths = len(self.pathDelays)
        self.mimoCorrelation = kwargs.get('mimoCorrelation', 'Low')
        if self.mimoCorrelation not in ['Low', 'Medium', 'MediumA', 'MediumB', 'High']:
            raise ValueError("Unsupported 'mimoCorrelation' (%s)." % (self.mimoCorrelation))
        self.polarization = kwargs.get('polarization', 'CoPolar')
        if self.polarization not in ['CoPolar', 'CrossPolar']:
            raise ValueError(f"Unsupported 'polarization' ({self.polarization}). It must be 'CoPolar' or 'CrossPolar'.")
        self.correlationMatrix = kwargs.get('correlationMatrix', None)
        if self.correlationMatrix is None:
            self.correlationMatrix = self.getSpatialCorrelationMatrix()
        self.sosType = kwargs.get('sosType', 'GMEDS1')
        if self.sosType not in ['GMEDS1', 'Xiao']:
            raise ValueError(f"Unsupported 'sosType' ({self.sosType}). It must 'GMEDS1' or 'Xiao'.")
        # 'Xiao' is a statistical model: new random angles and phases for every slot (tdl.py:1043-1067), i.e. the "static"
        # coefficients are rebuilt -- and the generator advanced -- whenever a slot is prepared
        self._static_per_slot = self.sosType == 'Xiao'
        self.sosNumSins = kwargs.get('sosNumSins', 32)
        nr, nt = self.nrNt
        self.sosTheta1N = self.rangen.random(size=(1, self.sosNumSins, nr, nt, self.numPaths)) * 2 * np.pi
        self.sosTheta2N = self.rangen.random(size=(1, self.sosNumSins, nr, nt, self.numPaths)) * 2 * np.pi
        self.restart()

    def restart(self, restartRanGen=False, applyToBwp=True):
        if (self.seed is not None) and restartRanGen:
            self.rangen = random.getGenerator(self.seed)
        # 'Xiao': the generator as it stands before slot 0's draws -- what staticCoefficientsAt() reproduces any slot's draws from
        self._slot0_state = self._rangenState() if self._static_per_slot else None
        self._legacy_run = None
        super().restart(restartRanGen, applyToBwp)

    def _rangenState(self):
        g = getattr(self.rangen, 'generator', self.rangen)
        return ('legacy', g.get_state()) if isinstance(g, np.random.RandomState) else ('pcg', g.bit_generator.state)

    def staticCoefficientsAt(self, slot):
        """'Xiao' only: the coefficients the slot-by-slot class surface uses for the ``slot``-th slot after the last restart()
        (tdl.py:1043-1067 draws one (theta, phi) pair per prepared slot), without touching the channel's own generator:
        a copy of the generator as it stood before slot 0 is moved ``slot`` pairs of draws forward.  This is what lets the
        batched engine (PdschLink) prepare any range of slots at once, on any rank."""
        if not self._static_per_slot:
            raise ValueError("staticCoefficientsAt: only the per-slot statistical model (sosType='Xiao') has per-slot coefficients")
        if self.seed is None:
            # without its own seed the channel draws from the package-wide generator it shares with the bits and the noise: the
            # number of draws between two slots is then not this channel's alone and a slot's coefficients cannot be reproduced
            raise ValueError("staticCoefficientsAt (sosType='Xiao' through PdschLink): the channel needs its own seed (seed=...)")
        nr, nt = self.nrNt
        per_slot = self.sosNumSins * self.numPaths * (1 + nr * nt)       # doubles drawn per slot
        kind, st = self._slot0_state
        if kind == 'pcg':
            src = getattr(self.rangen, 'generator', self.rangen).bit_generator
            bg = type(src)()                                              # the same kind of bit generator as the channel's own
            if not hasattr(bg, 'advance'):
                raise ValueError(f"staticCoefficientsAt: the channel's bit generator ({type(src).__name__}) cannot jump ahead")
            bg.state = st
            bg.advance(int(slot) * per_slot)                              # Generator.random(): one 64-bit output per double
            g = np.random.Generator(bg)
        else:
            # legacy RandomState cannot jump: a running copy is kept and moved forward (slots are asked for in ascending order by
            # the engine; going back restarts from the slot-0 state)
            cache = getattr(self, '_legacy_run', None)
            if cache is None or cache[0] > int(slot):
                g0 = np.random.RandomState()
                g0.set_state(st)
                cache = [0, g0]
            while cache[0] < int(slot):
                cache[1].random_sample(per_slot)
                cache[0] += 1
            self._legacy_run = cache
            g = np.random.RandomState()
            g.set_state(cache[1].get_state())
        return self.staticCoefficients(gen=g)

    @property
    def nrNt(self):
        return self.rxAntennaCount, self.txAntennaCount

    def scaleDelays(self):
        if self.profile in "ABCDE":
            self.pathDelays *= self.delaySpread

    def staticCoefficients(self, gen=None):
        """GMEDS-1 sum of sinusoids (tdl.py:1070-1088) as 4N complex exponentials per (r,t,path) + one LOS ray.

        QUIRKS kept: the discrete Doppler 'frequencies' already carry a 2*pi and get another one in the phase
        (tdl.py:1084-1088); the LOS tap uses sqrt(k) of an amplitude that is already a square root (tdl.py:1117-1120)."""
        nr, nt = self.nrNt
        N_, P = self.sosNumSins, self.numPaths
        if self.sosType == 'Xiao':
            # tdl.py:1043-1067 (Xiao et al., "Novel sum-of-sinusoids simulation models ...", eq. 6-7): N unit phasors per
            # (r,t,path), Doppler f_D cos(alpha_n) with alpha_n = (2 pi n + theta_n) / N; theta (per path) and phi (per
            # r,t,path) are drawn from the channel's generator in the reference's order, one pair of draws per call
            gen = self.rangen if gen is None else gen
            draw = gen.random if hasattr(gen, 'random') else gen.random_sample
            theta = draw(size=(1, N_, 1, 1, P)) * 2 * np.pi - np.pi
            phi = draw(size=(1, N_, nr, nt, P)) * 2 * np.pi - np.pi
            alpha = (2 * np.pi * (np.arange(N_, dtype=np.float64).reshape(1, -1, 1, 1, 1) + 1) + theta) / N_
            M = N_ + 1                                                                      # + one LOS ray
            A = np.zeros((nr, nt, P, M), dtype=np.complex128)
            nu = np.zeros((P, M))
            A[..., :N_] = np.sqrt(1 / N_) * np.exp(1j * np.transpose(phi[0], (1, 2, 3, 0)))
            nu[:, :N_] = (self.dopplerShift * np.cos(alpha[0, :, 0, 0, :])).T
            los_col = N_
        else:
          a_n = np.pi * (np.arange(N_, dtype=np.float64) + .5) / (2 * N_)
          a_0 = np.pi * (np.arange(P, dtype=np.float64) + 1) / (4 * N_ * (P + 2))
          f1 = 2 * np.pi * self.dopplerShift * np.cos(a_n[:, None] + a_0[None, :])       # (N, P)
          f2 = 2 * np.pi * self.dopplerShift * np.cos(a_n[:, None] - a_0[None, :])
          th1 = self.sosTheta1N[0]                                                        # (N, nr, nt, P)
          th2 = self.sosTheta2N[0]
          amp = np.sqrt(2 / N_)
          # rays: [ +f1 | -f1 | +f2 | -f2 | LOS ]
          M = 4 * N_ + 1
          A = np.zeros((nr, nt, P, M), dtype=np.complex128)
          nu = np.zeros((P, M))
          t1 = np.transpose(th1, (1, 2, 3, 0))                                            # (nr, nt, P, N)
          t2 = np.transpose(th2, (1, 2, 3, 0))
          A[..., 0 * N_:1 * N_] = 0.5 * amp * np.exp(1j * t1)
          A[..., 1 * N_:2 * N_] = 0.5 * amp * np.exp(-1j * t1)
          A[..., 2 * N_:3 * N_] = 0.5j * amp * np.exp(1j * t2)
          A[..., 3 * N_:4 * N_] = 0.5j * amp * np.exp(-1j * t2)
          nu[:, 0 * N_:1 * N_], nu[:, 1 * N_:2 * N_] = f1.T, -f1.T
          nu[:, 2 * N_:3 * N_], nu[:, 3 * N_:4 * N_] = f2.T, -f2.T
          los_col = 4 * N_
        if not np.isscalar(self.correlationMatrix):                                     # tdl.py:1101-1110
            cm = self.correlationMatrix
            if self.normalizeGains:
                cm = cm * nt * nr / np.trace(cm)
            sq = sqrtm(cm)
            A = np.einsum('apm,ab->bpm', A.reshape(nr * nt, P, M), sq).reshape(nr, nt, P, M)
        if self.hasLos:
            k1 = np.sqrt(toLinear(self.kFactorLos))
            A[:, :, 0, :] /= np.sqrt(k1 + 1)
            A[:, :, 0, los_col] = np.sqrt(k1) / np.sqrt(k1 + 1)
            nu[0, los_col] = self.losDopplerShift
        A = A * np.sqrt(toLinear(self.pathPowers)).reshape(1, 1, -1, 1)
        return A, nu, None, 0.0

    # ------------------------------------------------------------------------------- MIMO correlation matrices
    def getSpatialCorrelationMatrix(self):
        """TS 38.101-4 B.2.3 / TS 38.104 G.2.3 spatial correlation (tdl.py:1129-1198); a scalar n means I_n."""
        nr, nt = self.nrNt
        if nt * nr <= 1:
            return 1
        ng, nu = (nt, nr) if self.txDir == 'Downlink' else (nr, nt)
        if self.polarization == 'CrossPolar':
            ng, nu = ng // 2, nu // 2
        try:
            alpha, beta, gamma = _ABG[(self.txDir, self.polarization)][self.mimoCorrelation]
        except KeyError:
            assert 0, f"The combination '{self.txDir}, {self.polarization}, {self.mimoCorrelation}' is not supported!"

        def ula(n, rho):
            if n == 1:
                return 1
            if rho == 0:
                return n
            i = np.arange(n)
            return rho ** np.square((i[:, None] - i[None, :]) / (n - 1))

        gnb, ue = ula(ng, alpha), ula(nu, beta)
        eye = lambda v: np.eye(v) if np.isscalar(v) else v                                # noqa: E731
        if self.polarization == 'CrossPolar':
            pp = self.getPermutationMatrix()
            if self.txDir == 'Downlink':
                gg = np.float64([[1, 0, -gamma, 0], [0, 1, 0, gamma], [-gamma, 0, 1, 0], [0, gamma, 0, 1]])
                r = pp.dot(np.kron(np.kron(eye(gnb), gg), eye(ue))).dot(pp.T)
            else:
                gg = np.float64([[1, -gamma], [-gamma, 1]]) if nu == 1 else \
                    np.float64([[1, -gamma, 0, 0], [-gamma, 1, 0, 0], [0, 0, 1, gamma], [0, 0, gamma, 1]])
                r = pp.dot(np.kron(np.kron(eye(ue), gg), eye(gnb))).dot(pp.T)
        elif np.isscalar(ue) and np.isscalar(gnb):
            r = ue * gnb
        else:
            r = np.kron(eye(gnb), eye(ue)) if self.txDir == 'Downlink' else np.kron(eye(ue), eye(gnb))
        return self.ensurePSD(r)

    def ensurePSD(self, rSpat):
        nr, nt = self.nrNt
        a = _PSD.get((self.txDir, self.polarization, self.mimoCorrelation, "%dx%d" % (nt, nr)), 0)
        if a > 0:
            if np.isscalar(rSpat):
                rSpat = np.eye(rSpat)
            return (rSpat + a * np.eye(nt * nr)) / (1.0 + a)
        return rSpat

    def getPermutationMatrix(self):
        """TS 38.101-4 B.2.3.2.1 permutation matrix P (tdl.py:1223-1237)."""
        assert self.polarization == 'CrossPolar', "The permutation Matrix is only used for the 'CrossPolar' polarization mode!"
        nr, nt = self.nrNt
        pp = np.zeros((nt * nr, nt * nr))
        for j in range(nt // 2):
            for i in range(nr):
                pp[j * nr + i, 2 * j * nr + i] = 1
                pp[(j + nt // 2) * nr + i, (2 * j + 1) * nr + i] = 1
        return pp

    def print(self, indent=0, title=None, getStr=False):
        s = super().print(indent, f"TDL-{self.profile} Channel Properties:" if title is None else title, True)
        pad = indent * ' '
        s += pad + f"  delaySpread:      {self.delaySpread} ns\n" + pad + f"  hasLOS:           {self.hasLos}\n"
        s += pad + f"  rxAntennaCount:   {self.rxAntennaCount}\n" + pad + f"  txAntennaCount:   {self.txAntennaCount}\n"
        s += pad + f"  mimoCorrelation:  {self.mimoCorrelation}\n" + pad + f"  polarization:     {self.polarization}\n"
        if getStr:
            return s
        print(s)
