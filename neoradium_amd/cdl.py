"""TR 38.901 clustered-delay-line channels CDL-A..E (reference cdl.py:165-932).

Everything that does not depend on time -- ray angles, ray coupling, initial phases, antenna field patterns,
array location phasors, cluster powers -- is evaluated here on the host ONCE per channel object into the
coefficient tensor A[Nr,Nt,n,m] and the per-ray Doppler shifts nu[n,m]; the per-slot gains
sum_m A * exp(j 2 pi t nu) are computed on the GPU (channelmodel.ChannelModel.prepareForNextSlot).
"""
import os

import numpy as np

from .antenna import AntennaElement
from .channelmodel import ChannelModel
from .random import random
from .utils import toDb, toLinear, toRadian

_TABLES = None


def _tables():
    global _TABLES
    if _TABLES is None:
        _TABLES = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'channel_tables.npz'))
    return _TABLES


_DELAY_SPREADS = {"VeryShort": 10, "Short": 30, "Nominal": 100, "Long": 300, "VeryLong": 1000}   # TR 38.901 Table 7.7.3-1


def _wrap(angles, how):
    """Angle wrapping used around TR 38.901 7.7.5.1 (cdl.py:648-669)."""
    if how == "-pi,pi":
        return (angles + np.pi) % (2 * np.pi) - np.pi
    if how == "0,pi":
        a = angles % (2 * np.pi)
        a[a > np.pi] = 2 * np.pi - a[a > np.pi]
        return a
    if how == "0,2pi":
        return angles % (2 * np.pi)
    if how == "clip0,pi":
        return np.clip(angles, 0, np.pi)
    raise AssertionError(how)


class CdlChannel(ChannelModel):
    def __init__(self, bwp, profile='A', **kwargs):
        super().__init__(bwp, **kwargs)
        self.profile = profile
        if profile is not None and profile not in "ABCDE":
            raise ValueError(f"Unsupported CDL profile '{self.profile}'!")
        self.delaySpread = kwargs.get('delaySpread', 30)
        if isinstance(self.delaySpread, str):
            if self.delaySpread not in _DELAY_SPREADS:
                raise ValueError("'delaySpread' must be a number or one of 'VeryShort', 'Short', 'Nominal', 'Long', "
                                 "or 'VeryLong'")
            self.delaySpread = _DELAY_SPREADS[self.delaySpread]
        self.ueDirAZ = toRadian(kwargs.get('ueDirAZ', [0, 90]))
        self.txAntenna = kwargs.get('txAntenna', AntennaElement())
        self.rxAntenna = kwargs.get('rxAntenna', AntennaElement())
        self.txOrientation = toRadian(kwargs.get('txOrientation', [0, 0, 0]))
        self.rxOrientation = toRadian(kwargs.get('rxOrientation', [180, 0, 0]))
        self.kFactor = kwargs.get('kFactor', None)
        self.angleScaling = kwargs.get('angleScaling', None)
        if self.angleScaling is not None:
            a = self.angleScaling
            if (not isinstance(a, tuple)) or len(a) != 2 or any((not isinstance(v, (list, np.ndarray))) or len(v) != 4 for v in a):
                raise ValueError("'angleScaling' must be a tuple of two lists of length 4!")
            self.scalingAngleMeans, self.scalingAngleSpreads = toRadian(a[0]), toRadian(a[1])

        tab = None if profile is None else np.float64(_tables()['cdl_' + profile])
        col = (lambda i: None) if tab is None else (lambda i: tab[:, i].copy())
        self.pathDelays = kwargs.get('pathDelays', col(0))
        self.pathPowers = kwargs.get('pathPowers', col(1))
        self.aods, self.aoas = toRadian(kwargs.get('aods', col(2))), toRadian(kwargs.get('aoas', col(3)))
        self.zods, self.zoas = toRadian(kwargs.get('zods', col(4))), toRadian(kwargs.get('zoas', col(5)))
        self.hasLos = kwargs.get('hasLos', False if profile is None else (profile in "DE"))
        params = None if profile is None else _tables()['cdl_' + profile + '_params']
        self.xPolPower = kwargs.get('xPolPower', 10.0 if params is None else float(params[4]))
        for name in ('pathDelays', 'pathPowers', 'aods', 'aoas', 'zods', 'zoas'):
            if getattr(self, name) is None:
                raise ValueError(f"'{name}' is not specified for the custom CDL model!")
            setattr(self, name, np.float64(getattr(self, name)))
        if len({len(self.pathDelays), len(self.pathPowers), len(self.aods), len(self.aoas), len(self.zods), len(self.zoas)}) != 1:
            raise ValueError("Cluster information must have the same size!")
        self.kFactorLos = kwargs.get('kFactorLos', (self.pathPowers[0] - self.pathPowers[1]) if self.hasLos else None)
        if profile is not None:
            self.scaleDelays()
            if self.kFactor is not None:
                self.applyKFactorScaling()
        elif self.hasLos:
            # custom LOS model: split the first path into its LOS and NLOS parts (cdl.py:484-494)
            k1, p1 = toLinear(self.kFactorLos), toLinear(self.pathPowers[0])
            pw = -toDb(p1 + p1 / k1)
            self.pathPowers = np.concatenate(([pw, pw - self.kFactorLos], self.pathPowers[1:]))
            for name in ('pathDelays', 'aods', 'aoas', 'zods', 'zoas'):
                v = getattr(self, name)
                setattr(self, name, np.concatenate(([v[0]], v)))
        spreads = [4.0, 10.0, 2.0, 2.0] if params is None else params[:4]
        self.angleSpreads = toRadian(kwargs.get('angleSpreads', spreads))
        n, m = len(self.aods) - (1 if self.hasLos else 0), 20
        self.rayCoupling = kwargs.get('rayCoupling', None)
        self.randomRayCoupling = self.rayCoupling is None
        if not self.randomRayCoupling:
            self.rayCoupling = np.int32(self.rayCoupling)
            if self.rayCoupling.shape != (3, n, m):
                raise ValueError(f"Invalid 'rayCoupling' shape! Must be {(3, n, m)} but it is {self.rayCoupling.shape}")
            if np.any(self.rayCoupling >= m) or np.any(self.rayCoupling < 0):
                raise ValueError(f"'rayCoupling' values must be between 0 and {m} (inclusive)!")
        self.initialPhases = toRadian(kwargs.get('initialPhases', None))
        self.randomInitialPhases = self.initialPhases is None
        if not self.randomInitialPhases:
            self.initialPhases = np.float64(self.initialPhases)
            if self.initialPhases.shape != (2, 2, n, m):
                raise ValueError(f"Invalid 'initialPhases' shape! Must be {(2, 2, n, m)} but it is {self.initialPhases.shape}")
            if np.any(self.initialPhases < -np.pi) or np.any(self.initialPhases > np.pi):
                raise ValueError("'initialPhases' values must be between -𝛑 and 𝛑!")
        self.restart()

    # ------------------------------------------------------------------------------------------- randomness
    def restart(self, restartRanGen=False, applyToBwp=True):
        if (self.seed is not None) and restartRanGen:
            self.rangen = random.getGenerator(self.seed)
        if self.randomRayCoupling:
            self.rayCoupling = self.getRandomRayCoupling()
        if self.randomInitialPhases:
            self.initialPhases = self.getRandomInitialPhases()
        super().restart(restartRanGen, applyToBwp)

    def _nm(self):
        return len(self.aods) - (1 if self.hasLos else 0), 20

    def getRandomRayCoupling(self):
        """TR 38.901 7.7.1 Step 2; draw order = the reference's (3 x n permutations, cdl.py:814-818)."""
        n, m = self._nm()
        return np.int32([[self.rangen.choice(range(m), size=m, replace=False) for _ in range(n)] for _ in range(3)])

    def getRandomInitialPhases(self):
        n, m = self._nm()
        return 2 * np.pi * self.rangen.random(size=(2, 2, n, m)) - np.pi

    @classmethod
    def getMatlabRandomInit(cls, profile, seed):
        """Initial phases (degrees) and ray coupling reproducing MATLAB's nrCDLChannel streams (cdl.py:828-856)."""
        gen = random.getGenerator(np.random.RandomState(seed))
        los = 1 if profile in "DE" else 0
        n, m = len(_tables()['cdl_' + profile]), 20
        phi = np.transpose(gen.random(size=(4, m, n)), (0, 2, 1))[:, los:, :]
        phiInit = (360 * phi - 180).reshape(2, 2, n - los, m)
        order = np.argsort(gen.random(size=(3, m, n)), axis=1)
        coupling = np.zeros((3, m, n))
        coupling[[0, 2]] = order[[0, 2]]
        for i in range(n):
            coupling[1, :, i] = order[1, np.argsort(order[2, :, i]), i]
        coupling = np.int32(coupling.transpose((0, 2, 1))[:, los:, :])
        rows = np.arange(n - los)[:, None].repeat(m, 1)
        coupling[1] = coupling[1][(rows, coupling[2])]       # Matlab shuffles the zenith arrival angles twice
        return phiInit, coupling

    # ------------------------------------------------------------------------------------------------ model
    @property
    def nrNt(self):
        return (self.rxAntenna.getNumElements(), self.txAntenna.getNumElements())

    def scaleDelays(self):
        self.pathDelays *= self.delaySpread                  # TR 38.901 7.7.3

    def applyAngleScaling(self, phiD, phiA, thetaD, thetaA, p):
        """TR 38.901 7.7.5.1 + Annex A (cdl.py:890-930)."""
        m = phiA.shape[1]

        def model(ang):
            w = (np.exp(1j * ang) * p.reshape(-1, 1)).sum() / m
            return np.angle(w), np.sqrt(-2 * np.log(np.abs(w / p.sum())))

        def scale(ang, asD, maD):
            maM, asM = model(ang)
            return (ang - maM + maD) if asM == 0 else asD * (ang - maM) / asM + maD

        sp, mu = self.scalingAngleSpreads, self.scalingAngleMeans
        return (_wrap(scale(phiD, sp[0], mu[0]), "0,2pi"), _wrap(scale(phiA, sp[1], mu[1]), "0,2pi"),
                _wrap(scale(thetaD, sp[2], mu[2]), "clip0,pi"), _wrap(scale(thetaA, sp[3], mu[3]), "clip0,pi"))

    def _doppler(self, theta, phi):
        """Per-ray Doppler shift [Hz]: (r_rx . v_hat) f_D  (TR 38.901 Eq 7.5-22 last factor; cdl.py:871-887)."""
        vp, vt = self.ueDirAZ
        dbar = self.dopplerShift * np.array([np.sin(vt) * np.cos(vp), np.sin(vt) * np.sin(vp), np.cos(vt)])
        st = np.sin(theta)
        rhat = np.array([st * np.cos(phi), st * np.sin(phi), np.cos(theta)])
        return (rhat * dbar.reshape(3, 1, 1)).sum(0)

    def staticCoefficients(self):
        o = 1 if self.hasLos else 0
        offs = np.float64(_tables()['ray_offsets'])
        cASD, cASA, cZSD, cZSA = self.angleSpreads
        phiD = self.aods[o:].reshape(-1, 1) + cASD * offs                   # TR 38.901 7.7.1 Step 1
        phiA = self.aoas[o:].reshape(-1, 1) + cASA * offs
        thD = self.zods[o:].reshape(-1, 1) + cZSD * offs
        thA = self.zoas[o:].reshape(-1, 1) + cZSA * offs
        pN = toLinear(self.pathPowers[o:])
        if self.angleScaling is not None:
            phiD, phiA, thD, thA = self.applyAngleScaling(phiD, phiA, thD, thA, pN)
        phiD, phiA = _wrap(phiD, "-pi,pi"), _wrap(phiA, "-pi,pi")
        thD, thA = _wrap(thD, "0,pi"), _wrap(thA, "0,pi")
        n, m = phiD.shape
        rows = np.arange(n)[:, None].repeat(m, 1)                            # Step 2: ray coupling
        phiA = phiA[(rows, self.rayCoupling[0])]
        thA = thA[(rows, self.rayCoupling[1])]
        thD = thD[(rows, self.rayCoupling[2])]
        kappa = toLinear(self.xPolPower)                                      # Step 3
        pol = np.exp(1j * self.initialPhases) * np.sqrt([[1, 1 / kappa], [1 / kappa, 1]]).reshape(2, 2, 1, 1)
        fTx, lTx = self.txAntenna.getElementsFields(thD, phiD, self.txOrientation)
        fRx, lRx = self.rxAntenna.getElementsFields(thA, phiA, self.rxOrientation)
        A = np.einsum('ranm,abnm,tbnm->rtnm', fRx, pol, fTx) * lRx[:, None] * lTx[None, :]
        A = A * np.sqrt(pN / m).reshape(1, 1, -1, 1)
        nu = self._doppler(thA, phiA)
        Alos, nulos = None, 0.0
        if self.hasLos:                                                       # TR 38.901 Eq 7.5-29
            pd, pa = self.aods[0:1].reshape(1, 1), self.aoas[0:1].reshape(1, 1)
            td, ta = self.zods[0:1].reshape(1, 1), self.zoas[0:1].reshape(1, 1)
            p0 = toLinear(self.pathPowers[0])
            if self.angleScaling is not None:
                pd, pa, td, ta = self.applyAngleScaling(pd, pa, td, ta, p0)
            pd, pa, td, ta = _wrap(pd, "-pi,pi"), _wrap(pa, "-pi,pi"), _wrap(td, "0,pi"), _wrap(ta, "0,pi")
            ft, lt = self.txAntenna.getElementsFields(td, pd, self.txOrientation)
            fr, lr = self.rxAntenna.getElementsFields(ta, pa, self.rxOrientation)
            polLos = np.float64([[1, 0], [0, -1]])
            Alos = np.einsum('ra,ab,tb->rt', fr[:, :, 0, 0], polLos, ft[:, :, 0, 0]) * lr[:, 0, 0][:, None] * lt[:, 0, 0][None, :]
            Alos = Alos * np.sqrt(p0)
            nulos = float(self._doppler(ta, pa)[0, 0])
        return A, nu, Alos, nulos

    def print(self, indent=0, title=None, getStr=False):
        if title is None:
            title = "Customized CDL Channel Properties:" if self.profile is None else f"CDL-{self.profile} Channel Properties:"
        s = super().print(indent, title, True)
        pad = indent * ' '
        s += pad + f"  delaySpread:          {self.delaySpread} ns\n"
        s += pad + f"  Cross Pol. Power:     {self.xPolPower} dB\n"
        s += pad + f"  hasLOS:               {self.hasLos}\n"
        s += self.txAntenna.print(indent + 2, "TX Antenna:", True)
        s += self.rxAntenna.print(indent + 2, "RX Antenna:", True)
        if getStr:
            return s
        print(s)
