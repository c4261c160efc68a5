"""TR 38.901 antenna elements, panels and arrays -- the part the CDL channel needs (reference antenna.py).

Host-side, evaluated once per channel object: ``getElementsFields`` returns the polarised element fields and
the array location phasors per ray, from which cdl.py builds the static coefficient tensor that the GPU gain
kernel consumes.  Radiation-pattern plotting / directivity integrals (antenna.py:861-1040) are out of scope.
"""
import numpy as np

from .utils import toLinear


class AntennaBase:
    def __init__(self, **kwargs):
        self.isElement = isinstance(self, AntennaElement)

    def getNumElements(self):
        return 1

    @staticmethod
    def getRotationMatrix(orientation):
        """TR 38.901 Eq 7.1-4: forward composite rotation for bearing/downtilt/slant (alpha, beta, gamma) [rad]."""
        if not np.any(orientation):
            return np.eye(3)
        sa, sb, sg = np.sin(orientation)
        ca, cb, cg = np.cos(orientation)
        return np.float64([[ca * cb, ca * sb * sg - sa * cg, ca * sb * cg + sa * sg],
                           [sa * cb, sa * sb * sg + ca * cg, sa * sb * cg - ca * sg],
                           [-sb, cb * sg, cb * cg]])

    def getSteeringVector(self, theta, phi):
        t = np.asarray(theta, dtype=np.float64).reshape(-1, 1) * np.pi / 180
        p = np.asarray(phi, dtype=np.float64).reshape(1, -1) * np.pi / 180
        xyz = np.float64([np.sin(t) * np.cos(p), np.sin(t) * np.sin(p), np.cos(t) * np.ones_like(p)])
        return np.exp(2j * np.pi * np.tensordot(self.getAllPositions(), xyz, axes=1))

    def getElementsFields(self, theta, phi, orientation=np.float64([0, 0, 0])):
        """Per-ray global (theta, phi) field components and location phasors of every element
        (antenna.py:765-859; TR 38.901 7.1 coordinate transforms + Eq 7.5-22/-28).

        theta, phi: (n, m) zenith/azimuth angles [rad].  Returns field (nAnt, 2, n, m), loc (nAnt, n, m)."""
        n, m = theta.shape
        st, ct, sp, cp = np.sin(theta), np.cos(theta), np.sin(phi), np.cos(phi)
        rhat = np.array([st * cp, st * sp, ct])                                  # Eq 7.5-23
        R = self.getRotationMatrix(orientation)
        # local angles (Eq 7.1-7 / 7.1-8): components of rhat along the rotated axes (columns of R)
        thl = np.arccos((R[:, 2, None, None] * rhat).sum(0))
        phl = np.arctan2((R[:, 1, None, None] * rhat).sum(0), (R[:, 0, None, None] * rhat).sum(0))
        phl[thl == 0] = 0
        phl[thl == np.pi] = 0
        th_hat = np.float64([ct * cp, ct * sp, -st])                             # Eq 7.1-13
        ph_hat = np.float64([-sp, cp, np.zeros_like(cp)])                        # Eq 7.1-14
        cl = np.cos(thl)
        thl_hat = np.float64([cl * np.cos(phl), cl * np.sin(phl), -np.sin(thl)])
        g = R.dot(thl_hat.reshape(3, -1))
        psi = np.arctan2((ph_hat.reshape(3, -1) * g).sum(0), (th_hat.reshape(3, -1) * g).sum(0)).reshape(n, m)  # 7.1-12
        pairs = [e.getPolarizedFields(thl * 180 / np.pi, phl * 180 / np.pi) for e in self.allElements()]
        fth = np.array([a for a, _ in pairs]).reshape(-1, n, m)
        fph = np.array([b for _, b in pairs]).reshape(-1, n, m)
        field = np.stack((fth * np.cos(psi) - fph * np.sin(psi), fth * np.sin(psi) + fph * np.cos(psi)), axis=1)
        pos = R.dot(self.getAllPositions().T)                                    # global positions (wavelengths)
        loc = np.exp(1j * 2 * np.pi * (rhat[:, None, :, :] * pos[:, :, None, None]).sum(0))
        return field, loc


class AntennaElement(AntennaBase):
    """TR 38.901 Table 7.3-1 element pattern with polarisation model 1 or 2 (antenna.py:1042-1262)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.position = np.float64(kwargs.get('position', [0, 0, 0]))
        self.freqRange = kwargs.get('freqRange', [0, 100e9])
        self.polAngle = kwargs.get('polAngle', 0)
        self.polModel = kwargs.get('polModel', 2)
        self.beamWidth = kwargs.get('beamWidth', [65, 65])
        self.verticalSidelobeAttenuation = kwargs.get('verticalSidelobeAttenuation', 30)
        self.maxAttenuation = kwargs.get('maxAttenuation', 30)
        self.mainMaxGain = kwargs.get('mainMaxGain', 8)
        self.panel = kwargs.get('panel', None)

    @property
    def posInArray(self):
        return self.position + self.panel.position

    def clone(self, position, polAngle, panel):
        return AntennaElement(freqRange=self.freqRange, polAngle=polAngle, polModel=self.polModel,
                              beamWidth=self.beamWidth, verticalSidelobeAttenuation=self.verticalSidelobeAttenuation,
                              maxAttenuation=self.maxAttenuation, mainMaxGain=self.mainMaxGain, position=position,
                              panel=panel)

    def allElements(self, polarization=True):
        return [self]

    def getAllPositions(self, polarization=True):
        return np.float64([self.position])

    def verticalRadiationPower(self, theta):
        theta = np.asarray(theta, dtype=np.float64)
        return -np.minimum(12 * np.square((theta - 90) / self.beamWidth[0]), self.verticalSidelobeAttenuation)

    def horizonRadiationPower(self, phi):
        phi = np.asarray(phi, dtype=np.float64)
        if self.beamWidth[1] == 360:
            return np.zeros(phi.shape)
        return -np.minimum(12 * np.square(phi / self.beamWidth[1]), self.maxAttenuation)

    def getPowerPatternDb(self, theta, phi):
        theta, phi = np.asarray(theta, dtype=np.float64), np.asarray(phi, dtype=np.float64)
        if theta.ndim == 1 and phi.ndim == 1 and len(theta) != len(phi):
            a = self.verticalRadiationPower(theta).reshape(-1, 1) + self.horizonRadiationPower(phi).reshape(1, -1)
        else:
            a = self.verticalRadiationPower(theta) + self.horizonRadiationPower(phi)
        return np.float64(np.squeeze(-np.minimum(-a, self.maxAttenuation) + self.mainMaxGain))

    def getPowerPattern(self, theta, phi):
        return toLinear(self.getPowerPatternDb(theta, phi))

    def getField(self, theta, phi):
        return toLinear(self.getPowerPatternDb(theta, phi) / 2)

    def getPolarizedFields(self, theta, phi):
        """TR 38.901 7.3.2 (angles in degrees, local coordinates)."""
        field = self.getField(theta, phi)
        zeta = self.polAngle * np.pi / 180
        if self.polModel == 1:
            if self.polAngle == 0:
                c, s = 1, 0
            elif self.polAngle in [180, -180]:
                c, s = -1, 0
            else:
                t = np.asarray(theta, dtype=np.float64).reshape(-1, 1) * np.pi / 180
                p = np.asarray(phi, dtype=np.float64).reshape(1, -1) * np.pi / 180
                den = np.sqrt(1 - np.square(np.cos(zeta) * np.cos(t) - np.sin(zeta) * np.sin(p) * np.sin(t)))
                c = (np.cos(zeta) * np.sin(t) + np.sin(zeta) * np.sin(p) * np.cos(t)) / den
                s = np.sin(zeta) * np.cos(p) / den
            return field * c, field * s
        return field * np.cos(zeta), field * np.sin(zeta)

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + ("Antenna Element:" if title is None else title) + "\n"
        s += pad + f"  polAngle:                    {self.polAngle}°\n" + pad + f"  polModel:                    {self.polModel}\n"
        s += pad + f"  beamWidth:                   {self.beamWidth[0]}°,{self.beamWidth[1]}°\n"
        s += pad + f"  maxAttenuation:              {self.maxAttenuation} dB\n"
        s += pad + f"  mainMaxGain:                 {self.mainMaxGain} dBi\n"
        if getStr:
            return s
        print(s)


class AntennaPanel(AntennaBase):
    """Uniform rectangular panel of (dual-)polarised elements in the y-z plane (TR 38.901 7.3)."""

    def __init__(self, shape=[2, 2], **kwargs):
        super().__init__(**kwargs)
        self.shape = np.int16(shape)
        if self.shape.shape != (2,):
            raise ValueError("'shape' must be a list or NumPy array of length 2.")
        self.spacing = np.float64(kwargs.get('spacing', [.5, .5]))
        if self.spacing.shape != (2,):
            raise ValueError("'spacing' must be a list or NumPy array of length 2.")
        self.polarization = kwargs.get('polarization', "|")
        if self.polarization not in "|-+x":
            raise ValueError("'polarization' must be one of \"|\", \"-\", \"+\", or \"x\".")
        self.position = np.float64(kwargs.get('position', [0, 0, 0]))
        if self.position.shape != (3,):
            raise ValueError("'position' must be a list or NumPy array of length 3.")
        self.array = kwargs.get('array', None)
        self.matlabOrder = kwargs.get('matlabOrder', False)
        elements = kwargs.get('elements', None)
        if elements is None:
            template = AntennaElement(**kwargs)
        elif isinstance(elements, AntennaElement):
            template = elements
        elif isinstance(elements, list):
            template = None
            if len(elements) != self.shape[0] or any((not isinstance(r, list)) or len(r) != self.shape[1] for r in elements):
                raise ValueError("'elements' shape does not match the provided 'shape'!")
            self.elements = elements
        else:
            raise ValueError("'elements' must be an 'AntennaElement' object, a 2-D array of `AntennaElement` objects, "
                             "or None.")
        if template is not None:
            rows, cols = self.shape
            oz, oy = (self.shape - 1) * self.spacing / 2
            dz, dy = self.spacing
            slants = {"|": [0], "-": [90], "+": [0, 90], "x": [45, -45]}[self.polarization]
            self.elements = [[[template.clone([0, c * dy - oy, r * dz - oz], a, self) for a in slants]
                              for c in range(cols)] for r in range(rows)]

    def clone(self, position, array):
        return AntennaPanel(self.shape, spacing=self.spacing, polarization=self.polarization,
                            elements=self.elements[0][0][0], position=position, array=array)

    def getNumElements(self):
        return int(np.prod(self.shape)) * (1 if self.polarization in "-|" else 2)

    def getElement(self, elementRC=(0, 0), p=0):
        if elementRC == 0:
            elementRC = (0, 0)
        if elementRC == -1:
            elementRC = (-1, -1)
        return self.elements[elementRC[0]][elementRC[1]][p]

    def getElementPosition(self, elementRC=(0, 0), ref="Array"):
        return self.getElement(elementRC).position + (0 if ref == "Panel" else self.position)

    def allElements(self, polarization=True):
        """Element order: polarisation-major, then rows, then columns (or Matlab's column-major bottom-up order)."""
        npol = (2 if self.polarization in "+x" else 1) if polarization else 1
        rr, cc = self.shape
        out = []
        for p in range(npol):
            if self.matlabOrder:
                out += [self.elements[r][c][p] for c in range(cc) for r in range(rr - 1, -1, -1)]
            else:
                out += [self.elements[r][c][p] for r in range(rr) for c in range(cc)]
        return out

    def getAllPositions(self, polarization=True):
        return np.float64([e.position for e in self.allElements(polarization)])

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + ("Antenna Panel:" if title is None else title) + "\n"
        s += pad + f"  Total Elements:  {self.getNumElements()}\n"
        s += pad + f"  spacing:         {self.spacing[0]}𝜆, {self.spacing[1]}𝜆\n"
        s += pad + f"  shape:           {self.shape[0]} rows x {self.shape[1]} columns\n"
        s += pad + f"  polarization:    {self.polarization}\n"
        if getStr:
            return s
        print(s)


class AntennaArray(AntennaBase):
    """Rectangular array of identical panels (TR 38.901 7.3, Mg x Ng panels)."""

    def __init__(self, shape=[1, 1], **kwargs):
        super().__init__(**kwargs)
        self.shape = np.int16(shape)
        if self.shape.shape != (2,):
            raise ValueError("'shape' must be a list or NumPy array of length 2.")
        spacing = kwargs.get('spacing', None)
        panels = kwargs.get('panels', None)
        if panels is None:
            template = AntennaPanel()
        elif isinstance(panels, AntennaPanel):
            template = panels
        elif isinstance(panels, list):
            template = None
            if len(panels) != self.shape[0] or any((not isinstance(r, list)) or len(r) != self.shape[1] for r in panels):
                raise ValueError("'panels' shape does not match the provided 'shape'!")
            self.panels = panels
            self.spacing = np.float64(spacing) if spacing is not None else panels[0][0].shape * panels[0][0].spacing
        else:
            raise ValueError("'panels' must be an 'AntennaPanel' object, a 2-D array of `AntennaPanel` objects, or None.")
        if template is not None:
            self.spacing = (template.shape * template.spacing) if spacing is None else np.float64(spacing)
            if self.spacing.shape != (2,):
                raise ValueError("'spacing' must be a list or NumPy array of length 2.")
            rows, cols = self.shape
            oz, oy = (self.shape - 1) * self.spacing / 2
            dz, dy = self.spacing
            self.panels = [[template.clone([0, c * dy - oy, r * dz - oz], self) for c in range(cols)] for r in range(rows)]

    @property
    def polarization(self):
        return self.panels[0][0].polarization

    def allPanels(self):
        return [self.panels[r][c] for r in range(self.shape[0]) for c in range(self.shape[1])]

    def allElements(self, polarization=True):
        if polarization and (self.polarization in "+x"):
            out = []
            for p in (0, 1):
                for panel in self.allPanels():
                    out += [panel.elements[r][c][p] for r in range(panel.shape[0]) for c in range(panel.shape[1])]
            return out
        return [e for panel in self.allPanels() for e in panel.allElements(False)]

    def getAllPositions(self, polarization=True):
        return np.float64([e.posInArray for e in self.allElements(polarization)])

    def getNumElements(self):
        return int(np.prod(self.shape)) * self.panels[0][0].getNumElements()

    def getElement(self, panelRC=(0, 0), elementInPanelRC=(0, 0), p=0):
        if panelRC == 0:
            panelRC, elementInPanelRC = (0, 0), (0, 0)
        if panelRC == -1:
            panelRC, elementInPanelRC = (-1, -1), (-1, -1)
        return self.panels[panelRC[0]][panelRC[1]].elements[elementInPanelRC[0]][elementInPanelRC[1]][p]

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + ("Antenna Array:" if title is None else title) + "\n"
        s += pad + f"  Total Elements:  {self.getNumElements()}\n"
        s += pad + f"  shape:           {self.shape[0]} rows x {self.shape[1]} columns of panels\n"
        if getStr:
            return s
        print(s)
