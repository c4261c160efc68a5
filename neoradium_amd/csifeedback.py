"""CSI report: Type-I single-panel codebooks and the PMI / rank search (reference csifeedback.py:160-1330; TS 38.214 5.2.2.2.1).

What the reference can actually run is what is built (probed against it, tests/golden/csifeedback.npz):
single-panel Type-I codebooks on a one-row panel (N2 = 1: 2, 4, 8, 12, 16, 24, 32 ports), codebook modes 1 and 2, ranks
1-5, wideband and sub-band PMI, rank selection.  For N2 > 1 the reference's precoders come out as (2 N1, N2 x layers)
arrays that its own SINR routine rejects, its multi-panel generator fails to unpack its own index tuples, Type-II is
commented out and ranks >= 6 stop on an undefined attribute -- those raise NotImplementedError here.

The search's arithmetic -- the post-MMSE SINR of every (codebook entry, CSI-RS resource element, layer),
csifeedback.py:419-433 -- runs on the GPU (nrx_csi_sinr_f64); enumerating the codebook and picking maxima over a few
hundred sums is host bookkeeping.
"""
import itertools

import numpy as np

from . import ops
from ._dev import D, N
from .antenna import AntennaArray, AntennaPanel

# TS 38.214 Tables 5.2.2.1-2..5: (modulation, code rate x 1024, efficiency) per CQI index, index 0 = out of range
_CQI_ROWS = {
    1: "Q78 Q120 Q193 Q308 Q449 Q602 16.378 16.490 16.616 64.466 64.567 64.666 64.772 64.873 64.948",
    2: "Q78 Q193 Q449 16.378 16.490 16.616 64.466 64.567 64.666 64.772 64.873 256.711 256.797 256.885 256.948",
    3: "Q30 Q50 Q78 Q120 Q193 Q308 Q449 Q602 16.378 16.490 16.616 64.466 64.567 64.666 64.772",
    4: "Q78 Q193 Q449 16.378 16.616 64.567 64.666 64.772 64.873 256.711 256.797 256.885 256.948 1024.853 1024.948",
}
_EFF = {('Q', 30): 0.0586, ('Q', 50): 0.0977, ('Q', 78): 0.1523, ('Q', 120): 0.2344, ('Q', 193): 0.3770, ('Q', 308): 0.6016,
        ('Q', 449): 0.8770, ('Q', 602): 1.1758, ('16', 378): 1.4766, ('16', 490): 1.9141, ('16', 616): 2.4063,
        ('64', 466): 2.7305, ('64', 567): 3.3223, ('64', 666): 3.9023, ('64', 772): 4.5234, ('64', 873): 5.1152,
        ('64', 948): 5.5547, ('256', 711): 5.5547, ('256', 797): 6.2266, ('256', 885): 6.9141, ('256', 948): 7.4063,
        ('1024', 853): 8.3301, ('1024', 948): 9.2578}


def _cqi_table(n):
    rows = [[None, None, None]]
    for tok in _CQI_ROWS[n].split():
        mod, rate = ('Q', tok[1:]) if tok[0] == 'Q' else tok.split('.')
        rows.append(['QPSK' if mod == 'Q' else mod + 'QAM', int(rate), _EFF[(mod, int(rate))]])
    return rows


cqiTables = [None] + [_cqi_table(n) for n in (1, 2, 3, 4)]
cqiTableBLERs = [None, 0.1, 0.1, 0.1, 0.00001, 0.1]


def _check(name, value, valid, context=""):
    """Range check with the reference's message layout (csifeedback.py:38-58)."""
    if isinstance(valid, list):
        if value in valid:
            return
        q = "'%s'" if isinstance(valid[0], str) else "%s"
        raise ValueError("Invalid '%s'! ('%s' ∈ {%s}%s)" % (name, name, ", ".join(q % str(x) for x in valid), context))
    if isinstance(valid, tuple):
        if value in range(valid[0], valid[1] + 1):
            return
        raise ValueError("Invalid '%s'! ('%s' ∈ {%s}%s)" % (name, name, ",...,".join(str(x) for x in valid), context))
    if value != valid:
        q = "'%s'" if isinstance(valid, str) else "%s"
        raise ValueError("Invalid '%s'! (It must be " % (name) + q % str(valid) + context + ")")


class CsiReport:
    """CSI-ReportConfig (csifeedback.py:160-383) with the Type-I single-panel PMI / RI search."""

    def __init__(self, csiRsConfig, **kwargs):
        self.reportId = kwargs.get('id', 0)
        self.csiRsConfig = csiRsConfig
        self.bwp = csiRsConfig.bwp
        if any(s.csiType == "ZP" for s in csiRsConfig.csiRsSetList):
            raise ValueError("`ZP` resources are not allowed in 'csiRsConfig'.")
        self.measurementRes = kwargs.get('measurementRes', [])
        self.interfereResIm = kwargs.get('interfereResIm', [])
        self.interfereResNzp = kwargs.get('interfereResNzp', [])
        self.reportType = kwargs.get('reportType', "Periodic")
        _check('reportType', self.reportType, ["Periodic", "SpOnPUCCH", "SpOnPUSCH", "Aperiodic"])
        self.period, self.offset = kwargs.get('period', 5), kwargs.get('offset', 0)
        if self.reportType in ["Periodic", "SpOnPUCCH"]:
            _check('period', self.period, [5, 10, 20, 40, 80, 160, 320])
        elif self.reportType == "SpOnPUSCH":
            _check('period', self.period, [4, 5, 8, 10, 16, 20, 32, 40, 80, 160, 320])
        _check('offset', self.offset, (0, self.period - 1))
        self.quantity = kwargs.get('quantity', 'CriRiPmiCqi')
        _check('quantity', self.quantity, ['CriRiPmiCqi', 'CriRiLiPmiCqi', 'CriRiI1', 'CriRiCqi', 'CriRiI1Cqi', 'CriRsrp',
                                           'SsbRIdxRsrp', 'CriSinr', 'SsbIdxSinr'])
        self.groupBeams = kwargs.get('groupBeams', True)
        self.noOfRepRS = kwargs.get('noOfRepRS', 1)
        if not self.groupBeams:
            _check('noOfRepRS', self.noOfRepRS, (1, 4))
        self.codebookType = kwargs.get('codebookType', 'Type1SP')
        _check('codebookType', self.codebookType, ['Type1SP', 'Type1MP', 'Type2', 'EnancedType2'])
        self.txAntenna = kwargs.get('txAntenna', None)
        if self.txAntenna is None:
            self.n1, self.n2, self.ng = kwargs.get('n1', None), kwargs.get('n2', None), kwargs.get('ng', None)
            need = (self.n1, self.n2, self.ng) if self.codebookType == 'Type1MP' else (self.n1, self.n2)
            if any(v is None for v in need):
                raise ValueError("The antenna configuration is missing! (A 'txAntenna' or %sn1/n2 values must be specified)" %
                                 ("ng/" if self.codebookType == 'Type1MP' else ""))
            if self.ng is None:
                self.ng = 1
        elif isinstance(self.txAntenna, AntennaPanel):
            if self.codebookType == 'Type1MP':
                raise ValueError("Single-Panel 'txAntenna' is configured with Multi-Panel 'codebookType' (Type1MP)!")
            self.ng = 1
            self.n2, self.n1 = self.txAntenna.shape
        elif isinstance(self.txAntenna, AntennaArray):
            self.ng = int(np.prod(self.txAntenna.shape))
            if (self.ng > 1) and (self.codebookType == 'Type1SP'):
                raise ValueError("Multi-Panel 'txAntenna' is configured with Single-Panel 'codebookType' (Type1SP)!")
            self.n2, self.n1 = self.txAntenna.panel[0][0].shape
        else:
            raise ValueError("Unsupported antenna class '%s'!" % (self.txAntenna.__class__.__name__))
        if self.codebookType in ['Type1SP', 'Type2']:
            if (self.n1, self.n2) not in [(1, 1), (2, 1), (2, 2), (4, 1), (3, 2), (6, 1), (4, 2), (8, 1), (4, 3), (6, 2),
                                          (12, 1), (4, 4), (8, 2), (16, 1)]:
                raise ValueError("Invalid N1-N2 combination %d-%d. See TS 38.214, Table 5.2.2.2.1-2" % (self.n1, self.n2))
        elif self.codebookType == 'Type1MP':
            if (self.ng, self.n1, self.n2) not in [(2, 2, 1), (2, 4, 1), (4, 2, 1), (2, 2, 2), (2, 8, 1), (4, 4, 1), (2, 4, 2),
                                                   (4, 2, 2)]:
                raise ValueError("Invalid Ng-N1-N2 combination %d-%d-%d. See TS 38.214, Table 5.2.2.2.2-1" %
                                 (self.ng, self.n1, self.n2))
        if self.codebookType in ['Type1SP', 'Type1MP']:
            self.codebookMode = kwargs.get('codebookMode', 1)
            if self.ng == 4:
                _check('codebookMode', self.codebookMode, 1, " when Ng is 4")
            else:
                _check('codebookMode', self.codebookMode, [1, 2])
            self.o1, self.o2 = 4, (4 if self.n2 > 1 else 1)
        self.numPorts = 2 * self.ng * self.n1 * self.n2
        self.ac = self.n1 * self.o1 * self.n2 * self.o2 if self.codebookType in ['Type1SP', 'Type1MP'] else 0
        self.cbSubsetRestriction = kwargs.get('cbSubsetRestriction', max(8, 2 * self.ac) * '1')
        self.cbSubsetRestrictionI2 = kwargs.get('cbSubsetRestrictionI2', 16 * '1')
        self.cbRiRestriction = kwargs.get('cbRiRestriction', 8 * '1')
        self.prgSize = kwargs.get('prgSize', None)
        if self.prgSize is not None and self.prgSize not in [0, 2, 4]:
            raise ValueError("'prgSize' must be 0 (Wideband), 2, or 4)")
        nrb = self.bwp.numRbs
        sizes = [0] if nrb < 24 else ([4, 8] if nrb < 73 else ([8, 16] if nrb < 145 else [16, 32]))
        subbandSize = kwargs.get('subbandSize', sizes[0])
        _check('subbandSize', subbandSize, sizes)
        self.subbandSizePmi = kwargs.get('subbandSizePmi', subbandSize)
        self.subbandSizeCqi = kwargs.get('subbandSizeCqi', subbandSize)
        _check('subbandSizePmi', self.subbandSizePmi, sizes)
        _check('subbandSizeCqi', self.subbandSizeCqi, sizes)
        self.cqiTable = kwargs.get('cqiTable', 1)
        _check('cqiTable', self.cqiTable, [1, 2, 3, 4])

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title=None, getStr=False):
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + ("CSI Report Properties:" if title is None else title) + "\n"
        rows = [("Report ID:            ", self.reportId), ("Report Type:          ", self.reportType),
                ("codebookType:         ", self.codebookType)]
        if self.codebookType in ['Type1SP', 'Type1MP']:
            rows += [("codebookMode:         ", self.codebookMode)]
        if self.reportType in ["Periodic", "SpOnPUCCH", "SpOnPUSCH"]:
            rows += [("period:               ", self.period), ("offset:               ", self.offset)]
        rows += [("quantity:             ", self.quantity), ("groupBeams:           ", self.groupBeams)]
        if not self.groupBeams:
            rows += [("noOfRepRS:            ", self.noOfRepRS)]
        rows += [("Ng x N1 x N2:         ", f"{self.ng} x {self.n1} x {self.n2}")]
        if self.codebookType in ['Type1SP', 'Type1MP']:
            rows += [("o1 x o2:              ", f"{self.o1} x {self.o2}")]
        rows += [("numPorts:             ", self.numPorts), ("cbSubsetRestriction:  ", self.cbSubsetRestriction),
                 ("cbSubsetRestrictionI2:", self.cbSubsetRestrictionI2), ("cbRiRestriction:      ", self.cbRiRestriction),
                 ("prgSize:              ", self.prgSize), ("subbandSizePmi:       ", self.subbandSizePmi),
                 ("subbandSizeCqi:       ", self.subbandSizeCqi), ("cqiTable:             ", self.cqiTable)]
        for k, v in rows:
            s += pad + f"  {k}{v}\n"
        if getStr:
            return s
        print(s)

    # ------------------------------------------------------------------------------------------------- bookkeeping
    def subbands(self, sbSize):
        """Sizes (in RBs) of the sub-bands of the BWP, aligned to multiples of ``sbSize`` from CRB 0 (csifeedback.py:436-447)."""
        rb, end = self.bwp.startRb, self.bwp.startRb + self.bwp.numRbs
        first = True
        while rb < end:
            size = sbSize - (rb % sbSize) if first else (end % sbSize if rb + sbSize > end else sbSize)
            yield size
            rb, first = rb + size, False

    def getCqiToPmiIdxes(self, pmiSbSize):
        """For every CQI sub-band the PMI sub-bands it overlaps (csifeedback.py:539-560)."""
        cqi = [self.bwp.numRbs] if self.subbandSizeCqi == 0 else list(self.subbands(self.subbandSizeCqi))
        pmi = [self.bwp.numRbs] if pmiSbSize == 0 else list(self.subbands(pmiSbSize))
        out = [[] for _ in cqi]
        j, covered_p, covered_c = 0, pmi[0], 0
        for i, size in enumerate(cqi):
            out[i].append(j)
            covered_c += size
            while True:
                if covered_p == covered_c:
                    j += 1
                    if j < len(pmi):
                        covered_p = pmi[j]
                    covered_c = 0
                    break
                if covered_p > covered_c:
                    break
                covered_p += pmi[j]
                j += 1
                out[i].append(j)
        return out

    def removeNeighbors(self, idx):
        """Of each cluster of adjacent CSI-RS REs (a CDM group) keep one: an RE survives if its lower-subcarrier neighbour
        is not an RE, and the RE one symbol earlier did not survive that first rule (csifeedback.py:405-416)."""
        res = set(zip(idx[0].tolist(), idx[1].tolist()))
        first = {(l, k) for (l, k) in res if (l, k - 1) not in res}
        keep = sorted((l, k) for (l, k) in first if (l - 1, k) not in first)
        return (np.int64([l for l, _ in keep]), np.int64([k for _, k in keep]))

    # ------------------------------------------------------------------------------------------------- codebooks
    def v(self, l, m, tilde=False):
        if tilde in [True, '~']:
            ul = np.exp(4j * np.pi * l * np.arange(self.n1 // 2) / (self.n1 * self.o1))
        else:
            ul = np.exp(2j * np.pi * l * np.arange(self.n1) / (self.n1 * self.o1))
        um = np.exp(2j * np.pi * m * np.arange(self.n2) / (self.n2 * self.o2))
        return np.outer(ul, um)

    def getCombs(self, *argv):
        return [list(t) for t in itertools.product(*[(a if isinstance(a, list) else range(a)) for a in argv])]

    def _i13_len(self, numLayers):
        if numLayers == 2:
            return 2 if (self.n1, self.n2) == (2, 1) else 4                     # Table 5.2.2.2.1-3
        if self.numPorts >= 16:
            return 4
        return {(2, 1): 1, (4, 1): 3, (2, 2): 3}.get((self.n1, self.n2), 4)     # Table 5.2.2.2.1-4

    def type1SpIndexes(self, numLayers):
        """Every allowed ([i11, i12, i13], i2) of the Type-I single-panel codebook for ``numLayers`` in the reference's
        enumeration order (csifeedback.py:599-721; TS 38.214 Tables 5.2.2.2.1-1, -5 .. -12)."""
        b1, b2 = self.n1 * self.o1, self.n2 * self.o2
        mask = self.cbSubsetRestriction
        i2mask = self.cbSubsetRestrictionI2 if self.quantity == 'CriRiI1Cqi' else 16 * '1'
        ctx = " with %d layers, CB Mode %d, %dx%d Ant" % (numLayers, self.codebookMode, self.n1, self.n2)
        if self.numPorts == 2:
            _check('numLayer', numLayers, [1, 2], " when 'numPorts' is 2")
            allowed = mask[-4:] if numLayers == 1 else mask[-6:-4]
            for i1 in range(4):
                if allowed[i1]:                 # (a character: always true; two layers run off the 2-bit field as in the reference)
                    yield [i1, 0, 0], 0
            return
        mode2 = self.codebookMode == 2
        if numLayers in (1, 2):
            n_i2 = (4 if not mode2 else 16) if numLayers == 1 else (2 if not mode2 else 8)
            per = n_i2 // 4 if mode2 else 0                                     # i2 values per (l, m) offset in mode 2
            i13s = [0] if numLayers == 1 else list(range(self._i13_len(2)))
            if not mode2:
                i11s, i12s = range(b1), range(b2)
            elif self.n2 > 1:
                i11s, i12s = range(b1 // 2), range(b2 // 2)
            else:
                i11s, i12s = range(b1 // 2), [0]
            for i11, i12, i13, i2 in itertools.product(i11s, i12s, i13s, range(n_i2)):
                if not mode2:
                    l, m = i11, i12
                elif self.n2 > 1:
                    l, m = 2 * i11 + (i2 // per) % 2, 2 * i12 + i2 // (2 * per)
                else:
                    l, m = 2 * i11 + i2 // per, 0
                if mask[b2 * l + m] == '0' or i2mask[i2] == '0':
                    continue
                yield [i11, i12, i13], i2
        elif numLayers in (3, 4):
            wide = self.numPorts >= 16
            for i11, i12, i13, i2 in itertools.product(range(b1 // 2 if wide else b1), range(b2), range(self._i13_len(numLayers)),
                                                       range(2)):
                if wide:
                    if mask[b2 * (2 * i11 - 1) + i12] + mask[b2 * (2 * i11) + i12] + mask[b2 * (2 * i11 + 1) + i12] != '111':
                        continue
                elif mask[b2 * i11 + i12] == '0':
                    continue
                if i2mask[i2] == '0':
                    continue
                yield [i11, i12, i13], i2
        elif numLayers in (5, 6, 7, 8):
            if numLayers <= 6:
                if self.n2 > 1:
                    r1, r2 = b1, b2
                elif self.n1 > 2:
                    r1, r2 = b1, 1
                else:
                    raise ValueError("Unsupported case" + ctx + "!")
            else:
                if (self.n1, self.n2) == (4, 1):
                    r1, r2 = b1 // 2, 1
                elif self.n1 > 4 and self.n2 == 1:
                    r1, r2 = b1, 1
                elif (self.n1, self.n2) == (2, 2):
                    r1, r2 = b1, b2
                elif self.n1 > 2 and self.n2 == 2:
                    r1, r2 = b1, b2 // 2
                elif self.n1 > 2 and self.n2 > 2:
                    r1, r2 = b1, b2
                else:
                    raise ValueError("Unsupported case" + ctx + "!")
            for i11, i12, i2 in itertools.product(range(r1), range(r2), range(2)):
                if mask[b2 * i11 + i12] == '0' or i2mask[i2] == '0':
                    continue
                yield [i11, i12, 0], i2
        else:
            raise ValueError("Unsupported number of layers %d! (codebookType: Type1SP)" % (numLayers))

    def _k12(self, numLayers, i13):
        """(k1, k2) beam offsets: TS 38.214 Table 5.2.2.2.1-3 (two layers) and -4 (three / four layers, < 16 ports)."""
        o1, o2, n1, n2 = self.o1, self.o2, self.n1, self.n2
        if numLayers == 2:
            if i13 < 2:
                return (i13 * o1, 0)
            if n1 > n2 > 1:
                return (0, o2) if i13 == 2 else (2 * o1, 0)
            if n1 == n2:
                return (0, o2) if i13 == 2 else (o1, o2)
            if n1 > 2 and n2 == 1:
                return (i13 * o1, 0)
            raise ValueError("Unsupported N1/N2 combination (N1=%d, N2=%d)!" % (n1, n2))
        if i13 == 0:
            return (o1, 0)
        tab = {(4, 1): [(2 * o1, 0), (3 * o1, 0)], (6, 1): [(2 * o1, 0), (3 * o1, 0), (4 * o1, 0)],
               (2, 2): [(0, o2), (o1, o2)], (3, 2): [(0, o2), (o1, o2), (2 * o1, 0)]}.get((n1, n2), [])
        if i13 - 1 < len(tab):
            return tab[i13 - 1]
        raise ValueError("Unsupported N1/N2 combination (i1,3=%d, N1=%d, N2=%d)!" % (i13, n1, n2))

    def getType1SpPrecoder(self, numLayers, i1=0, i2=0):
        """The (numPorts, numLayers) precoder of PMI (i1 = [i11, i12, i13], i2) (csifeedback.py:724-1037)."""
        if not (isinstance(i1, (tuple, list)) and len(i1) == 3):
            raise ValueError("'i1' must be a tuple or list of length 3!")
        i11, i12, i13 = i1
        b1, b2 = self.n1 * self.o1, self.n2 * self.o2
        ctx = " with %d layers, CB Mode %d, %dx%d Ant" % (numLayers, self.codebookMode, self.n1, self.n2)
        if self.numPorts == 2:                                                  # Table 5.2.2.2.1-1
            if numLayers == 1:
                _check('i11', i11, (0, 3), ctx)
                return np.array([[1], [[1, 1j, -1, -1j][i11]]]) / np.sqrt(2)
            if numLayers == 2:
                _check('i11', i11, [0, 1], ctx)
                return np.array([[1, 1], [1, -1]] if i11 == 0 else [[1, 1], [1j, -1j]]) / 2
            raise ValueError("'numLayers' must be 1 or 2 when 'numPorts' is 2!")
        if self.n2 > 1:
            raise NotImplementedError("Type-I single-panel precoders for N2 > 1: the reference stacks (N1, N2) beam matrices "
                                      "into a (2 N1, N2 x layers) array there, which its own getSINR rejects")
        if numLayers >= 6:
            raise NotImplementedError("%d layers: the reference stops on an undefined attribute (csifeedback.py:944)" % numLayers)
        mode2 = self.codebookMode == 2
        _check('i11', i11, (0, (b1 // 2 if (mode2 and numLayers <= 2) or (numLayers in (3, 4) and self.numPorts >= 16) else b1) - 1), ctx)
        _check('i12', i12, 0, ctx) if numLayers <= 2 and mode2 else _check('i12', i12, (0, b2 - 1), ctx)
        phi = lambda n: np.exp(1j * np.pi * n / 2)                              # noqa: E731
        if numLayers <= 2:
            n_i2 = (4 if not mode2 else 16) if numLayers == 1 else (2 if not mode2 else 8)
            _check('i2', i2, (0, n_i2 - 1), ctx)
            per = n_i2 // 4
            l, n = (2 * i11 + i2 // per, i2 % per) if mode2 else (i11, i2)
            if numLayers == 1:
                vl = self.v(l, 0)
                return np.concatenate([vl, phi(n) * vl]) / np.sqrt(self.numPorts)
            _check('i13', i13, (0, self._i13_len(2) - 1), ctx)
            k1, _ = self._k12(2, i13)
            vl, vp = self.v(l, 0), self.v(l + k1, 0)
            return np.concatenate([np.concatenate([vl, vp], -1), np.concatenate([phi(n) * vl, -phi(n) * vp], -1)]) / \
                np.sqrt(2 * self.numPorts)
        _check('i2', i2, [0, 1], ctx)
        ph = phi(i2)
        if numLayers in (3, 4):
            _check('i13', i13, (0, self._i13_len(numLayers) - 1), ctx)
            if self.numPorts < 16:                                              # Tables 5.2.2.2.1-7/-8, first table
                k1, _ = self._k12(numLayers, i13)
                vl, vp = self.v(i11, i12), self.v(i11 + k1, i12)
                top = [vl, vp, vl] + ([vp] if numLayers == 4 else [])
                bot = [ph * vl, ph * vp, -ph * vl] + ([-ph * vp] if numLayers == 4 else [])
                return np.concatenate([np.concatenate(top, -1), np.concatenate(bot, -1)]) / np.sqrt(numLayers * self.numPorts)
            vt, th = self.v(i11, i12, '~'), np.exp(1j * np.pi * i13 / 4)       # second table: half-length beams, co-phased halves
            sg = np.float64([[1, 1, 1, 1], [1, -1, 1, -1], [1, 1, -1, -1], [1, -1, -1, 1]])[:, :numLayers]
            blocks = [1, th, ph, th * ph]
            return np.concatenate([np.concatenate([blocks[r] * sg[r, c] * vt for c in range(numLayers)], -1)
                                   for r in range(4)]) / np.sqrt(numLayers * self.numPorts)
        # five layers, Table 5.2.2.2.1-9 (N2 = 1, N1 > 2)
        if self.n1 <= 2:
            raise ValueError("Unsupported case for numLayers=%d: N1=%d, N2=%d" % (numLayers, self.n1, self.n2))
        _check('i12', i12, 0, ctx)
        vl, vp, vs = self.v(i11, 0), self.v(i11 + self.o1, 0), self.v(i11 + 2 * self.o1, 0)
        return np.concatenate([np.concatenate([vl, vl, vp, vp, vs], -1),
                               np.concatenate([ph * vl, -ph * vl, vp, -vp, vs], -1)]) / np.sqrt(5 * self.numPorts)

    def getCodebook(self, numLayers):
        """-> ([[i1, i2], ...], codebook (Ncb, numPorts, numLayers)) (csifeedback.py:563-576)."""
        if self.codebookType != 'Type1SP':
            raise NotImplementedError("codebookType '%s': only the Type-I single-panel codebooks run in the reference" %
                                      (self.codebookType))
        idx, cb = [], []
        for i1, i2 in self.type1SpIndexes(numLayers):
            idx.append([i1, i2])
            cb.append(self.getType1SpPrecoder(numLayers, i1, i2))
        return idx, np.array(cb)

    # ------------------------------------------------------------------------------------------------- the search
    def getSINR(self, h, w, noiseVar):
        """Post-MMSE SINR per (codebook entry, channel sample, layer): h (L,K,Nr,Nt) or (n,Nr,Nt), w (Ncb,Nt,Nl) ->
        (Ncb, n, Nl) (csifeedback.py:419-433) -- on the GPU."""
        h = np.complex128(h).reshape(-1, h.shape[-2], h.shape[-1])
        return N(ops.csi_sinr(D(h), D(np.complex128(w)), noiseVar))

    def bestPmiForRank(self, channel, numLayers, noiseVar):
        """Wideband i1 (+ i2) maximising the summed SINR over the CSI-RS REs and layers, then per sub-band the best i2
        among the entries that share that i1 (csifeedback.py:450-514).
        -> ([i1, [i2 per sub-band]], [W per sub-band], [SINR (REs, layers) per sub-band])."""
        grid = self.bwp.createGrid(self.numPorts)
        self.csiRsConfig.populateGrid(grid)
        p, l, k = grid.getReIndexes("CSIRS_NZP")
        res = self.removeNeighbors((l[p == 0], k[p == 0]))
        h_re = channel[res]                                                     # (REs, Nr, Nt)
        idx, cb = self.getCodebook(numLayers)
        sinr = self.getSINR(h_re, cb, noiseVar)                                 # (Ncb, REs, Nl)
        best = int(sinr.sum((1, 2)).argmax())
        wb_i1, wb_i2 = idx[best]
        if self.prgSize is None:
            sb = self.subbandSizePmi if self.bwp.numRbs >= 24 else 0
        else:
            sb = self.prgSize
        if sb == 0:
            return [wb_i1, [wb_i2]], [[cb[best]]], [sinr[best]]
        same_i1 = [i for i, (i1, _) in enumerate(idx) if np.all(np.asarray(i1) == np.asarray(wb_i1))]
        i2s, ws, sb_sinr = [], [], []
        rb = 0
        for n, size in enumerate(self.subbands(sb)):
            sel = np.flatnonzero((res[1] >= rb * 12) & (res[1] < (rb + size) * 12))
            if sel.size == 0:
                raise ValueError(f"Invalid CSI-RS config. Subband {n} does not have any CSI-RS REs!")
            cand = sinr[same_i1][:, sel, :]
            j = int(cand.sum((1, 2)).argmax())
            i2s.append(idx[same_i1[j]][1])
            ws.append(cb[same_i1[j]])
            sb_sinr.append(cand[j])
            rb += size
        return [wb_i1, i2s], ws, sb_sinr

    def getBestRank(self, channel, noiseVar):
        """Rank with the largest summed per-layer SINR among those ``cbRiRestriction`` allows (csifeedback.py:517-536).
        -> (rank, PMI, [SINR (REs, layers) per sub-band])."""
        _, _, nr, nt = channel.shape
        if nt != self.numPorts:
            raise ValueError("The given numver of transmit antenna from channel must mach the number of ports!")
        max_rank = {'Type1SP': min(nr, nt, 8), 'Type1MP': min(nr, 4), 'Type2': min(nr, 2)}[self.codebookType]
        best = (-100000, 0, None, None)
        for rank in range(1, max_rank + 1):
            if self.cbRiRestriction[-rank] != '1':
                continue
            pmi, _, sb_sinr = self.bestPmiForRank(channel, rank, noiseVar)
            per_sb = np.float64([s.mean(0) for s in sb_sinr])                  # (sub-bands, layers)
            total = (per_sb.mean(0) * rank).sum()
            if total > best[0]:
                best = (total, rank, pmi, sb_sinr)
        return best[1], best[2], best[3]
