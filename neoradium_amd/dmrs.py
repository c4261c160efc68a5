"""PDSCH DMRS configuration and pilot generation (host side; reference dmrs.py:184-551).

DMRS values depend only on (configuration, slotNoInFrame, symbol), so they are generated on the host once per
slot number and cached; the kernels consume them as pilot tables (ops.chest_ls) or as a pre-filled grid template
(engine).  PTRS (dmrs.py:554-797): configuration and insertion, like the reference (which has no PTRS-based estimator).
"""
import numpy as np

from .utils import goldBits, toLinear


def _positions(spec):
    """'ld: a|b|c|d' rows -> {ld: [list per additionalPos]} (TS 38.211 Tables 7.4.1.1.2-3 / -4, l-bar values)."""
    out = {}
    for row in spec.strip().split(';'):
        lds, cols = row.split(':')
        cols = [[int(v) for v in c.split(',')] if c.strip() else [] for c in cols.split('|')]
        for ld in lds.split(','):
            out[int(ld)] = cols
    return out


# l0 is written as 0 and replaced by dmrs-TypeA-Position for mapping type A.  ld = 13/14 use l1 = 11 as the
# reference does (dmrs.py:74-79).
_POS = {
    (1, 'A'): _positions("0,1,2:|||; 3,4,5,6,7:0|0|0|0; 8,9:0|0,7|0,7|0,7; 10,11:0|0,9|0,6,9|0,6,9;"
                         "12:0|0,9|0,6,9|0,5,8,11; 13,14:0|0,11|0,7,11|0,5,8,11"),
    (1, 'B'): _positions("0,1,14:|||; 2,3,4:0|0|0|0; 5,6,7:0|0,4|0,4|0,4; 8:0|0,6|0,3,6|0,3,6;"
                         "9,10:0|0,7|0,4,7|0,4,7; 11:0|0,8|0,4,8|0,3,6,9; 12,13:0|0,9|0,5,9|0,3,6,9"),
    (2, 'A'): _positions("0,1,2,3:|||; 4,5,6,7,8,9:0|0||; 10,11,12:0|0,8||; 13,14:0|0,10||"),
    (2, 'B'): _positions("0,1,2,3,4,14:|||; 5,6,7:0|0||; 8,9:0|0,5||; 10,11:0|0,7||; 12,13:0|0,8||"),
}


def _weights(configType, port):
    """w_f(k'), w_t(l') of TS 38.211 Tables 7.4.1.1.2-1/-2 (incl. the Rel-18 enhanced ports; dmrs.py:139-181)."""
    p = port % 100
    s = -1 if p % 2 else 1
    if configType == 1:
        enhanced_half = p >= 8
        wt = [1, 1] if (p % 8) < 4 else [1, -1]
    else:
        enhanced_half = p >= 12
        wt = [1, 1] if (p % 12) < 6 else [1, -1]
    wf = [1, s, -1, -s] if enhanced_half else [1, s, 1, s]
    return wf, wt


class DMRS:
    def __init__(self, pxxch=None, **kwargs):
        self.pxxch = pxxch
        self.configType = kwargs.get('configType', 1)
        if self.configType not in [1, 2]:
            raise ValueError("Invalid DMRS 'configType' value! (It must be 1 or 2)")
        self.enhanced = kwargs.get('enhanced', False)
        self.symbols = kwargs.get('symbols', 1)
        if self.symbols not in [1, 2]:
            raise ValueError("Invalid DMRS 'symbols' value! (It must be 1 or 2)")
        self.typeA1stPos = kwargs.get('typeA1stPos', 2)
        if self.typeA1stPos not in [2, 3]:
            raise ValueError("Invalid 'typeA1stPos' value! (It must be 2 or 3)")
        s0 = pxxch.symSet[0]
        if (s0 not in [0, 1, 2]) and (s0 != 3 or self.typeA1stPos != 3):
            raise ValueError("Invalid symbol allocation: start = %d" % (s0))
        self.additionalPos = kwargs.get('additionalPos', 0)
        if self.symbols == 1:
            if self.additionalPos not in range(4):
                raise ValueError("Invalid 'additionalPos' value! (It must be in [0..3])")
        elif self.additionalPos not in [0, 1]:
            raise ValueError("Invalid 'additionalPos' value! (It must be 0 or 1 for 2-symbol DMRS)")
        ports = kwargs.get('portSet', self.pxxch.portSet)
        if len(ports) != pxxch.numLayers:
            raise ValueError("The number of ports in 'portSet' must match the number of layers (%d)" % (pxxch.numLayers))
        nvalid = (4 if self.symbols == 1 else 8) if self.configType == 1 else (6 if self.symbols == 1 else 12)
        for p in ports:
            if p not in range(nvalid):
                raise ValueError("Invalid DMRS 'port number' %d! (Valid Range: %d..%d)" % (p, 0, nvalid - 1))
        self.pxxch.portSet = ports
        if self.pxxch.numLayers > nvalid:
            raise ValueError("Invalid DMRS 'symbols' specified (%d) for a %d-layer PDSCH!" % (self.symbols, self.pxxch.numLayers))
        ngroups = 2 if self.configType == 1 else 3
        self.cdmGroups = [(p // 2) % ngroups for p in ports]
        self.deltaShifts = self.cdmGroups if self.configType == 1 else [2 * g for g in self.cdmGroups]
        other = kwargs.get('otherCdmGroups', [])
        for cdm in other:
            if cdm in self.cdmGroups:
                raise ValueError("Invalid 'otherCdmGroups' value (%d)! It is already used by this PDSCH." % (cdm))
            if cdm not in range(ngroups):
                raise ValueError("Invalid 'otherCdmGroups' value (%d)! Valid CDM groups are %s." %
                                 (cdm, "0 and 1" if ngroups == 2 else "0, 1, and 2"))
        self.allCdmGroups = sorted(list(set(self.cdmGroups)) + other)
        self.dataREs = []
        self.nIDs = kwargs.get('nIDs', [])
        self.scID = kwargs.get('scID', 0)
        if self.scID not in [0, 1]:
            raise ValueError("Invalid 'scID' value! (It must be 0 or 1)")
        self.sameSeq = kwargs.get('sameSeq', True)
        self.lBar, self.symSet = self.getSymSet()
        # PDSCH-to-DMRS EPRE ratio, TS 38.214 Table 4.1-1 (0 / -3 / -4.77 dB for 1 / 2 / 3 CDM groups without data)
        self.epreRatioDb = kwargs.get('epreRatioDb', [0, -3, -4.77][max(self.allCdmGroups)])
        self.ptrs = None
        self._cache = {}

    @property
    def ptrsEnabled(self):
        return False if self.ptrs is None else (self.ptrs.timeDensity != 0)

    def setPTRS(self, **kwargs):
        """Attach phase-tracking reference signals (dmrs.py PTRS, TS 38.211 7.4.1.2) to this DMRS."""
        self.ptrs = PTRS(self, **kwargs)
        self._cache = {}

    def getSymSet(self):
        """DMRS symbol positions, TS 38.211 7.4.1.1.2 (dmrs.py:390-428)."""
        px = self.pxxch
        if len(px.symSet) == 0:
            return [], []
        if px.mappingType == 'A':
            ld = int(px.symSet[-1]) + 1
            if self.additionalPos == 3:
                assert self.typeA1stPos == 2, "Unsupported combination of 'additionalPos' and 'typeA1stPos'!"
            if ld in [2, 3]:
                assert self.typeA1stPos == 2, "Unsupported combination of 'ld' and 'typeA1stPos'!"
            lbar = np.int32(_POS[(self.symbols, 'A')][ld][self.additionalPos])
            syms = np.int32([self.typeA1stPos] + lbar[1:].tolist())
        else:
            ld = int(px.symSet[-1]) - int(px.symSet[0]) + 1
            if ld == 7:
                assert px.bwp.cpType == 'normal', "Unsupported configuration: ld=7 with extended cyclic prefix!"
            if ld == 6:
                assert px.bwp.cpType == 'extended', "Unsupported configuration: ld=6 with normal cyclic prefix!"
            lbar = np.int32(_POS[(self.symbols, 'B')][ld][self.additionalPos])
            syms = lbar + int(px.symSet[0])
        if self.symbols == 2:
            lbar = np.int32([l + d for l in lbar for d in (0, 1)])
            syms = np.int32([l + d for l in syms for d in (0, 1)])
        keep = [i for i, l in enumerate(syms) if l in px.symSet]
        return lbar[keep], syms[keep]

    def _baseREs(self):
        return np.arange(0, 11, 2) if self.configType == 1 else np.int32([0, 1, 6, 7])

    def getUnusedREs(self):
        base = self._baseREs()
        used = set(base.tolist())
        for sh in list(self.deltaShifts) + (self.configType * np.int32(self.allCdmGroups)).tolist():
            used.update((base + sh).tolist())
        return [x for x in range(12) if x not in used]

    def _sequence(self, l, cdmGroup, numBits):
        """r(n) of TS 38.211 7.4.1.1.1 for symbol l of the current slot (dmrs.py:502-519)."""
        bwp = self.pxxch.bwp
        if self.sameSeq:
            nscid, lam = self.scID, 0
        else:
            nscid, lam = (self.scID if cdmGroup in [0, 2] else 1 - self.scID), cdmGroup
        nid = self.nIDs[nscid] if len(self.nIDs) > nscid else bwp.cellId
        cinit = ((1 << 17) * (bwp.symbolsPerSlot * bwp.slotNoInFrame + int(l) + 1) * (2 * nid + 1) +
                 (1 << 17) * (lam // 2) + 2 * nid + nscid) & 0x7FFFFFFF
        b = goldBits(cinit, numBits).astype(np.float64)
        pair = (1 - 2 * b).reshape(-1, 2) / np.sqrt(2)
        return pair[:, 0] + 1j * pair[:, 1]

    def _portValues(self, p, li, l, rbs):
        """(subcarriers, values) of port index p on DMRS symbol l for the PRBs ``rbs`` (dmrs.py:520-541)."""
        base = self._baseREs()
        n = len(base)
        bwp = self.pxxch.bwp
        off = bwp.startRb * n                                  # sequence starts at CRB 0
        r = self._sequence(l, self.cdmGroups[p], 2 * (off + bwp.numRbs * n))[off:]
        wf, wt = _weights(self.configType, self.pxxch.portSet[p])
        kp = np.arange(n) % (4 if self.enhanced else 2)
        lp = 0 if self.symbols == 1 else li % 2
        beta = toLinear(-self.epreRatioDb / 2)
        rbs = np.asarray(rbs, dtype=np.int64)
        k = (12 * rbs[:, None] + base[None, :] + self.deltaShifts[p]).reshape(-1)
        v = (beta * np.float64(wf)[kp][None, :] * wt[lp]) * r[(rbs[:, None] * n + np.arange(n)[None, :])]
        return k, v.reshape(-1)

    def populateGrid(self, grid):
        """Write the DMRS values and the NO_DATA marks into ``grid`` (dmrs.py:458-551)."""
        slotMap = self.pxxch.slotMap
        base = self._baseREs()
        noData = (self.configType * np.int32(self.allCdmGroups)).tolist()
        RES, UNA, DM = (grid.retNameToId[n] for n in ("RESERVED", "UNASSIGNED", "DMRS"))
        marked = []
        for p in range(len(self.pxxch.portSet)):
            for li, l in enumerate(self.symSet):
                rbs = slotMap[l]
                if len(rbs) == 0:
                    continue
                k, v = self._portValues(p, li, l, rbs)
                if li == 0 and self.ptrs is not None:           # the PTRS repeats the first DMRS symbol's r(n) (dmrs.py:538-539)
                    self.ptrs.saveDmrsL0Values(self.pxxch.portSet[p], k, self._rawValues(p, l, rbs))
                cur = grid.reTypeIds[p, l, k]
                bad = ~np.isin(cur, [RES, UNA, DM])
                if bad.any():
                    kk = int(k[bad][0])
                    raise ValueError(f"Trying to allocate the RE at ({p},{l},{kk}) for DMRS," +
                                     f"while it is currently allocated for \"{grid.reTypeAt(p, l, kk)}\"!")
                ok = cur != RES
                grid[p, l, k[ok]] = (v[ok], "DMRS")
                if li == 0:
                    marked += (base + self.deltaShifts[p]).tolist()
                for sh in noData:
                    kn = (12 * np.asarray(rbs)[:, None] + base[None, :] + sh).reshape(-1)
                    free = grid.reTypeIds[p, l, kn] == UNA
                    grid[p, l, kn[free]] = "NO_DATA"
                    if li == 0 and free[:len(base)].any():
                        marked += (base[free[:len(base)]] + sh).tolist()
        self.dataREs = [x for x in range(12) if x not in marked]
        if self.ptrsEnabled:
            self.ptrs.populateGrid(grid)

    def _rawValues(self, p, l, rbs):
        """r(n) of the port's DMRS subcarriers on symbol l without beta / w_f / w_t, in the order of _portValues."""
        n = len(self._baseREs())
        bwp = self.pxxch.bwp
        off = bwp.startRb * n
        r = self._sequence(l, self.cdmGroups[p], 2 * (off + bwp.numRbs * n))[off:]
        rbs = np.asarray(rbs, dtype=np.int64)
        return r[(rbs[:, None] * n + np.arange(n)[None, :])].reshape(-1)

    def getPilots(self):
        """Pilot table of the current slot: (pilots (P,nDs,nK), subcarriers (P,nK) int32, DMRS symbols)."""
        key = self.pxxch.bwp.slotNoInFrame
        if key not in self._cache:
            rbs = self.pxxch.slotMap[self.symSet[0]]
            P = len(self.pxxch.portSet)
            ks, pil = [], []
            for p in range(P):
                rows = []
                for li, l in enumerate(self.symSet):
                    k, v = self._portValues(p, li, l, rbs)
                    rows.append(v)
                ks.append(k)
                pil.append(np.stack(rows))
            self._cache[key] = (np.stack(pil), np.int32(ks), np.int32(self.symSet))
        return self._cache[key]

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title="DMRS Properties:", getStr=False):
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + title + "\n"
        for name, val in (("configType", self.configType), ("nIDs", self.nIDs), ("scID", self.scID),
                          ("sameSeq", int(self.sameSeq)), ("symbols", "Single" if self.symbols == 1 else "Double"),
                          ("typeA1stPos", self.typeA1stPos), ("additionalPos", self.additionalPos),
                          ("cdmGroups", self.cdmGroups), ("deltaShifts", self.deltaShifts),
                          ("allCdmGroups", self.allCdmGroups), ("symSet", self.symSet),
                          ("REs (before shift)", list(self._baseREs())), ("epreRatioDb", f"{self.epreRatioDb} (dB)")):
            s += pad + f"  {name}: {val}\n"
        if getStr:
            return s
        print(s)


# TS 38.211 Table 7.4.1.2.2-1: subcarrier offset k_ref^RE by DMRS configuration type, DMRS port and resourceElementOffset
_PTRS_REF_RE = {1: [[0, 2, 6, 8], [2, 4, 8, 10], [1, 3, 7, 9], [3, 5, 9, 11]],
                2: [[0, 1, 6, 7], [1, 6, 7, 0], [2, 3, 8, 9], [3, 8, 9, 2], [4, 5, 10, 11], [5, 10, 11, 4]]}


class PTRS:
    """Phase-tracking reference signals of a PDSCH (reference dmrs.py:554-797; TS 38.211 7.4.1.2, TS 38.214 5.1.6.3):
    configuration (time / frequency density directly or from the MCS / bandwidth thresholds), the PTRS symbol set, and the
    insertion into the grid (values: the first DMRS symbol's r(n) of the associated port at the same subcarrier).  PTRS REs
    are excluded from the data REs by PDSCH.getReIndexes / getBitSizes.  Like the reference, nothing on the receive side
    uses them (there is no common-phase-error estimator in the reference)."""

    def __init__(self, dmrs, **kwargs):
        self.pxxch = dmrs.pxxch
        self.dmrs = dmrs
        self.mcsi = kwargs.get('mcsi', None)
        self.iMCS = kwargs.get('iMCS', None)
        self.nRBi = kwargs.get('nRBi', None)
        if (self.mcsi is not None) or (self.iMCS is not None) or (self.nRBi is not None):
            if (self.mcsi is None) or (self.iMCS is None) or (self.nRBi is None):
                raise ValueError("The parameters 'mcsi', 'iMCS', and 'nRBi' must all be None or all have valid values.")
            # (dmrs.py:640-641, 649-650: the reference REJECTS Python lists here -- its type test is inverted -- so tuples /
            #  arrays are what works; kept, it is argument checking a notebook can run into)
            if type(self.mcsi) == list or len(self.mcsi) != 3:
                raise ValueError("The parameters 'mcsi' must be a list with 3 values!")
            if self.iMCS < self.mcsi[0]:
                self.timeDensity = self.freqDensity = 0                 # PTRS disabled (TS 38.214 Table 5.1.6.3-1)
            elif self.iMCS < self.mcsi[1]:
                self.timeDensity = 4
            elif self.iMCS < self.mcsi[2]:
                self.timeDensity = 2
            else:
                self.timeDensity = 1
            numRBs = len(self.pxxch.prbSet)
            if type(self.nRBi) == list or len(self.nRBi) != 2:
                raise ValueError("The parameters 'nRBi' must be a list with 2 values!")
            if numRBs < self.nRBi[0]:
                self.timeDensity = self.freqDensity = 0                 # Table 5.1.6.3-2
            elif numRBs < self.nRBi[1]:
                self.freqDensity = 2
            else:
                self.freqDensity = 4
        else:
            self.timeDensity = kwargs.get('timeDensity', 1)
            if self.timeDensity not in [1, 2, 4]:
                raise ValueError("Invalid 'timeDensity' value! (It must be 1, 2, or 4)")
            if self.timeDensity >= len(self.pxxch.symSet):
                self.timeDensity = 0                                    # TS 38.214 5.1.6.3
            self.freqDensity = kwargs.get('freqDensity', 2)
            if self.freqDensity not in [2, 4]:
                raise ValueError("Invalid 'freqDensity' value! (It must be 2 or 4)")
        self.reOffset = kwargs.get('reOffset', 0)
        if self.reOffset in ['00', '01', '10', '11']:
            self.reOffset = {'00': 0, '01': 1, '10': 2, '11': 3}[self.reOffset]
        if self.reOffset not in [0, 1, 2, 3]:
            raise ValueError("Invalid 'reOffset' value! (It must be 0, 1, 2, or 3)")
        self.portSet = kwargs.get('portSet', self.pxxch.portSet[0:1])
        self.dmrsL0Values = {portNo: {} for portNo in self.portSet}
        self.epreRatio = kwargs.get('epreRatio', 0)
        if self.epreRatio not in [0, 1]:
            raise ValueError("Invalid 'epreRatio' value! (It must be 0 or 1)")
        # PTRS symbols: every timeDensity-th symbol, counted from the last DMRS symbol (TS 38.211 7.4.1.2.2, dmrs.py:713-719)
        self.symSet = []
        skip = 0
        if len(self.pxxch.symSet):
            for sym in range(int(self.pxxch.symSet[0]), int(self.pxxch.symSet[-1]) + 1):
                if sym in self.dmrs.symSet:
                    skip = self.timeDensity
                if skip == 0:
                    if sym in self.pxxch.symSet:
                        self.symSet += [sym]
                    skip = self.timeDensity
                skip -= 1

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title="PTRS Properties:", getStr=False):
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + title + "\n"
        if (self.mcsi is not None) or (self.iMCS is not None) or (self.nRBi is not None):
            s += pad + "  MCS1,MCS2,MCS3: %d %d %d\n" % (self.mcsi[0], self.mcsi[1], self.mcsi[2])
            s += pad + "  Imcs: %d\n" % (self.iMCS)
            s += pad + "  Nrb1, Nrb2: %d %d\n" % (self.nRBi[0], self.nRBi[1])
        for name, val in (("timeDensity", "%d" % self.timeDensity), ("freqDensity", "%d" % self.freqDensity),
                          ("reOffset", "%d" % self.reOffset), ("portSet", str(self.portSet)),
                          ("epreRatio", "%d" % self.epreRatio), ("symSet", str(self.symSet))):
            s += pad + f"  {name}: {val}\n"
        if getStr:
            return s
        print(s)

    def saveDmrsL0Value(self, portNo, k, value):
        if portNo not in self.portSet:
            return
        self.dmrsL0Values[portNo][int(k)] = value

    def saveDmrsL0Values(self, portNo, ks, values):
        if portNo not in self.portSet:
            return
        self.dmrsL0Values[portNo].update(zip((int(k) for k in ks), values))

    def populateGrid(self, grid):
        """Write the PTRS values into ``grid`` (dmrs.py:740-797); called by DMRS.populateGrid."""
        px = self.pxxch
        if len(px.symSet) == 0 or len(self.dmrs.symSet) == 0:
            return
        slotMap = px.slotMap
        beta = 1.0
        if self.epreRatio == 0:
            beta = toLinear([0, 3, 4.77, 6, 7, 7.78][len(self.portSet)] / 2)     # TS 38.214 Table 4.1-2
        skip = [grid.retNameToId[n] for n in ("DMRS", "CSIRS_ZP", "CSIRS_NZP", "RESERVED")]
        fine = [grid.retNameToId[n] for n in ("UNASSIGNED", "PTRS")]
        for p, portNo in enumerate(px.portSet):
            if portNo not in self.portSet:
                continue
            refRE = _PTRS_REF_RE[self.dmrs.configType][portNo][self.reOffset]
            for l in self.symSet:
                rbs = sorted(int(r) for r in slotMap[l])     # (interleaving shuffles the RBs of the map)
                numRBs = len(rbs)
                if numRBs == 0:
                    continue
                if (numRBs % self.freqDensity) == 0:
                    refRB = px.rnti % self.freqDensity
                else:
                    refRB = px.rnti % (numRBs % self.freqDensity)
                for kc in range(refRE + 12 * refRB, 12 * numRBs, 12 * self.freqDensity):
                    k = rbs[kc // 12] * 12 + kc % 12
                    cur = grid.reTypeIds[p, l, k]
                    if cur in skip:
                        continue
                    if cur not in fine:
                        raise ValueError(f"Trying to allocate the RE at ({p},{l},{k}) for PTRS," +
                                         f"while it is currently allocated for \"{grid.reTypeAt(p, l, k)}\"!")
                    grid[p, l, k] = (beta * self.dmrsL0Values[portNo][k], "PTRS")
