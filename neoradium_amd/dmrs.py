"""PDSCH DMRS configuration and pilot generation (host side; reference dmrs.py:184-551).

DMRS values depend only on (configuration, slotNoInFrame, symbol), so they are generated on the host once per
slot number and cached; the kernels consume them as pilot tables (ops.chest_ls) or as a pre-filled grid template
(engine).  PTRS (dmrs.py:554-797) is outside the PDSCH link path and not built.
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
        return False

    def setPTRS(self, **kwargs):
        raise NotImplementedError("PTRS is not built in neoradium_amd (outside the PDSCH link-level hot path)")

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


class PTRS:
    def __init__(self, *a, **k):
        raise NotImplementedError("PTRS is not built in neoradium_amd (outside the PDSCH link-level hot path)")
