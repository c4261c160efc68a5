"""PDSCH configuration, resource mapping and (de)modulation entry points (reference pdsch.py:145-1245).

Index building (which REs carry data, layer mapping order) is host logic done with vectorised NumPy; bit
scrambling + QAM mapping + scatter (``populateGrid``) and gather + max-log demapping + descrambling
(``getLLRsFromGrid``) are single fused kernels of libnrx.
"""
import numpy as np

from . import ops
from ._dev import D, N
from .dmrs import DMRS, PTRS            # noqa: F401  (re-exported like the reference)
from .modulation import Modem
from .utils import goldBits, getMultiLineStr

# TS 38.214 Table 5.1.3.2-1 (TBS for N_info <= 3824)
_TBS_TABLE = np.int32(
    list(range(24, 193, 8)) + list(range(208, 385, 16)) + list(range(408, 577, 24)) + list(range(608, 769, 32)) +
    [808, 848, 888, 928, 984, 1032, 1064, 1128, 1160, 1192, 1224, 1256, 1288, 1320, 1352, 1416, 1480, 1544, 1608, 1672,
     1736, 1800, 1864, 1928, 2024, 2088, 2152, 2216, 2280, 2408, 2472, 2536, 2600, 2664, 2728, 2792, 2856, 2976, 3104,
     3240, 3368, 3496, 3624, 3752, 3824])


class ReservedRbSet:
    def __init__(self, *a, **k):
        raise NotImplementedError("ReservedRbSet is not built (it is not functional in the reference v0.4.0 either: "
                                  "pdsch.py:73-77 references undefined names)")


class PDSCH:
    def __init__(self, bwp, **kwargs):
        self.bwp = bwp
        self.mappingType = kwargs.get('mappingType', 'A')
        assert self.mappingType in "AB", "Unsupported mapping type \"%s\"!" % (self.mappingType)
        self.numLayers = kwargs.get('numLayers', 1)
        assert self.numLayers in range(1, 9), "Number of Layers must be between 1 and 8!"
        self.numCW = 2 if self.numLayers > 4 else 1
        self.reservedRbSets = kwargs.get('reservedRbSets', [])
        if len(self.reservedRbSets):
            raise NotImplementedError("reservedRbSets are not built (broken in the reference v0.4.0 as well)")
        self.reservedReMap = kwargs.get('reservedReMap', [])
        modulation = kwargs.get('modulation', '16QAM')
        if isinstance(modulation, str):
            modulation = self.numCW * [modulation]
        elif isinstance(modulation, (list, tuple)):
            modulation = list(modulation)
        else:
            raise ValueError("'modulation' must be a string, a list strings, or a tuple strings. ('%s' is not supported)" %
                             (type(modulation).__name__))
        if len(modulation) < self.numCW:
            modulation = 2 * modulation
        modulation = modulation[:self.numCW]
        for m in modulation:
            if m not in ['QPSK', '16QAM', '64QAM', '256QAM', '1024QAM']:
                raise ValueError("Unsupported modulation \"%s\"!" % (m))
        self.modems = [Modem(modulation[0])]
        if self.numCW > 1:
            self.modems += [self.modems[0] if modulation[0] == modulation[1] else Modem(modulation[1])]

        sliv = kwargs.get('sliv', None)
        symStart, symLen = kwargs.get('symStart', None), kwargs.get('symLen', None)
        if sliv is not None:                               # TS 38.214 5.1.2.1
            s, l = sliv % 14, sliv // 14 + 1
            if s + l > 14:
                s, l = 13 - s, 16 - l
            check = (14 * (l - 1) + s) if l <= 8 else (14 * (14 - l + 1) + (14 - 1 - s))
            assert sliv == check, "Failed to convert SLIV(%d) to start and length values!" % (sliv)
            self.symSet = np.uint32(range(s, s + l))
        elif (symStart is not None) and (symLen is not None):
            self.symSet = np.uint32(range(symStart, symStart + symLen))
        else:
            if self.mappingType == 'A':
                default = range(self.bwp.symbolsPerSlot)
            else:
                default = range(13) if self.bwp.cpType == 'normal' else range(6)
            self.symSet = np.sort(np.uint32(kwargs.get('symSet', default)))
        self.prbSet = np.sort(np.uint32(kwargs.get('prbSet', range(0, self.bwp.numRbs))))
        if self.symSet[-1] > 14 or self.symSet[0] < 0:
            raise ValueError("Invalid 'symSet' values! (They must be in [0..13]")
        if self.prbSet[-1] > self.bwp.numRbs or self.prbSet[0] < 0:
            raise ValueError("Invalid 'prbSet' values! (They must be in [0..%d]" % (self.bwp.numRbs))
        self.interleavingBundleSize = kwargs.get('interleavingBundleSize', 0)
        if self.interleavingBundleSize not in [0, 2, 4]:
            raise ValueError("'interleavingBundleSize' must be 0 (Interleaving disabled), 2, or 4")
        self._slotMap = None
        self.rnti = kwargs.get('rnti', 1)
        self.nID = kwargs.get('nID', 1)
        self.prgSize = kwargs.get('prgSize', 0)
        if self.prgSize not in [0, 2, 4]:
            raise ValueError("'prgSize' must be 0 (Wideband), 2, or 4)")
        self._checkSymbolAllocation()
        self.portSet = list(range(self.numLayers))
        self.dmrs = None
        self.dataIndices = None
        self._scr = {}

    def _checkSymbolAllocation(self):
        """TS 38.214 Table 5.1.2.1-1 valid (S, L) combinations (pdsch.py:403-426)."""
        s, l, m = int(self.symSet[0]), len(self.symSet), self.bwp.symbolsPerSlot
        if self.mappingType == 'A':
            if l not in range(3, m + 1):
                raise ValueError("Invalid symbol allocation: length = %d  ∉ [3..%d]" % (l, m))
            if (s + l) not in range(3, m + 1):
                raise ValueError("Invalid symbol allocation: start+length = %d+%d = %d ∉ [3..%d]" % (s, l, s + l, m))
        elif self.bwp.cpType == 'normal':
            if s not in range(13):
                raise ValueError("Invalid symbol allocation: start = %d ∉ [0..12]" % (s))
            if l not in range(2, 14):
                raise ValueError("Invalid symbol allocation: length = %d ∉ [2..13]" % (l))
            if (s + l) not in range(2, 15):
                raise ValueError("Invalid symbol allocation: start+length = %d+%d = %d ∉ [2..14]" % (s, l, s + l))
        else:
            if s not in range(11):
                raise ValueError("Invalid symbol allocation: start = %d ∉ [0..10]" % (s))
            if l not in [2, 4, 6]:
                raise ValueError("Invalid symbol allocation: length = %d ∉ {2,4,6}" % (l))
            if (s + l) not in range(2, m + 1):
                raise ValueError("Invalid symbol allocation: start+length = %d+%d = %d ∉ [2..12]" % (s, l, s + l))

    # ------------------------------------------------------------------------------------------------ config
    def setDMRS(self, **kwargs):
        self.dmrs = DMRS(self, **kwargs)

    def setPTRS(self, **kwargs):
        if self.dmrs is None:
            raise ValueError("Cannot set PTRS without first defining a DMRS object for this PDSCH!")
        self.dmrs.setPTRS(**kwargs)

    @property
    def slotNo(self): return self.bwp.carrier.slotNo
    @property
    def frameNo(self): return self.bwp.carrier.frameNo
    @property
    def slotNoInFrame(self): return self.bwp.carrier.slotNoInFrame

    @property
    def slotMap(self):
        """Per symbol, the PRBs of this PDSCH in allocation order (pdsch.py:513-527)."""
        if self._slotMap is None:
            prbs = self.getVrbToPrbMapping().tolist()
            self._slotMap = [prbs if s in self.symSet else [] for s in range(self.bwp.symbolsPerSlot)]
        return self._slotMap

    def getVrbToPrbMapping(self):
        """TS 38.211 7.3.1.6 interleaved VRB-to-PRB mapping (pdsch.py:554-580)."""
        L = self.interleavingBundleSize
        if L == 0:
            return self.prbSet
        n = int(np.ceil((self.bwp.numRbs + (self.bwp.startRb % L)) / L))
        R, Cc = 2, n // 2
        f = np.zeros(n, dtype=np.int32)
        f[:R * Cc] = np.arange(R * Cc).reshape(R, Cc).T.reshape(-1)
        f[n - 1] = n - 1
        d0 = self.bwp.startRb % L
        prb = np.int32([j * L + b for j in f for b in range(L)])
        prb = prb[d0:d0 + self.bwp.numRbs] - d0
        return np.int32(prb)[self.prbSet]

    def populateReservedREs(self, grid):
        if len(self.reservedReMap) == 0:
            return
        if len(self.reservedReMap) not in [1, len(self.portSet)]:
            raise ValueError("The reserved REs must be given for exactly 1 or %d ports." % (len(self.portSet)))
        for p in range(len(self.portSet)):
            pm = self.reservedReMap[0] if len(self.reservedReMap) == 1 else self.reservedReMap[p]
            if len(pm) not in [0, 1, self.bwp.symbolsPerSlot]:
                raise ValueError("The reserved REs must be given for exactly 1 or %d symbols." % (self.bwp.symbolsPerSlot))
            for l in range(self.bwp.symbolsPerSlot if len(pm) else 0):
                res = pm[0] if len(pm) == 1 else pm[l]
                if len(res):
                    grid[p, l, np.int32(res)] = "RESERVED"

    # ------------------------------------------------------------------------------------------- scrambling
    def _cinit(self, q):
        return self.rnti * (1 << 15) + q * (1 << 14) + self.nID          # TS 38.211 7.3.1.1

    def _scrambling(self, q, n):
        key = (self._cinit(q),)
        cur = self._scr.get(key)
        if cur is None or len(cur) < n:
            cur = goldBits(self._cinit(q), n).astype(np.uint8)
            self._scr[key] = cur
        return cur[:n]

    def scrambleBits(self, q, bits):
        return np.asarray(bits) ^ self._scrambling(q, len(bits)).astype(np.int8)

    def scrambleLLRs(self, q, llrs):
        return llrs * (1 - 2 * np.float64(self._scrambling(q, len(llrs))))

    # ---------------------------------------------------------------------------------------------- indexes
    def getGrid(self, useReDesc=False):
        grid = self.bwp.createGrid(self.numLayers, useReDesc)
        self.allocateResources(grid)
        return grid

    def _scan(self, grid, wanted_ids):
        """(port, symbol, subcarrier) triples of the REs of this PDSCH whose type is in ``wanted_ids``, in the
        reference's order: port -> symbol -> PRB (allocation order) -> RE (pdsch.py:698-769, :833-852)."""
        ps, ls, ks = [], [], []
        for p in range(len(self.portSet)):
            for sym, rbs in enumerate(self.slotMap):
                if len(rbs) == 0:
                    continue
                k = (12 * np.asarray(rbs, dtype=np.int64)[:, None] + np.arange(12)[None, :]).reshape(-1)
                k = k[np.isin(grid.reTypeIds[p, sym, k], wanted_ids)]
                ps.append(np.full(len(k), p)); ls.append(np.full(len(k), sym)); ks.append(k)
        if not ps:
            return (np.int32([]), np.int32([]), np.int32([]))
        return (np.int32(np.concatenate(ps)), np.int32(np.concatenate(ls)), np.int32(np.concatenate(ks)))

    def allocateResources(self, grid):
        """Mark the data REs of this PDSCH (everything in the allocation that is not DMRS/NO_DATA/reserved...)."""
        self.populateReservedREs(grid)
        if self.dmrs is not None:
            self.dmrs.populateGrid(grid)
        ids = grid.retNameToId
        blocked = [ids[n] for n in ("DMRS", "CSIRS_ZP", "CSIRS_NZP", "RESERVED", "PTRS", "NO_DATA")]
        free = [ids["UNASSIGNED"], ids["PDSCH"]]
        for p in range(len(self.portSet)):
            for sym in self.symSet:
                rbs = self.slotMap[sym]
                k = (12 * np.asarray(rbs, dtype=np.int64)[:, None] + np.arange(12)[None, :]).reshape(-1)
                cur = grid.reTypeIds[p, sym, k]
                other = ~np.isin(cur, blocked + free)
                if other.any():
                    kk = int(k[other][0])
                    raise ValueError(f"Trying to allocate the RE at ({p},{sym},{kk}) for PDSCH," +
                                     f"while it is currently allocated for \"{grid.reTypeAt(p, sym, kk)}\"!")
        idx = self._scan(grid, free)
        grid[idx] = (0, "PDSCH")
        self.dataIndices = idx

    def getReIndexes(self, grid, reTypeStr):
        return self._scan(grid, [grid.retNameToId[reTypeStr]])

    def getNumREsFromIndexes(self, indexes):
        n = len(indexes[0])
        if self.numCW == 1:
            return [n]
        starts = np.append([0], np.where(np.diff(indexes[0]) == 1)[0] + 1)
        n0 = int(starts[self.numLayers // 2])
        return [n0, n - n0]

    def getBitSizes(self, grid, reTypeStr="PDSCH"):
        counts = self.getNumREsFromIndexes(self.getReIndexes(grid, reTypeStr))
        return [counts[i] * self.modems[i].qm for i in range(self.numCW)]

    def getLayerMapIndexes(self, psdchIndexes, numREsInCw=None):
        """TS 38.211 7.3.1.3 layer mapping as an index permutation (pdsch.py:619-639): symbol i of a codeword goes
        to layer i mod v, position i div v."""
        if numREsInCw is None:
            numREsInCw = self.getNumREsFromIndexes(psdchIndexes)
        starts = np.append([0], np.where(np.diff(psdchIndexes[0]) == 1)[0] + 1)
        v1 = self.numLayers if self.numCW == 1 else self.numLayers // 2
        out = []
        for cw, (lay, cnt) in enumerate(((starts[:v1], numREsInCw[0]),) +
                                        (((starts[v1:], numREsInCw[1]),) if self.numCW > 1 else ())):
            v = len(lay)
            n = (cnt + v - 1) // v
            m = (lay[None, :] + np.arange(n)[:, None]).reshape(-1)[:cnt]
            out.append((psdchIndexes[0][m], psdchIndexes[1][m], psdchIndexes[2][m]))
        return out

    def _flat(self, grid, idx):
        """Flat complex-element offsets inside one (P,L,K) grid."""
        _, L, K = grid.shape
        return (np.int64(idx[0]) * L + idx[1]) * K + idx[2]

    # ---------------------------------------------------------------------------------------------- mapping
    def populateGrid(self, grid, bits=None):
        """Scramble, modulate and map the codeword bits onto the grid (pdsch.py:855-932)."""
        if bits is None:
            return
        if isinstance(bits, tuple):
            bits = list(bits)
        elif isinstance(bits, np.ndarray):
            bits = [bits] if bits.ndim == 1 else [bits[i] for i in range(bits.shape[0])]
        elif not isinstance(bits, list):
            raise ValueError("'bits' must be a NumPy array, a tuple of NumPy arrays, or a list of NumPy arrays.")
        if self.numCW != len(bits):
            raise ValueError(f"Number of codewords is {self.numCW} but {len(bits)} set(s) of bits are provided!")
        nre = []
        for cw in range(self.numCW):
            qm = self.modems[cw].qm
            if len(bits[cw]) % qm:
                raise ValueError("The length of 'bitstream' (%d) must be a multiple of 'qm' (%d)!" % (len(bits[cw]), qm))
            nre.append(len(bits[cw]) // qm)
        lm = self.getLayerMapIndexes(self.dataIndices, nre)
        dev_grid = D(grid.grid[None])
        for cw in range(self.numCW):
            qm = self.modems[cw].qm
            b = np.uint8(bits[cw])[None]
            ops.qam_map(D(b), qm, scr=D(self._scrambling(cw, b.shape[1])),
                        re_index=np.int32(self._flat(grid, lm[cw])), out=dev_grid)
            grid.reTypeIds[lm[cw]] = grid.retNameToId["PDSCH"]
        grid.grid = N(dev_grid)[0]
        if grid.reDesc is not None:
            for cw in range(self.numCW):
                grid.reDesc[lm[cw]] = ["CW%d-%d" % (cw, i) for i in range(nre[cw])]

    def getLLRsFromGrid(self, rxGrid, pdschIndexes, llrScales=None, noiseVar=None, useMax=True):
        """Gather, demap (max-log by default), descramble, weight by llrScales (pdsch.py:935-1005)."""
        lm = self.getLayerMapIndexes(pdschIndexes)
        nv = rxGrid.noiseVar if noiseVar is None else noiseVar
        dev_grid = D(np.complex128(rxGrid.grid)[None])
        dev_sc = None if llrScales is None else D(np.float64(llrScales)[None])
        out = []
        for cw in range(self.numCW):
            qm = self.modems[cw].qm
            n = len(lm[cw][0])
            llr = ops.qam_demap(dev_grid, D(np.float64([nv])), qm, scr=D(self._scrambling(cw, n * qm)),
                                re_index=np.int32(self._flat(rxGrid, lm[cw])), scales=dev_sc, exact=not useMax,
                                nv_floor=1e-10)
            out.append(N(llr)[0])
        return out

    def getHardBitsFromGrid(self, rxGrid, pdschIndexes, llrScales=None, noiseVar=None, useMax=True):
        llrs = self.getLLRsFromGrid(rxGrid, pdschIndexes, llrScales, noiseVar, useMax)
        return [np.int8(l < 0) for l in llrs]

    def getDataSymbols(self, grid):
        return grid[self.dataIndices]

    # --------------------------------------------------------------------------------------------- precoding
    def getPrecodingMatrix(self, channelMatrix):
        """SVD precoder per PRG (pdsch.py:1080-1165), host LAPACK like the reference.

        QUIRK kept for parity of BLER curves: a group is closed when the FIRST PRB of the next group arrives and the
        last group is never closed, so the "wideband" precoder is the SVD of the channel averaged over the first PRB
        only, and with prgSize 2/4 the trailing PRBs get no precoder (pdsch.py:1142-1163)."""
        numRBs = channelMatrix.shape[1] // 12
        if numRBs < len(self.prbSet):
            raise ValueError("The number of RBs in the 'channelMatrix' (%d) cannot be less than RBs in the PDSCH (%d)!" %
                             (numRBs, len(self.prbSet)))

        def groupPrecoder(rbs):
            res = np.int32([rb * 12 + re for rb in rbs for re in range(12)])
            mean = channelMatrix[:, res, :, :].mean(axis=(0, 1))
            _, _, vH = np.linalg.svd(mean)
            return (np.conj(vH).T)[:, :self.numLayers] / np.sqrt(self.numLayers)

        f, cur, rbs = [], -1, []
        for prb in self.prbSet:
            group = 0 if self.prgSize == 0 else (int(prb) + self.bwp.startRb) // self.prgSize
            rbs += [int(prb)]
            if group != cur:
                f += [(rbs, groupPrecoder(rbs))]
                cur, rbs = group, []
        if (len(self.prbSet) == numRBs) and (self.prgSize == 0):
            return f[0][1]
        return f

    # --------------------------------------------------------------------------------------------------- TBS
    def getTxBlockSize(self, codeRates, xOverhead=0, scaleFactor=1.0):
        """TS 38.214 5.1.3.2 transport block size per codeword (pdsch.py:1168-1245, float arithmetic kept)."""
        if isinstance(codeRates, (float, np.float32, np.float64)):
            codeRates = [codeRates]
        elif isinstance(codeRates, (list, np.ndarray, tuple)):
            codeRates = list(codeRates)
        else:
            raise ValueError("'codeRates' must be a float value, or a list, tuple, or NumPy array of 1 or 2 float "
                             "values. ('%s' is not supported)" % (type(codeRates).__name__))
        if len(codeRates) < self.numCW:
            codeRates = self.numCW * codeRates
        codeRates = codeRates[:self.numCW]
        if scaleFactor not in [1 / 4, 1 / 2, 1]:
            raise ValueError("'scaleFactor' must be one of: 0.25, 0.5, or 1")
        npRE = 12 * len(self.symSet)
        if self.dmrs is not None:
            npRE -= len(self.dmrs.symSet) * (12 - len(self.dmrs.dataREs))
        assert npRE > 0
        if npRE <= xOverhead:
            raise ValueError("'xOverhead' must be less than %d." % (npRE))
        npRE -= xOverhead
        numREs = min(156, npRE) * len(self.prbSet)
        layers = [self.numLayers] if self.numCW == 1 else [self.numLayers // 2, self.numLayers - self.numLayers // 2]
        out = []
        for c in range(self.numCW):
            nInfo = scaleFactor * numREs * codeRates[c] * self.modems[c].qm * layers[c]
            if nInfo <= 3824:
                n = max(3, int(np.log2(nInfo)) - 6)
                npInfo = max(24, (1 << n) * (nInfo // (1 << n)))
                out += [_TBS_TABLE[_TBS_TABLE >= npInfo][0]]
            else:
                n = int(np.log2(nInfo - 24)) - 5
                npInfo = max(3840, (1 << n) * np.round((nInfo - 24) / (1 << n)))
                if codeRates[c] <= 0.25:
                    eightC = 8 * np.ceil((npInfo + 24) / 3816)
                elif npInfo > 8424:
                    eightC = 8 * np.ceil((npInfo + 24) / 8424)
                else:
                    eightC = 8
                out += [int(eightC * np.ceil((npInfo + 24) / eightC)) - 24]
        return out

    def __repr__(self): return self.print(getStr=True)

    def print(self, indent=0, title="PDSCH Properties:", getStr=False):
        pad = indent * ' '
        s = ("\n" if indent == 0 else "") + pad + title + "\n"
        mods = self.modems[0].modulation
        if len(self.modems) > 1 and self.modems[0].modulation != self.modems[1].modulation:
            mods += ", " + self.modems[1].modulation
        for name, val in (("mappingType", self.mappingType), ("nID", self.nID), ("rnti", self.rnti),
                          ("numLayers", self.numLayers), ("numCodewords", self.numCW), ("modulation", mods),
                          ("portSet", self.portSet)):
            s += pad + f"  {name}: {val}\n"
        s += getMultiLineStr("symSet", self.symSet, indent, "%3d", 3, numPerLine=20)
        s += getMultiLineStr("prbSet", self.prbSet, indent, "%3d", 3, numPerLine=20)
        s += pad + "  interleavingBundleSize: %d\n" % (self.interleavingBundleSize)
        s += pad + "  PRG Size: %s\n" % ("Wideband" if self.prgSize == 0 else str(self.prgSize))
        s += self.bwp.print(indent + 2, "Bandwidth Part:", True)
        if self.dmrs is not None:
            s += self.dmrs.print(indent + 2, "DMRS:", True)
        if getStr:
            return s
        print(s)
