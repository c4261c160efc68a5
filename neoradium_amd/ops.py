"""Batched device-tensor front end of the C ABI (include/nrx.h).

Every function takes/returns torch tensors resident on the GPU, validates shapes on the host (a faulting kernel
can take the whole node down), allocates outputs with torch, and enqueues one libnrx call on the current HIP
stream.  Leading batch dimension = Monte-Carlo slots / transport blocks.
"""
import ctypes as C
import os
import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr, stream

CRC_ID = {'6': 0, '11': 1, '16': 2, '24A': 3, '24B': 4, '24C': 5}
CRC_LEN = {'6': 6, '11': 11, '16': 16, '24A': 24, '24B': 24, '24C': 24}
_FT = {torch.float32: 'f32', torch.float64: 'f64'}


def _dev(t):
    if not t.is_cuda:
        raise ValueError("nrx ops need tensors on the GPU (no CPU fallback)")
    # one process per GPU: launches go to the CURRENT device's stream and libnrx's per-device tables are resolved for the
    # current device, so a tensor living elsewhere must be refused (torch.cuda.set_device / torch.cuda.device(...) first)
    if t.device.index != torch.cuda.current_device():
        raise ValueError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}")
    return t.device


def _u8(t):
    if t.dtype not in (torch.uint8, torch.int8):
        raise ValueError(f"bit tensors must be uint8/int8, got {t.dtype}")
    return t.contiguous()


# ------------------------------------------------------------------------------------------------------- CRC
def crc(bits, poly):
    """chancodebase.py:83-128 getCrc on a (rows, n) or (n,) device bit tensor -> (rows, L) / (L,)."""
    if poly not in CRC_ID:
        raise ValueError(f"Unsupported CRC polynomial '{poly}'")
    flat = bits.dim() == 1
    b = _u8(bits.reshape(1, -1) if flat else bits)
    rows, n = b.shape
    out = torch.empty((rows, CRC_LEN[poly]), dtype=torch.uint8, device=_dev(b))
    check(lib().nrx_crc(ptr(b), rows, n, n, CRC_ID[poly], ptr(out), stream()))
    return out[0] if flat else out


# ------------------------------------------------------------------------------------------------------ LDPC
def ldpc_segment(tb, cfg, add_tb_crc=True):
    """appendCrc('24A') + doSegmentation (ldpc.py:981-1030): (n_tb, A) bits -> (n_tb*C, K)."""
    tb = _u8(tb)
    n_tb, A = tb.shape
    cbs = torch.empty((n_tb * cfg.C, cfg.K), dtype=torch.uint8, device=_dev(tb))
    check(lib().nrx_ldpc_segment(ptr(tb), n_tb, A, 1 if add_tb_crc else 0, C.byref(cfg), ptr(cbs), stream()))
    return cbs


def ldpc_encode(cbs, cfg, puncture=True, rows=None, _reuse=None):
    """ldpc.py:1033-1090 encode: (n_cb, K) -> (n_cb, N) (or N+2Zc without puncturing).  ``rows``: only the parity of the
    first ``rows`` base-graph rows is produced (columns >= 22 + rows (BG1) / 10 + rows (BG2) of the output are zero): for a
    caller that rate-matches rv 0 into fewer bits than that (``ldpc_active_rows``).
    ``_reuse`` is PdschLink's private arrangement, not part of the surface: the (buffer, columns-cleared-from) pair an earlier call
    with the same shape and ``rows`` returned for that link -- the never-written parity columns were cleared then and nothing but
    this function writes the buffer, so they are not cleared again (219 MB per 256 slots at the metric configuration)."""
    cbs = _u8(cbs)
    if cbs.dim() != 2 or cbs.shape[1] != cfg.K:
        raise ValueError(f"code blocks must be (n_cb, K={cfg.K}), got {tuple(cbs.shape)}")
    width = cfg.N if puncture else cfg.N + 2 * cfg.Zc
    kb = 22 if cfg.bg == 1 else 10
    first = min(width, (kb + int(rows) - (2 if puncture else 0)) * cfg.Zc) if rows else width
    reuse = _reuse is not None
    if reuse:
        out = _reuse
        if not isinstance(out, _EncBuf) or out.dtype != torch.uint8 or tuple(out.shape) != (cbs.shape[0], width) or not out.is_contiguous() \
                or _dev(out) != _dev(cbs) or out.cleared_from != first:
            raise ValueError("ldpc_encode: the reused buffer is not the result of an earlier call of the same shape and row count")
    else:
        out = torch.empty((cbs.shape[0], width), dtype=torch.uint8, device=_dev(cbs))
    check(lib().nrx_ldpc_encode(ptr(cbs), cbs.shape[0], C.byref(cfg), 1 if puncture else 0, int(rows or 0), ptr(out), stream()))
    if not reuse:
        # the parity columns of the other rows are not computed: they are ZEROED (a view, a clone or a slice of the result must
        # never expose uninitialised memory), and ldpc_rate_match refuses a transmission that would read them
        if first < width:
            out[:, first:].zero_()
        out = out.as_subclass(_EncBuf)
        out.cleared_from, out.rows_held = first, (int(rows) if rows else None)
    return out


class _EncBuf(torch.Tensor):
    """A coded-bit matrix whose columns >= ``cleared_from`` were zeroed by ldpc_encode (see its ``_reuse``).  The type is the proof:
    any torch operation on it returns a plain tensor, so only the very object ldpc_encode returned can come back."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))


def _rv_array(rv, n_tb, dev):
    if rv.dtype != torch.int32 or rv.numel() != n_tb or rv.device != dev:
        raise ValueError("per-transport-block rv must be an int32 device tensor with one entry per transport block")
    return rv.contiguous()


def ldpc_rate_match(coded, cfg, G, nl, qm, rv=0, nref=0):
    """ldpc.py:1093-1159 rateMatch: (n_tb*C, N) -> (n_tb, sum E_r).  ``rv``: int, or an int32 device tensor with
    one redundancy version per transport block (batched HARQ processes)."""
    part = getattr(coded, 'rows_held', None)               # (a rows-truncated encode's own result, see ldpc_encode)
    coded = _u8(coded)
    if coded.dim() != 2 or coded.shape[1] != cfg.N or coded.shape[0] % cfg.C:
        raise ValueError(f"coded blocks must be (n_tb*C, N={cfg.N}), got {tuple(coded.shape)}")
    n_tb = coded.shape[0] // cfg.C
    f = nl * qm
    gout = ((G + f - 1) // f) * f
    dev = _dev(coded)
    if part:       # a rows-truncated encode holds the parity of its first rows only: rv 0 without wrap-around may read it
        need = ldpc_active_rows(cfg, max(_lib.ldpc_cb_lens(int(G), cfg.C, nl, qm))) if (not torch.is_tensor(rv) and rv == 0 and nref == 0) \
            else (46 if cfg.bg == 1 else 42)
        if need > part:
            raise ValueError(f"ldpc_rate_match: the coded blocks hold the parity of {part} base-graph rows, this transmission reads {need}")
    out = torch.empty((n_tb, gout), dtype=torch.uint8, device=dev)
    if torch.is_tensor(rv):
        check(lib().nrx_ldpc_rate_match_harq(ptr(coded), n_tb, C.byref(cfg), int(G), nl, qm, ptr(_rv_array(rv, n_tb, dev)),
                                             nref, ptr(out), stream()))
    else:
        check(lib().nrx_ldpc_rate_match(ptr(coded), n_tb, C.byref(cfg), int(G), nl, qm, rv, nref, ptr(out), stream()))
    return out


def ldpc_rate_recover(llr, cfg, nl, qm, rv=0, nref=0, circ=None, reset=None):
    """ldpc.py:1330-1418 recoverRate: (n_tb, G) LLRs -> (n_tb*C, N); ``circ`` (n_tb*C, Ncb-F) accumulates in place.
    Batched HARQ: ``rv`` int32 device tensor (one per transport block), ``reset`` uint8 device tensor (non-zero = the
    soft buffer of that transport block restarts from zero)."""
    if llr.dtype not in _FT:
        raise ValueError("LLRs must be float32 or float64")
    llr = llr.contiguous()
    n_tb, G = llr.shape
    ncb = cfg.N if nref == 0 else min(cfg.N, nref)
    if circ is not None:
        if tuple(circ.shape) != (n_tb * cfg.C, ncb - cfg.F) or circ.dtype != llr.dtype:
            raise ValueError(f"HARQ buffer shape mismatch! It must be a {n_tb * cfg.C}x{ncb - cfg.F} {llr.dtype} tensor!")
    dev = _dev(llr)
    out = torch.empty((n_tb * cfg.C, cfg.N), dtype=llr.dtype, device=dev)
    if torch.is_tensor(rv):
        if circ is None:
            raise ValueError("per-transport-block rv needs the HARQ soft buffer `circ`")
        if reset is not None and (reset.dtype != torch.uint8 or reset.numel() != n_tb):
            raise ValueError("reset must be a uint8 device tensor with one entry per transport block")
        fn = getattr(lib(), 'nrx_ldpc_rate_recover_harq_' + _FT[llr.dtype])
        check(fn(ptr(llr), n_tb, G, C.byref(cfg), nl, qm, ptr(_rv_array(rv, n_tb, dev)),
                 ptr(None if reset is None else reset.contiguous()), nref, ptr(circ), ptr(out), stream()))
        return out
    if reset is not None:
        raise ValueError("reset goes with a per-transport-block rv tensor")
    fn = getattr(lib(), 'nrx_ldpc_rate_recover_' + _FT[llr.dtype])
    check(fn(ptr(llr), n_tb, G, C.byref(cfg), nl, qm, rv, nref, ptr(circ), ptr(out), stream()))
    return out


_ws_cache = {}


def _decode_ws(cfg, device):
    need = lib().nrx_ldpc_decode_ws_bytes(C.byref(cfg), 1)
    key = (device, cfg.bg, stream())            # one workspace per stream: launches on different streams may overlap
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def ldpc_active_rows(cfg, e_max, rv=0, accumulated=False):
    """Rows of the base graph that can change an information bit when at most ``e_max`` rate-matched bits per code block
    were received in ONE transmission with redundancy version 0 into an empty soft buffer (ldpc.py:1093-1159, 1330-1418):
    the transmitted bits fill the circular buffer (fillers skipped) from position 0, i.e. punctured-codeword columns
    0 .. (F + e_max - 1) // Zc; the extension column of row r is base-graph column r + 22.  Everything else (rv != 0,
    HARQ accumulation, wrap-around) needs all rows."""
    rows_all = 46 if cfg.bg == 1 else 42
    if rv != 0 or accumulated or e_max >= cfg.N - cfg.F:
        return rows_all
    last = e_max - 1 + (cfg.F if e_max > cfg.K - 2 * cfg.Zc - cfg.F else 0)      # position in the punctured code word
    col = last // cfg.Zc + 2                                                     # base-graph column
    core = 26 if cfg.bg == 1 else 14
    return min(rows_all, max(4, col - core + 1 + 4))


def ldpc_decode(llr, cfg, n_iter=5, only_info=True, belief=False, rows=None):
    """ldpc.py:1495-1581 decode: (n_cb, N) LLRs -> (n_cb, K or N+2Zc) hard bits (uint8) or beliefs.

    float64 input runs the bit-exact float64 kernel, float32 input the single-precision throughput kernel.
    ``rows`` (hard information bits only): decode with the first `rows` rows of the base graph; identical output when
    the dropped rows' extension columns are all-zero (see nrx_ldpc_decode_rows_* and :func:`ldpc_active_rows`)."""
    if llr.dtype not in _FT:
        raise ValueError("LLRs must be float32 or float64")
    llr = llr.contiguous()
    if llr.dim() != 2 or llr.shape[1] != cfg.N:
        raise ValueError(f"rate-recovered LLRs must be (n_cb, N={cfg.N}), got {tuple(llr.shape)}")
    n_cb = llr.shape[0]
    cols = cfg.K if only_info else cfg.N + 2 * cfg.Zc
    dev = _dev(llr)
    hard = None if belief else torch.empty((n_cb, cols), dtype=torch.uint8, device=dev)
    bel = torch.empty((n_cb, cols), dtype=llr.dtype, device=dev) if belief else None
    if rows is not None and int(rows) < (46 if cfg.bg == 1 else 42):
        if belief or not only_info:
            raise ValueError("rows: only the hard decisions of the information bits are defined with dropped rows")
        if llr.dtype == torch.float64:
            ws = _decode_ws(cfg, dev)
            check(lib().nrx_ldpc_decode_rows_f64(ptr(llr), n_cb, C.byref(cfg), int(n_iter), int(rows), ptr(hard), ptr(ws),
                                                 ws.numel(), stream()))
        else:
            check(lib().nrx_ldpc_decode_rows_f32(ptr(llr), n_cb, C.byref(cfg), int(n_iter), int(rows), ptr(hard), None, 0,
                                                 stream()))
        return hard
    if llr.dtype == torch.float64:
        ws = _decode_ws(cfg, dev)
        check(lib().nrx_ldpc_decode_f64(ptr(llr), n_cb, C.byref(cfg), int(n_iter), cols, ptr(hard), ptr(bel), ptr(ws),
                                        ws.numel(), stream()))
    else:
        check(lib().nrx_ldpc_decode_f32(ptr(llr), n_cb, C.byref(cfg), int(n_iter), cols, ptr(hard), ptr(bel), None, 0,
                                        stream()))
    return bel if belief else hard


def select_zero(flags):
    """Indices of the zero entries of a uint8 flag vector, ascending, and their number: (sel int32 (n,), n_sel int32 (1,)), both
    on the device (nrx_select_failed): the work list of the *_sel entries, never read by the host."""
    flags = _u8(flags).reshape(-1)
    n = flags.numel()
    dev = _dev(flags)
    sel = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    n_sel = torch.empty(1, dtype=torch.int32, device=dev)
    check(lib().nrx_select_failed(ptr(flags), n, ptr(sel), ptr(n_sel), stream()))
    return sel, n_sel


def ldpc_decode_selected(llr, cfg, n_iter, rows, sel, n_sel, out):
    """ldpc_decode(rows=...) of the code blocks sel[0 .. n_sel) only (device list, device count), hard bits written into the
    matching rows of ``out`` (n_cb, K) uint8.  False when this configuration has no selection-capable kernel (float64,
    BG1, Zc 384): the caller then gathers and decodes the rows itself."""
    if llr.dtype != torch.float64 or not (cfg.bg == 1 and cfg.Zc == 384):
        return False
    llr = llr.contiguous()
    n_cb = llr.shape[0]
    ws = _decode_ws(cfg, _dev(llr))
    rc = lib().nrx_ldpc_decode_rows_sel_f64(ptr(llr), n_cb, C.byref(cfg), int(n_iter), int(rows), ptr(out), ptr(ws), ws.numel(),
                                            ptr(sel), ptr(n_sel), stream())
    if rc == -3:
        return False
    check(rc)
    return True


def ldpc_recover_decode_merge(llr, cfg, nl, qm, n_iter, rows=0):
    """recoverRate (first transmission) + decode + checkCrcAndMerge in one launch (nrx_ldpc_recover_decode_merge_f64):
    (n_tb, G) float64 LLRs in the per-code-block de-interleaved layout of ``qam_demap(code_blocks=(C, nl))`` ->
    (tb_out (n_tb, C*(cb_len-24)), cb_ok (n_tb, C)), bit-identical to ldpc_rate_recover (on the symbol-major LLRs) ->
    ldpc_decode(rows=...) -> ldpc_crc_merge.  Returns None when this configuration has no fused instantiation (the caller
    then runs the three separate stages on symbol-major LLRs)."""
    if llr.dtype != torch.float64 or llr.dim() != 2:
        return None
    if not (cfg.bg == 1 and cfg.Zc == 384 and cfg.C > 1):          # (cheap pre-check; the library decides)
        return None
    llr = llr.contiguous()
    n_tb, G = llr.shape
    dev = _dev(llr)
    tb_out = torch.empty((n_tb, cfg.C * (cfg.cb_len - 24)), dtype=torch.uint8, device=dev)
    cb_ok = torch.empty((n_tb, cfg.C), dtype=torch.uint8, device=dev)
    rc = lib().nrx_ldpc_recover_decode_merge_f64(ptr(llr), n_tb, G, C.byref(cfg), nl, qm, int(n_iter), int(rows or 0),
                                                 ptr(tb_out), ptr(cb_ok), stream())
    if rc == -3:                                                   # NRX_E_UNSUPPORTED
        return None
    check(rc)
    return tb_out, cb_ok


_fused_state = {}       # (device index) -> parked-state buffer of the continuation form, grown on demand and reused


def ldpc_recover_decode_merge_two_pass(llr, cfg, nl, qm, first_iter, n_iter, rows=0, restart=False, stages=None):
    """The opt-in two-pass schedule on the fused entry: every code block decoded with ``first_iter`` iterations; the ones whose
    CRC24B fails CONTINUE from their parked decoder state for the remaining ``n_iter - first_iter`` (for them the result is
    exactly that of one run of ``n_iter`` iterations, and no iteration is done twice), or -- ``restart=True`` -- are decoded
    again from scratch with ``n_iter``.  ``stages`` = ascending iteration counts between the two (continuation form only): the
    failing blocks are checked again at each of them.  The list of failing blocks and its length stay on the device
    (nrx_select_failed): no host read, no gathers.  -> (tb_out, cb_ok) or None when the configuration has no fused instantiation."""
    if llr.dtype != torch.float64 or llr.dim() != 2 or not (cfg.bg == 1 and cfg.Zc == 384 and cfg.C > 1):
        return None
    llr = llr.contiguous()
    n_tb, G = llr.shape
    dev = _dev(llr)
    n_cb = n_tb * cfg.C
    if restart:
        first = ldpc_recover_decode_merge(llr, cfg, nl, qm, first_iter, rows=rows)
        if first is None:
            return None
        tb_out, cb_ok = first
        state = None
    else:
        per = int(lib().nrx_ldpc_fused_state_bytes(C.byref(cfg), nl, qm, G, int(rows or 0)))
        if per == -3:
            return None
        if per < 0:
            check(per)
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        state = _fused_state.get(key)
        if state is None or state.numel() < n_cb * per:
            state = _fused_state[key] = torch.empty(n_cb * per, dtype=torch.uint8, device=dev)
        tb_out = torch.empty((n_tb, cfg.C * (cfg.cb_len - 24)), dtype=torch.uint8, device=dev)
        cb_ok = torch.empty((n_tb, cfg.C), dtype=torch.uint8, device=dev)
        check(lib().nrx_ldpc_recover_decode_merge_park_f64(ptr(llr), n_tb, G, C.byref(cfg), nl, qm, int(first_iter), int(rows or 0),
                                                           ptr(tb_out), ptr(cb_ok), ptr(state), stream()))
    sel = torch.empty(n_cb, dtype=torch.int32, device=dev)
    n_sel = torch.empty(1, dtype=torch.int32, device=dev)
    check(lib().nrx_select_failed(ptr(cb_ok), n_cb, ptr(sel), ptr(n_sel), stream()))
    if restart:
        check(lib().nrx_ldpc_recover_decode_merge_sel_f64(ptr(llr), n_tb, G, C.byref(cfg), nl, qm, int(n_iter), int(rows or 0),
                                                          ptr(tb_out), ptr(cb_ok), ptr(sel), ptr(n_sel), stream()))
        return tb_out, cb_ok
    marks = [int(v) for v in (stages or ()) if int(first_iter) < int(v) < int(n_iter)] + [int(n_iter)]
    done = int(first_iter)
    for k, upto in enumerate(marks):
        last = k == len(marks) - 1
        check(lib().nrx_ldpc_resume_decode_merge_sel_f64(n_tb, G, C.byref(cfg), nl, qm, upto - done, int(rows or 0), ptr(tb_out),
                                                         ptr(cb_ok), ptr(sel), ptr(n_sel), ptr(state), 0 if last else 1, stream()))
        done = upto
        if not last:        # the blocks that still fail (a subset of the selection: the others' cb_ok is 1 now)
            sel2 = torch.empty(n_cb, dtype=torch.int32, device=dev)
            n_sel2 = torch.empty(1, dtype=torch.int32, device=dev)
            check(lib().nrx_select_failed(ptr(cb_ok), n_cb, ptr(sel2), ptr(n_sel2), stream()))
            sel, n_sel = sel2, n_sel2
    return tb_out, cb_ok


def shader_clock_hz(dev, spin=200000):
    """The shader clock under a float64 load on every CU (nrx_debug_clock_probe: s_memtime against the 100 MHz s_memrealtime), Hz."""
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    sink = torch.zeros(1, dtype=torch.float64, device=dev)
    check(lib().nrx_debug_clock_probe(ptr(out), ptr(sink), int(spin), stream()))
    t, r = [int(x) for x in out.cpu()]
    return 1e8 * t / max(r, 1)


def ldpc_cert_bounds(cfg, rows):
    """nrx_ldpc_cert_bounds (host): (gamma, gamma1, dmax) of the first ``rows`` rows -- the a-priori magnitude bounds the
    early-termination certificate prices its error budget with."""
    out = (C.c_double * 3)()
    check(lib().nrx_ldpc_cert_bounds(C.byref(cfg), int(rows), out))
    return float(out[0]), float(out[1]), int(out[2])


def ldpc_decode_certified(llr, cfg, n_iter, checks, rows=None, max_sweeps=8, flags=0):
    """ldpc.py:1495-1581 decode with the certified early exit for ANY configuration (nrx_ldpc_decode_certified_f64): (n_cb, N) float64
    rate-recovered LLRs -> (hard (n_cb, K) uint8, exit_iter (n_cb,) uint8: the iteration a block was certified at, 0 = it ran all
    ``n_iter``).  ``checks``: ascending iteration counts (<= 8) after which the certificate is evaluated."""
    if llr.dtype != torch.float64 or llr.dim() != 2 or llr.shape[1] != cfg.N:
        raise ValueError(f"LLRs must be a float64 (n_cb, N={cfg.N}) tensor")
    llr = llr.contiguous()
    n_cb = llr.shape[0]
    dev = _dev(llr)
    r = int(rows or 0)
    per = int(lib().nrx_ldpc_decode_certified_ws_bytes(C.byref(cfg), r))
    ws = torch.empty(max(n_cb * per, 8), dtype=torch.uint8, device=dev)
    hard = torch.empty((n_cb, cfg.K), dtype=torch.uint8, device=dev)
    ex = torch.zeros(n_cb, dtype=torch.uint8, device=dev)
    ck = (C.c_int32 * max(len(checks), 1))(*[int(c) for c in checks])
    check(lib().nrx_ldpc_decode_certified_f64(ptr(llr), n_cb, C.byref(cfg), int(n_iter), r, ck, len(checks), ptr(hard), ptr(ex), ptr(ws),
                                              ws.numel(), int(max_sweeps), int(flags), stream()))
    return hard, ex


_cert_bufs = {}
_cert_scratch = {}


def ldpc_recover_decode_merge_certified(llr, cfg, nl, qm, stages, n_iter, rows=0, max_sweeps=4, flags=0, in_kernel=True, persistent=False):
    """The CERTIFIED early exit on the fused entry (opt-in; the reference has no early stop, ldpc.py:1545).  ``stages`` = ascending
    iteration counts at which the blocks still running are checked: a block whose CRC24B passes AND whose frozen decoder state holds
    the stability certificate (nrx_ldpc_certify_f64: every later iteration provably leaves its hard decisions unchanged) stops there;
    every other block continues from its parked state, the last ones to ``n_iter``.  Work lists stay on the device.
    -> (tb_out, cb_ok, exit_iter uint8 (n_cb,): the iteration a block was certified at, 0 = ran all n_iter), or None when the
    configuration has no fused instantiation.  ``max_sweeps`` = relaxation sweeps the certificate may take before it refuses (both forms
    honour it; 99.5 % of the certified blocks need 2; the library clamps it to 16).  ``flags`` != 0 breaks the certificate on purpose
    (tests only: bit 0 drops the sign / posterior conditions (S), (Q), bit 1 the closure (M), bit 2 also tries blocks whose CRC fails).
    ``in_kernel`` (default):
    the certificate is evaluated in the stage kernel's tail (nrx_ldpc_stage_certify_decode_merge_f64: a certified block never parks);
    False: stage, then the stand-alone nrx_ldpc_certify_f64 on the parked states -- the same conditions, another search order.
    ``persistent`` (round 6): the whole schedule as ONE launch whose code-block slots draw blocks from a device queue and take each
    through all its stages (nrx_ldpc_certified_persistent_f64): nothing parked, nothing reloaded; ``persistent_error()`` reads the
    launch's error word afterwards (tests)."""
    if llr.dtype != torch.float64 or llr.dim() != 2 or not (cfg.bg == 1 and cfg.Zc == 384 and cfg.C > 1):
        return None
    llr = llr.contiguous()
    n_tb, G = llr.shape
    dev = _dev(llr)
    n_cb = n_tb * cfg.C
    per = int(lib().nrx_ldpc_fused_state_bytes(C.byref(cfg), nl, qm, G, int(rows or 0)))
    if per == -3:
        return None
    if per < 0:
        check(per)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    state = _fused_state.get(key)
    if state is None or state.numel() < n_cb * per:
        state = _fused_state[key] = torch.empty(n_cb * per, dtype=torch.uint8, device=dev)
    bufs = _cert_bufs.get(key)
    if bufs is None or bufs[0].numel() < 2 * n_cb:
        bufs = _cert_bufs[key] = (torch.empty(2 * n_cb, dtype=torch.float64, device=dev), torch.empty(n_cb, dtype=torch.int32, device=dev),
                                  torch.empty(n_cb, dtype=torch.int32, device=dev), torch.empty(2, dtype=torch.int32, device=dev))
    lam, sel_a, sel_b, cnt = bufs
    tb_out = torch.empty((n_tb, cfg.C * (cfg.cb_len - 24)), dtype=torch.uint8, device=dev)
    cb_ok = torch.empty((n_tb, cfg.C), dtype=torch.uint8, device=dev)
    exit_iter = torch.zeros(n_cb, dtype=torch.uint8, device=dev)
    marks = sorted({int(v) for v in stages if 0 < int(v) < int(n_iter)})
    if not marks:
        raise ValueError("stages must hold at least one iteration count below n_iter")
    L = lib()
    cfgp, r = C.byref(cfg), int(rows or 0)
    if persistent:
        # ONE launch: every code-block slot of the grid takes its blocks through all their stages (nrx_ldpc_certified_persistent_f64)
        plan = [marks[0]] + [b - a for a, b in zip(marks, marks[1:] + [int(n_iter)])]
        if len(plan) > 4:
            raise ValueError("the persistent certified schedule takes at most three checks")
        n_slots = 2 * torch.cuda.get_device_properties(dev).multi_processor_count
        need = (n_slots // 2) * 2 * (26 + 2 * 15) * 384 * 4
        scr = _cert_scratch.get(key)
        if scr is None or scr.numel() < need:
            scr = _cert_scratch[key] = torch.empty(need, dtype=torch.uint8, device=dev)
        pst = _pers_state.get(key)
        if pst is None or pst[0].numel() < n_slots * per:
            pst = _pers_state[key] = (torch.empty(n_slots * per, dtype=torch.uint8, device=dev), torch.zeros(2, dtype=torch.int32, device=dev))
        arr = (C.c_int32 * len(plan))(*plan)
        check(L.nrx_ldpc_certified_persistent_f64(ptr(llr), n_tb, G, cfgp, nl, qm, arr, len(plan), r, ptr(tb_out), ptr(cb_ok), ptr(pst[0]),
                                                  ptr(lam), ptr(exit_iter), ptr(scr), scr.numel(), ptr(pst[1]), int(max_sweeps), int(flags), stream()))
        _pers_state['last_queue'] = pst[1]
        return tb_out, cb_ok, exit_iter
    if in_kernel:
        n_wg = min((n_cb + 1) // 2, int(os.environ.get('NRX_CERT_WGS', '256')))      # (developer knob: workgroups of the stage launches = scratch slots)
        need = n_wg * 2 * (26 + 2 * 15) * 384 * 4
        scr = _cert_scratch.get(key)
        if scr is None or scr.numel() < need:
            scr = _cert_scratch[key] = torch.empty(need, dtype=torch.uint8, device=dev)
        sw = int(max_sweeps)
        check(L.nrx_ldpc_stage_certify_decode_merge_f64(ptr(llr), n_tb, G, cfgp, nl, qm, marks[0], r, ptr(tb_out), ptr(cb_ok), None, None, ptr(state),
                                                        ptr(lam), ptr(exit_iter), ptr(scr), scr.numel(), marks[0], int(n_iter), sw, int(flags), stream()))
        done = marks[0]
        sels = (sel_a, sel_b)
        for k, upto in enumerate(marks[1:] + [int(n_iter)]):
            last = k == len(marks) - 1
            sel, n_sel = sels[k & 1], cnt[(k & 1):(k & 1) + 1]
            check(L.nrx_select_failed(ptr(exit_iter), n_cb, ptr(sel), ptr(n_sel), stream()))
            if last:
                check(L.nrx_ldpc_resume_decode_merge_sel_f64(n_tb, G, cfgp, nl, qm, upto - done, r, ptr(tb_out), ptr(cb_ok), ptr(sel), ptr(n_sel),
                                                             ptr(state), 0, stream()))
            else:
                check(L.nrx_ldpc_stage_certify_decode_merge_f64(None, n_tb, G, cfgp, nl, qm, upto - done, r, ptr(tb_out), ptr(cb_ok), ptr(sel),
                                                                ptr(n_sel), ptr(state), ptr(lam), ptr(exit_iter), ptr(scr), scr.numel(), upto,
                                                                int(n_iter), sw, int(flags), stream()))
            done = upto
        return tb_out, cb_ok, exit_iter
    check(L.nrx_ldpc_stage_decode_merge_f64(ptr(llr), n_tb, G, cfgp, nl, qm, marks[0], r, ptr(tb_out), ptr(cb_ok), None, None,
                                            ptr(state), ptr(lam), stream()))
    check(L.nrx_ldpc_certify_f64(ptr(state), n_tb, G, cfgp, nl, qm, r, None, None, ptr(cb_ok), ptr(lam), marks[0], int(n_iter),
                                 int(max_sweeps), int(flags), ptr(exit_iter), stream()))
    done = marks[0]
    sels = (sel_a, sel_b)
    for k, upto in enumerate(marks[1:] + [int(n_iter)]):
        last = k == len(marks) - 1
        sel, n_sel = sels[k & 1], cnt[(k & 1):(k & 1) + 1]
        check(L.nrx_select_failed(ptr(exit_iter), n_cb, ptr(sel), ptr(n_sel), stream()))       # the blocks without a certificate go on
        if last:
            check(L.nrx_ldpc_resume_decode_merge_sel_f64(n_tb, G, cfgp, nl, qm, upto - done, r, ptr(tb_out), ptr(cb_ok), ptr(sel), ptr(n_sel),
                                                         ptr(state), 0, stream()))
        else:
            check(L.nrx_ldpc_stage_decode_merge_f64(None, n_tb, G, cfgp, nl, qm, upto - done, r, ptr(tb_out), ptr(cb_ok), ptr(sel), ptr(n_sel),
                                                    ptr(state), ptr(lam), stream()))
            check(L.nrx_ldpc_certify_f64(ptr(state), n_tb, G, cfgp, nl, qm, r, ptr(sel), ptr(n_sel), ptr(cb_ok), ptr(lam), upto, int(n_iter),
                                         int(max_sweeps), int(flags), ptr(exit_iter), stream()))
        done = upto
    return tb_out, cb_ok, exit_iter


_pers_state = {}


def persistent_error():
    """The error word of the last persistent certified launch (non-zero = a slot barrier gave up: a bug).  Synchronises."""
    q = _pers_state.get('last_queue')
    return 0 if q is None else int(q[1].item())


def ldpc_fused_supported(cfg, nl, qm, G, rows):
    """Host-side mirror of nrx_ldpc_recover_decode_merge_f64's capability test (so that a caller can choose the demapper's
    output layout before it demaps)."""
    if rows is None or not (cfg.bg == 1 and cfg.Zc == 384 and cfg.iLS == 1 and cfg.C > 1 and cfg.cb_len > 24):
        return False
    f = nl * qm
    if G % f:
        return False
    lens = _lib.ldpc_cb_lens(G, cfg.C, nl, qm)
    return int(rows) <= 15 and ldpc_active_rows(cfg, max(lens)) <= 15 and max(lens) <= cfg.N - cfg.F and sum(lens) == G


def ldpc_crc_merge(dec, cfg, want_tb=True, want_tb_crc=True):
    """ldpc.py:1584-1619 checkCrcAndMerge (+ TB CRC24A check): (n_tb*C, K) -> tb_out (n_tb,M>=B), cb_ok (n_tb,C), tb_ok."""
    dec = _u8(dec)
    if dec.dim() != 2 or dec.shape[1] != cfg.K or dec.shape[0] % cfg.C:
        raise ValueError(f"decoded blocks must be (n_tb*C, K={cfg.K}), got {tuple(dec.shape)}")
    n_tb = dec.shape[0] // cfg.C
    dev = _dev(dec)
    cb_ok = torch.empty((n_tb, cfg.C), dtype=torch.uint8, device=dev)
    width = cfg.C * (cfg.cb_len - 24) if cfg.C > 1 else cfg.B
    tb_out = torch.empty((n_tb, width), dtype=torch.uint8, device=dev) if want_tb else None
    tb_ok = torch.empty((n_tb,), dtype=torch.uint8, device=dev) if (want_tb and want_tb_crc) else None
    check(lib().nrx_ldpc_crc_merge(ptr(dec), n_tb, C.byref(cfg), ptr(tb_out), ptr(cb_ok), ptr(tb_ok), stream()))
    return tb_out, cb_ok, tb_ok


def count_errors(cb_ok, tb_out, tb_ref, counters):
    """Accumulate (blockErrors, totalBlocks, bitErrors, totalBits) into the int64[4] device tensor ``counters``."""
    n_tb, A = tb_ref.shape
    check(lib().nrx_count_errors(ptr(cb_ok), cb_ok.numel(), ptr(tb_out), ptr(_u8(tb_ref)), n_tb, A,
                                 tb_out.shape[1], ptr(counters), stream()))
    return counters


# ------------------------------------------------------------------------------------------------- modem / grid
_CT = {torch.complex64: ('f32', torch.float32), torch.complex128: ('f64', torch.float64)}


def _ct(t):
    if t.dtype not in _CT:
        raise ValueError(f"complex64/complex128 tensor expected, got {t.dtype}")
    return _CT[t.dtype]


def _i32(x, dev):
    if x is None:
        return None
    if not torch.is_tensor(x):
        x = torch.as_tensor(x, dtype=torch.int32)
    return x.to(device=dev, dtype=torch.int32).contiguous()


def _host_i32(seq):
    arr = (C.c_int32 * len(seq))(*[int(v) for v in seq])
    return arr


def qam_map(bits, qm, scr=None, re_index=None, out=None, out_elems=None, dtype=torch.complex128):
    """Modem.modulate (+ PDSCH scrambling + layer/RE scatter): (n, n_sym*qm) bits -> (n, out_elems) complex.

    With ``re_index`` the symbols are scattered into ``out`` (an existing (n, out_elems) grid buffer, e.g. one
    pre-filled with DMRS); without it the output is the plain (n, n_sym) symbol array."""
    bits = _u8(bits)
    n, nb = bits.shape
    if nb % qm:
        raise ValueError("The length of 'bitstream' (%d) must be a multiple of 'qm' (%d)!" % (nb, qm))
    n_sym = nb // qm
    dev = _dev(bits)
    if out is None:
        out_elems = n_sym if out_elems is None else out_elems
        out = torch.zeros((n, out_elems), dtype=dtype, device=dev)
    flat = out.reshape(n, -1)
    assert flat.is_contiguous()
    if re_index is None and flat.shape[1] < n_sym:
        raise ValueError("output too small")
    sfx, _ = _ct(flat)
    ri = _i32(re_index, dev)
    if ri is not None:
        if ri.numel() != n_sym:
            raise ValueError(f"re_index must have {n_sym} entries, got {ri.numel()}")
    scr_t = None if scr is None else _u8(scr.to(dev))
    if scr_t is not None and scr_t.numel() < nb:
        raise ValueError("scrambling sequence shorter than the bit stream")
    fn = getattr(lib(), 'nrx_qam_map_' + sfx)
    check(fn(ptr(bits), nb, ptr(scr_t), qm, ptr(ri), n_sym, ptr(flat), flat.shape[1], n, stream()))
    return out


def layer_planes(re_inv, planes):
    """Does the inverse RE map have the layer structure nrx_pdsch_populate's ``planes`` promises (plane p of a data RE holds
    symbol number i + p where plane 0 holds i)?  One device reduction + host read: callers that reuse a map (PdschLink) ask
    once and pass the answer to :func:`pdsch_populate` as ``planes``."""
    ok = planes > 1 and re_inv.numel() % planes == 0
    if ok:
        m = re_inv.reshape(planes, -1).to(torch.int64)
        want = torch.where(m[0:1] >= 0, m[0:1] + torch.arange(planes, device=m.device)[:, None], m[0:1].clamp(max=-1))
        ok = bool(((m == want) | ((m < 0) & (want < 0))).all().item())
    return planes if ok else 0


def pdsch_populate(bits, qm, scr, re_inv, templates, templ_sel, planes=None):
    """getGrid + populateGrid in one pass (nrx_pdsch_populate_*): (n, G) bits -> (n, *templates.shape[1:]) grid; data REs
    from the bit stream through the inverse RE map ``re_inv`` (int32, one entry per grid element, -1 = not a data RE),
    everything else from ``templates[templ_sel[b]]``.  ``planes`` = :func:`layer_planes` of this map (0: no layer structure);
    None: it is worked out here, on every call (the verdict is a property of the map's CONTENTS, so it is not cached by
    address)."""
    bits = _u8(bits)
    n, nb = bits.shape
    dev = _dev(bits)
    templates = templates.contiguous()
    elems = templates[0].numel()
    sfx, _ = _ct(templates)
    if re_inv.dtype != torch.int32 or re_inv.numel() != elems:
        raise ValueError("re_inv must be an int32 tensor with one entry per grid element")
    if templ_sel.dtype != torch.int64 or templ_sel.numel() != n:
        raise ValueError("templ_sel must be an int64 tensor with one entry per batch item")
    scr_t = None if scr is None else _u8(scr.to(dev))
    if scr_t is not None and scr_t.numel() < nb:
        raise ValueError("scrambling sequence shorter than the bit stream")
    out = torch.empty((n,) + tuple(templates.shape[1:]), dtype=templates.dtype, device=dev)
    fn = getattr(lib(), 'nrx_pdsch_populate_' + sfx)
    re_inv = re_inv.contiguous()
    if planes is None:
        planes = layer_planes(re_inv, templates.shape[1]) if templates.dim() == 4 else 0
    elif planes not in (0, templates.shape[1] if templates.dim() == 4 else 0):
        raise ValueError("planes must be 0 or the number of layer planes of the templates")
    check(fn(ptr(bits), nb, ptr(scr_t), qm, ptr(re_inv), ptr(templates), ptr(templ_sel.contiguous()), elems,
             ptr(out), n, planes, stream()))
    return out


def ldpc_rows_read(cfg, rows, f32):
    """Base-graph rows whose extension LLRs the decoder entries READ for a request of ``rows`` rows: the row count of the
    instantiation that runs (BG1, Zc = 384: 13 / 15 / 16 / 22 / 31 / all in float32, 13 / 15 / 31 / all in float64, see
    nrx_ldpc_decode_rows_*), all rows for every other graph -- a superset is always safe."""
    total = 46 if cfg.bg == 1 else 42
    rows = total if not rows else int(rows)
    # the library's developer switches change which kernel runs (NRX_LDPC_ALLROWS / NRX_LDPC_NOSPEC: the float32 decoder reads every
    # row; the others fall back to kernels with their own row handling): with any of them set, every column counts as read
    if any(os.environ.get(k) is not None for k in ('NRX_LDPC_ALLROWS', 'NRX_LDPC_NOSPEC', 'NRX_LDPC_NOCHIP64', 'NRX_LDPC_NOCHIP384',
                                                    'NRX_LDPC_NOHYBRID')):
        return total
    if cfg.bg == 1 and cfg.Zc == 384:
        for r in ((13, 15, 16, 22, 31) if f32 else (13, 15, 31)):
            if rows <= r:
                return r
    return total


def qam_demap(syms, noise_var, qm, n_sym=None, scr=None, re_index=None, scales=None, exact=False, nv_floor=0.0,
              llr_dtype=None, code_blocks=None, rate_recovered=None):
    """Modem.getLLRsFromSymbols / PDSCH.getLLRsFromGrid: (n, E) complex -> (n, n_sym*qm) LLRs.
    ``code_blocks`` = (C, n_layers): write every code block's LLRs de-interleaved (nrx_qam_demap_cb_*: the input layout of
    :func:`ldpc_recover_decode_merge`) instead of symbol-major."""
    flat = syms.reshape(syms.shape[0], -1).contiguous()
    n, E = flat.shape
    sfx, rt = _ct(flat)
    dev = _dev(flat)
    ri = _i32(re_index, dev)
    if n_sym is None:
        n_sym = E if ri is None else ri.numel()
    if ri is not None and ri.numel() != n_sym:
        raise ValueError("re_index length mismatch")
    if ri is None and n_sym > E:
        raise ValueError("n_sym exceeds the number of symbols")
    nv = torch.as_tensor(noise_var, dtype=rt, device=dev).reshape(-1).contiguous()
    if nv.numel() not in (1, n):
        raise ValueError("noise_var must be a scalar or one value per batch item")
    sc = None
    if scales is not None:
        sc = scales.reshape(n, -1).to(rt).contiguous()
        if sc.shape[1] != E:
            raise ValueError("scales must have the shape of syms")
    scr_t = None if scr is None else _u8(scr.to(dev))
    if scr_t is not None and scr_t.numel() < n_sym * qm:
        raise ValueError("scrambling sequence shorter than the LLR stream")
    llr_dtype = rt if llr_dtype is None else llr_dtype
    if (rt, llr_dtype) == (torch.float64, torch.float32):
        sfx = 'f64o32'
    elif llr_dtype != rt:
        raise ValueError("unsupported LLR dtype for this input type")
    if rate_recovered is not None:
        # (cfg, n_layers, n_cols[, out]): the demapper's stores do the rate recovery of a first transmission (nrx_qam_demap_rr_*) ->
        # (n * C, N), columns [0, n_cols) of the punctured code word initialised; None when E_r wraps around the buffer
        cfg, n_layers, n_cols = rate_recovered[:3]
        if exact:
            raise ValueError("rate_recovered: max-log LLRs only")
        out = rate_recovered[3] if len(rate_recovered) > 3 else torch.empty((n * cfg.C, cfg.N), dtype=llr_dtype, device=dev)
        if out.shape != (n * cfg.C, cfg.N) or out.dtype != llr_dtype or not out.is_contiguous() or out.device != dev:
            raise ValueError("rate_recovered: the output buffer must be a contiguous (n * C, N) tensor of the LLR type")
        rc = getattr(lib(), 'nrx_qam_demap_rr_' + sfx)(ptr(flat), E, ptr(sc), ptr(nv), 0 if nv.numel() == 1 else 1, ptr(scr_t), qm, ptr(ri),
                                                       n_sym, C.byref(cfg), int(n_layers), int(n_cols), ptr(out), n, float(nv_floor), stream())
        if rc == -3:                 # NRX_E_UNSUPPORTED
            return None
        check(rc)
        return out
    llr = torch.empty((n, n_sym * qm), dtype=llr_dtype, device=dev)
    if code_blocks is not None:
        if exact:
            raise ValueError("code_blocks: max-log LLRs only")
        fn = getattr(lib(), 'nrx_qam_demap_cb_' + sfx)
        check(fn(ptr(flat), E, ptr(sc), ptr(nv), 0 if nv.numel() == 1 else 1, ptr(scr_t), qm, ptr(ri), n_sym,
                 int(code_blocks[0]), int(code_blocks[1]), ptr(llr), n_sym * qm, n, float(nv_floor), stream()))
        return llr
    fn = getattr(lib(), 'nrx_qam_demap_' + sfx)
    check(fn(ptr(flat), E, ptr(sc), ptr(nv), 0 if nv.numel() == 1 else 1, ptr(scr_t), qm, ptr(ri), n_sym, ptr(llr),
             n_sym * qm, n, 1 if exact else 0, float(nv_floor), stream()))
    return llr


def precode(grid, f):
    """Grid.precode (wideband F): (n,Nl,L,K) x (Nt,Nl) or (n,Nt,Nl) -> (n,Nt,L,K)."""
    grid = grid.contiguous()
    sfx, _ = _ct(grid)
    n, nl, L, K = grid.shape
    f = f.to(grid.dtype).contiguous()
    shared = f.dim() == 2
    nt = f.shape[-2]
    if f.shape[-1] != nl or (not shared and f.shape[0] != n):
        raise ValueError("The last dimension of 'f' (%d) must match the first dimension of the grid (%d)" % (f.shape[-1], nl))
    out = torch.empty((n, nt, L, K), dtype=grid.dtype, device=_dev(grid))
    fn = getattr(lib(), 'nrx_precode_' + sfx)
    check(fn(ptr(grid), ptr(f), 0 if shared else nt * nl, nl, nt, L * K, ptr(out), n, stream()))
    return out


def apply_channel_fd(grid, h):
    """Grid.applyChannel: grid (n,Nt,L,K), h (n,L,K,Nr,Nt) or (L,K,Nr,Nt) -> (n,Nr,L,K)."""
    grid = grid.contiguous()
    sfx, _ = _ct(grid)
    n, nt, L, K = grid.shape
    h = h.to(grid.dtype).contiguous()
    shared = h.dim() == 4
    if tuple(h.shape[-4:-2]) != (L, K) or h.shape[-1] != nt:
        raise ValueError("Mismatch in the number of transmitter antennas (%d vs %d)!" % (h.shape[-1], nt))
    nr = h.shape[-2]
    out = torch.empty((n, nr, L, K), dtype=grid.dtype, device=_dev(grid))
    fn = getattr(lib(), 'nrx_apply_channel_fd_' + sfx)
    check(fn(ptr(grid), ptr(h), 0 if shared else L * K * nr * nt, nt, nr, L * K, ptr(out), n, stream()))
    return out


def mmse_equalize(rx, hf, noise_var):
    """Grid.equalize: rx (n,Nr,L,K), hf (n,L,K,Nr,P) -> eq (n,P,L,K), llrScales (n,P,L,K)."""
    rx = rx.contiguous()
    sfx, rt = _ct(rx)
    n, nr, L, K = rx.shape
    hf = hf.to(rx.dtype).contiguous()
    shared = hf.dim() == 4
    if tuple(hf.shape[-4:-1]) != (L, K, nr):
        raise ValueError("Mismatch in the number of receiver antennas, OFDM symbols, or subcarriers!")
    P = hf.shape[-1]
    dev = _dev(rx)
    nv = torch.as_tensor(noise_var, dtype=rt, device=dev).reshape(-1).contiguous()
    eq = torch.empty((n, P, L, K), dtype=rx.dtype, device=dev)
    sc = torch.empty((n, P, L, K), dtype=rt, device=dev)
    fn = getattr(lib(), 'nrx_mmse_equalize_' + sfx)
    check(fn(ptr(rx), ptr(hf), 0 if shared else L * K * nr * P, ptr(nv), 0 if nv.numel() == 1 else 1, nr, P, L * K,
             ptr(eq), ptr(sc), n, stream()))
    return eq, sc


def td_path_spectra_bins(taps, tap_off, K, nfft):
    """nrx_td_path_spectra_bins_f64: (P, flen) taps at columns tap_off -> (P, K) complex128, the nfft-point spectrum of every row of the
    coefficient matrix at the K centred subcarriers (a constant of the channel: once per link; see mmse_equalize_paths)."""
    if not (isinstance(taps, torch.Tensor) and taps.is_cuda):
        raise ValueError("td_path_spectra_bins: taps must be a GPU tensor")
    dev = _dev(taps)
    taps = taps.to(torch.float64).contiguous()
    P, flen = taps.shape
    tap_off = _i32(tap_off, dev)
    spec = torch.empty((P, int(K)), dtype=torch.complex128, device=dev)
    check(lib().nrx_td_path_spectra_bins_f64(ptr(taps), ptr(tap_off), P, flen, int(K), int(nfft), ptr(spec), stream()))
    return spec


def mmse_equalize_paths(rx, gains, spec, chan_off, noise_var, nfft, sym_mask=None):
    """Grid.equalize on the harness's perfect CSI (channelMatrix @ precoder) with no channel matrix in memory
    (nrx_mmse_equalize_paths_f64): rx (n,Nr,L,K) complex128, gains (n,T>=L,Nr,Nl,P) = the path gains with the wideband precoder folded
    in, spec from td_path_spectra_bins, chan_off (n,) int32 -> eq (n,Nl,L,K), llrScales (n,Nl,L,K); None where there is no
    instantiation (the caller forms the matrix)."""
    if rx.dtype != torch.complex128:
        return None
    rx = rx.contiguous()
    gains = gains.to(torch.complex128).contiguous()
    n, nr, L, K = rx.shape
    if gains.shape[0] != n or gains.shape[1] < L or gains.shape[2] != nr or spec.shape != (gains.shape[4], K):
        raise ValueError("mmse_equalize_paths: gains / path spectra do not match the received grid")
    nl, P = gains.shape[3], gains.shape[4]
    dev = _dev(rx)
    chan_off = chan_off.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
    if chan_off.numel() != n:
        raise ValueError("mmse_equalize_paths: one timing offset per item")
    nv = torch.as_tensor(noise_var, dtype=torch.float64, device=dev).reshape(-1).contiguous()
    eq = torch.empty((n, nl, L, K), dtype=torch.complex128, device=dev)
    sc = torch.empty((n, nl, L, K), dtype=torch.float64, device=dev)
    mask = (1 << L) - 1 if sym_mask is None else int(sym_mask)
    rc = lib().nrx_mmse_equalize_paths_f64(ptr(rx), ptr(gains), gains.shape[1], ptr(spec), ptr(chan_off), ptr(nv), 0 if nv.numel() == 1 else 1,
                                           nr, nl, P, L, K, int(nfft), mask & 0xffffffff, ptr(eq), ptr(sc), n, stream())
    if rc == -3:                 # NRX_E_UNSUPPORTED
        return None
    check(rc)
    return eq, sc


def noise_level(x, snr_lin=None, mult=1.0, nv_mult=1.0, gather=None):
    """Complex variance per batch item (np.var) and, with ``snr_lin``, sigma = sqrt(var*mult/snr), nv = sigma^2*nv_mult."""
    flat = x.reshape(x.shape[0], -1).contiguous()
    sfx, rt = _ct(flat)
    n, m = flat.shape
    dev = _dev(flat)
    acc = torch.empty(192 * n, dtype=torch.float64, device=dev)       # <= 64 workgroup partials x 3 per item
    var = torch.empty(n, dtype=rt, device=dev)
    g = _i32(gather, dev)
    snr = sigma = nv = None
    if snr_lin is not None:
        snr = torch.as_tensor(snr_lin, dtype=torch.float64, device=dev).reshape(-1).contiguous()
        sigma = torch.empty(n, dtype=rt, device=dev)
        nv = torch.empty(n, dtype=rt, device=dev)
    fn = getattr(lib(), 'nrx_noise_level_' + sfx)
    check(fn(ptr(flat), m, m, ptr(g), 0 if g is None else g.numel(), n, ptr(acc), ptr(var), ptr(snr),
             0 if snr is None or snr.numel() == 1 else 1, float(mult), ptr(sigma), ptr(nv), float(nv_mult), stream()))
    return var, sigma, nv


def add_noise(x, z, sigma):
    """x + (sigma/sqrt2) * z with caller-supplied standard-normal pairs z (parity mode)."""
    x = x.contiguous()
    sfx, rt = _ct(x)
    n = x.shape[0]
    z = z.to(x.dtype).contiguous()
    if z.shape != x.shape:
        raise ValueError(f"Shape Mismatch: Grid: {tuple(x.shape)} vs Noise: {tuple(z.shape)}")
    sg = torch.as_tensor(sigma, dtype=rt, device=_dev(x)).reshape(-1).contiguous()
    out = torch.empty_like(x)
    fn = getattr(lib(), 'nrx_add_noise_' + sfx)
    check(fn(ptr(x), ptr(z), ptr(sg), 0 if sg.numel() == 1 else 1, x[0].numel(), ptr(out), n, stream()))
    return out


def _item_ids(item_ids, n, dev):
    """Per-item generator keys (absolute slot numbers of a non-contiguous selection): int64 device tensor of n entries."""
    if item_ids is None:
        return None
    if item_ids.dtype != torch.int64 or item_ids.numel() != n or item_ids.device != dev:
        raise ValueError("item_ids must be an int64 device tensor with one entry per batch item")
    return item_ids.contiguous()


def set_noise_precision(f64):
    """Transform of the device noise generator (throughput mode's synthetic AWGN): False = Box-Muller on the float32 transcendental
    unit (default: normals of float32 precision), True = in float64 like the reference's normals (random.py:203).  Process-wide."""
    check(lib().nrx_set_noise_precision(1 if f64 else 0))


def noise_precision():
    """'f64' or 'f32': the transform the device noise generator uses (see :func:`set_noise_precision`; NRX_RNG_F64=1 sets the default)."""
    return 'f64' if lib().nrx_get_noise_precision() else 'f32'


def awgn(x, sigma, seed, stream_id=0, batch_offset=0, item_ids=None):
    """x + complex AWGN of std sigma[b] from the counter-based device generator (throughput mode), keyed by
    (seed, stream_id, item, element) with item = item_ids[b] if given else batch_offset + b."""
    x = x.contiguous()
    sfx, rt = _ct(x)
    n = x.shape[0]
    sg = torch.as_tensor(sigma, dtype=rt, device=_dev(x)).reshape(-1).contiguous()
    out = torch.empty_like(x)
    fn = getattr(lib(), 'nrx_awgn_' + sfx)
    check(fn(ptr(x), ptr(sg), 0 if sg.numel() == 1 else 1, x[0].numel(), ptr(out), n, int(seed), int(stream_id),
             int(batch_offset), ptr(_item_ids(item_ids, n, _dev(x))), stream()))
    return out


# -------------------------------------------------------------------------------------------------------- OFDM
def ofdm_modulate(grid, nfft, cp_lens, window_len=0, pad=0, f=None):
    """Grid.ofdmModulate (+windowing): (n,P,L,K) -> (n,P,slotLen+pad) (the pad samples are zeros, Waveform.pad).

    With ``f`` (Nt,Nl) or (n,Nt,Nl) the grid holds Nl layers and the wideband precoder is applied while loading
    (Grid.precode fused; output has Nt rows per item)."""
    grid = grid.contiguous()
    sfx, _ = _ct(grid)
    n, P, L, K = grid.shape
    if len(cp_lens) != L:
        raise ValueError("one CP length per OFDM symbol is required")
    S = int(sum(cp_lens)) + L * nfft
    dev = _dev(grid)
    if f is not None:
        f = f.to(grid.dtype).contiguous()
        shared = f.dim() == 2
        nt = f.shape[-2]
        if f.shape[-1] != P or (not shared and f.shape[0] != n):
            raise ValueError("The last dimension of 'f' (%d) must match the first dimension of the grid (%d)" % (f.shape[-1], P))
    else:
        nt = P
    wave = torch.empty((n, nt, S + pad), dtype=grid.dtype, device=dev)
    if pad:
        wave[:, :, S:].zero_()
    if f is None and n * P >= 64:      # enough rows to fill the chip twice over: symbols in parallel (the same samples)
        fn = getattr(lib(), 'nrx_ofdm_modulate_sym_' + sfx)
        tails = torch.empty((n, nt, L, int(window_len)), dtype=grid.dtype, device=dev) if window_len else None
        check(fn(ptr(grid), n * P, K, nfft, _host_i32(cp_lens), L, int(window_len), ptr(wave), S + pad, ptr(tails), stream()))
    elif f is None:
        fn = getattr(lib(), 'nrx_ofdm_modulate_' + sfx)
        check(fn(ptr(grid), n * P, K, nfft, _host_i32(cp_lens), L, int(window_len), ptr(wave), S + pad, stream()))
    else:
        fn = getattr(lib(), 'nrx_ofdm_modulate_precoded_' + sfx)
        tails = torch.empty((n, nt, L, int(window_len)), dtype=grid.dtype, device=dev) if window_len else None
        check(fn(ptr(grid), n, P, nt, ptr(f), 0 if shared else nt * P, K, nfft, _host_i32(cp_lens), L, int(window_len),
                 ptr(wave), S + pad, ptr(tails), stream()))
    return wave


def ofdm_demodulate(wave, nfft, cp_lens, K, t_off=None, awgn=None, cp_offset_ratio=0.5, grid64=False):
    """Waveform.sync(t_off).ofdmDemodulate: (n,Nr,S_in) -> (n,Nr,L,K).

    ``awgn`` = (sigma, seed, stream_id, batch_offset[, item_ids]): add the noise of :func:`awgn` while loading (same values).
    ``grid64`` with a complex64 waveform: float32 transform, complex128 grid (the values of ``.to(complex128)`` afterwards)."""
    wave = wave.contiguous()
    sfx, _ = _ct(wave)
    out64 = bool(grid64) and wave.dtype == torch.complex64
    if out64:
        sfx = 'f32o64' 
    n, nr, S_in = wave.shape
    L = len(cp_lens)
    dev = _dev(wave)
    to = _i32(t_off, dev)
    if to is not None:
        to = to.reshape(-1)
        if to.numel() not in (1, n):
            raise ValueError("one timing offset per batch item expected")
    grid = torch.empty((n, nr, L, K), dtype=torch.complex128 if out64 else wave.dtype, device=dev)
    if awgn is not None:
        if cp_offset_ratio != 0.5:
            raise ValueError("the fused AWGN + demodulation entry uses cpOffsetRatio = 0.5")
        sigma, seed, stream_id, batch_offset = awgn[:4]
        ids = _item_ids(awgn[4] if len(awgn) > 4 else None, n, dev)
        _, rt = _ct(wave)
        sg = torch.as_tensor(sigma, dtype=rt, device=dev).reshape(-1).contiguous()
        fn = getattr(lib(), 'nrx_ofdm_demodulate_awgn_' + sfx)
        check(fn(ptr(wave), S_in, S_in, ptr(to), 0 if to is None or to.numel() == 1 else 1, n, nr, K, nfft,
                 _host_i32(cp_lens), L, ptr(sg), 0 if sg.numel() == 1 else 1, int(seed), int(stream_id), int(batch_offset),
                 ptr(ids), ptr(grid), stream()))
        return grid
    fn = getattr(lib(), 'nrx_ofdm_demodulate_' + sfx)
    check(fn(ptr(wave), S_in, S_in, ptr(to), 0 if to is None or to.numel() == 1 else 1, n, nr, K, nfft,
             _host_i32(cp_lens), L, float(cp_offset_ratio), ptr(grid), stream()))
    return grid


# ----------------------------------------------------------------------------------------- tapped delay line
def cdl_gains(A, nu, times, A_los=None, nu_los=0.0):
    """Time-varying CDL path gains: A (Nr,Nt,N,M) c128, nu (N,M) f64, times (n,T) f64 -> (n,T,Nr,Nt,P).
    A (n,Nr,Nt,N,M) with nu (n,N,M): ray coefficients of their own for every item (TDL 'Xiao', tdl.py:1043-1067)."""
    A = A.to(torch.complex128).contiguous()
    per_item = A.dim() == 5
    nr, nt, N, M = A.shape[-4:]
    dev = _dev(A)
    nu = nu.to(device=dev, dtype=torch.float64).contiguous()
    times = times.to(device=dev, dtype=torch.float64).contiguous()
    n, T = times.shape
    if per_item and (A.shape[0] != n or tuple(nu.shape) != (n, N, M)):
        raise ValueError("cdl_gains: per-item coefficients need A (n,Nr,Nt,N,M) and nu (n,N,M) for times (n,T)")
    if not per_item and tuple(nu.shape) != (N, M):
        raise ValueError("cdl_gains: nu must be (N,M)")
    al = None if A_los is None else A_los.to(device=dev, dtype=torch.complex128).contiguous()
    P = N + (0 if al is None else 1)
    gains = torch.empty((n, T, nr, nt, P), dtype=torch.complex128, device=dev)
    fn = lib().nrx_cdl_gains_items_f64 if per_item else lib().nrx_cdl_gains_f64
    check(fn(ptr(A), ptr(nu), ptr(al), float(nu_los), ptr(times), n, T, nr, nt, N, M, ptr(gains), stream()))
    return gains


def cir(gains, coeff, nc):
    """gains (n,T,Nr,Nt,P) x coeff (P,cl) -> cir (n,T,Nr,Nt,cl), chanOffset (n,) int32 (from the first nc instants)."""
    gains = gains.to(torch.complex128).contiguous()
    n, T, nr, nt, P = gains.shape
    dev = _dev(gains)
    coeff = coeff.to(device=dev, dtype=torch.float64).contiguous()
    if coeff.shape[0] != P:
        raise ValueError("coefficient matrix / path count mismatch")
    cl = coeff.shape[1]
    out = torch.empty((n, T, nr, nt, cl), dtype=torch.complex128, device=dev)
    off = torch.empty((n,), dtype=torch.int32, device=dev)
    check(lib().nrx_cir_f64(ptr(gains), ptr(coeff), n, T, nc, nr, nt, P, cl, ptr(out), ptr(off), stream()))
    return out, off


def chan_setup(gains, coeff, nc, K, nfft, k0, n_k):
    """chanOffset and the channel matrix at subcarriers [k0, k0 + n_k) straight from the path gains, in one launch and without a
    CIR in memory (nrx_chan_setup_f64): returns (H_sub (n, nc, n_k, Nr, Nt), off (n,) int32) -- bit-identical to
    ``cir`` + ``channel_matrix_sub`` -- or None when the configuration is outside what the fused kernel is built for."""
    gains = gains.to(torch.complex128).contiguous()
    n, T, nr, nt, P = gains.shape
    dev = _dev(gains)
    coeff = coeff.to(device=dev, dtype=torch.float64).contiguous()
    if coeff.shape[0] != P:
        raise ValueError("coefficient matrix / path count mismatch")
    H = torch.empty((n, nc, n_k, nr, nt), dtype=torch.complex128, device=dev)
    off = torch.empty((n,), dtype=torch.int32, device=dev)
    rc = lib().nrx_chan_setup_f64(ptr(gains), ptr(coeff), n, T, nc, nr, nt, P, coeff.shape[1], K, nfft, k0, n_k, ptr(off), ptr(H),
                                  stream())
    if rc == -3:           # NRX_E_UNSUPPORTED: not an error, the caller takes the two separate entries
        return None
    check(rc)
    return H, off


def chan_setup_paths(gains, coeff, spec, nc, K, nfft, k0, n_k):
    """chan_setup from the paths' spectra (nrx_chan_setup_paths_f64): gains (n,T,Nr,Nt,P), coeff (P,cl), spec (P,K) from
    td_path_spectra_bins -> (H_sub (n, nc, n_k, Nr, Nt), off (n,) int32); the values of ``chan_setup`` up to the order of the sums."""
    gains = gains.to(torch.complex128).contiguous()
    n, T, nr, nt, P = gains.shape
    dev = _dev(gains)
    coeff = coeff.to(device=dev, dtype=torch.float64).contiguous()
    if coeff.shape[0] != P or tuple(spec.shape) != (P, K) or not spec.is_contiguous():
        raise ValueError("coefficient matrix / path spectra do not match the gains")
    H = torch.empty((n, nc, n_k, nr, nt), dtype=torch.complex128, device=dev)
    off = torch.empty((n,), dtype=torch.int32, device=dev)
    check(lib().nrx_chan_setup_paths_f64(ptr(gains), ptr(coeff), ptr(spec), K, n, T, nc, nr, nt, P, coeff.shape[1], K, nfft, k0, n_k, ptr(off),
                                         ptr(H), stream()))
    return H, off


def channel_matrix(cir_t, off, nc, K, nfft):
    """ChannelModel.getChannelMatrix: cir (n,T,Nr,Nt,cl) -> H (n,nc,K,Nr,Nt)."""
    cir_t = cir_t.contiguous()
    n, T, nr, nt, cl = cir_t.shape
    dev = _dev(cir_t)
    H = torch.empty((n, nc, K, nr, nt), dtype=torch.complex128, device=dev)
    check(lib().nrx_channel_matrix_f64(ptr(cir_t), n, T, nc, nr, nt, cl, ptr(_i32(off, dev)), K, nfft, ptr(H), stream()))
    return H


def apply_td(x, cir1, set_lens):
    """ChannelModel.applyToSignal: x (n,Nt,ns), cir1 (n,nc+1,Nr,Nt,cl) -> (n,Nr,ns)."""
    x = x.to(torch.complex128).contiguous()
    cir1 = cir1.contiguous()
    n, nt, ns = x.shape
    if cir1.shape[0] != n or cir1.shape[3] != nt or cir1.shape[1] != len(set_lens):
        raise ValueError("The number of transmit antennas in the signal does not match the channel.")
    nr, cl = cir1.shape[2], cir1.shape[4]
    y = torch.empty((n, nr, ns), dtype=torch.complex128, device=_dev(x))
    check(lib().nrx_apply_td_f64(ptr(x), n, nt, ns, ptr(cir1), len(set_lens), nr, cl, _host_i32(set_lens), ptr(y),
                                 stream()))
    return y


def chest_ls(rx, pilots, port_ks, dmrs_syms, l_cdm=1, k_cdm=2, pil_set=None):
    """Grid.estimateChannelLS (linear): rx (n,Nr,L,K), pilots (sets,P,nDs,nK) -> (n,L,K,Nr,P)."""
    rx = rx.contiguous()
    sfx, _ = _ct(rx)
    n, nr, L, K = rx.shape
    dev = _dev(rx)
    pilots = pilots.to(device=dev, dtype=rx.dtype).contiguous()
    if pilots.dim() == 3:
        pilots = pilots[None]
    sets, P, nds, nk = pilots.shape
    if nds != len(dmrs_syms):
        raise ValueError("pilot / DMRS symbol count mismatch")
    pk = _i32(port_ks, dev)
    if tuple(pk.shape) != (P, nk):
        raise ValueError("port_ks must be (P, nK)")
    if int(pk.max()) >= K or int(pk.min()) < 0:
        raise ValueError("pilot subcarrier index out of range")
    ps = _i32(pil_set, dev)
    if ps is not None and (ps.numel() != n or int(ps.max()) >= sets):
        raise ValueError("pil_set must hold one valid pilot-set index per batch item")
    hest = torch.empty((n, L, K, nr, P), dtype=rx.dtype, device=dev)
    fn = getattr(lib(), 'nrx_chest_ls_' + sfx)
    check(fn(ptr(rx), ptr(pilots), ptr(ps), ptr(pk), _host_i32(dmrs_syms), nds, l_cdm, k_cdm, nk, L, K, nr, P, ptr(hest),
             n, stream()))
    return hest


def _chest_tables(rx, pilots, port_ks, dmrs_syms, pil_set):
    rx = rx.contiguous()
    if rx.dtype != torch.complex128:
        raise ValueError("the extended estimator is built for complex128")
    n, nr, L, K = rx.shape
    dev = _dev(rx)
    pilots = pilots.to(device=dev, dtype=rx.dtype).contiguous()
    if pilots.dim() == 3:
        pilots = pilots[None]
    sets, P, nds, nk = pilots.shape
    if nds != len(dmrs_syms):
        raise ValueError("pilot / DMRS symbol count mismatch")
    pk = _i32(port_ks, dev)
    if tuple(pk.shape) != (P, nk):
        raise ValueError("port_ks must be (P, nK)")
    if int(pk.max()) >= K or int(pk.min()) < 0:
        raise ValueError("pilot subcarrier index out of range")
    ps = _i32(pil_set, dev)
    if ps is not None and (ps.numel() != n or int(ps.max()) >= sets):
        raise ValueError("pil_set must hold one valid pilot-set index per batch item")
    return rx, pilots, pk, ps, (n, nr, L, K, P, nds, nk)


def chest_ls_ex(rx, pilots, port_ks, dmrs_syms, l_cdm=1, k_cdm=2, pil_set=None, polar=False, want_hk=False):
    """Grid.estimateChannelLS(polarInt=polar, kernel='linear'): -> hest (n,L,K,Nr,P) [, hk (n,nTg,K,Nr,P)]."""
    rx, pilots, pk, ps, (n, nr, L, K, P, nds, nk) = _chest_tables(rx, pilots, port_ks, dmrs_syms, pil_set)
    dev = _dev(rx)
    n_g, n_j = nds // max(l_cdm, 1), nk // max(k_cdm, 1)
    hest = torch.empty((n, L, K, nr, P), dtype=rx.dtype, device=dev)
    hk = torch.empty((n, n_g, K, nr, P), dtype=rx.dtype, device=dev) if want_hk else None
    pol = torch.empty((n * n_g * nr * P * n_j * 2,), dtype=torch.float64, device=dev) if polar else None
    check(lib().nrx_chest_ls_ex_f64(ptr(rx), ptr(pilots), ptr(ps), ptr(pk), _host_i32(dmrs_syms), nds, l_cdm, k_cdm, nk, L, K,
                                    nr, P, 1 if polar else 0, ptr(pol), ptr(hk), ptr(hest), n, stream()))
    return (hest, hk) if want_hk else hest


_noise_tabs = {}


def chest_noise_deltas(rx, pilots, port_ks, dmrs_syms, hk, nfft, cp_min, l_cdm=1, k_cdm=2, pil_set=None, ks_sample=None):
    """The pilot residuals of grid.py:808-835 against the delay-domain-windowed estimate: (n, P*nDs*nK*Nr) complex128.
    ``ks_sample``: the subcarriers every port's denoised estimate is sampled at (default: the last port's, as the
    reference does)."""
    rx, pilots, pk, ps, (n, nr, L, K, P, nds, nk) = _chest_tables(rx, pilots, port_ks, dmrs_syms, pil_set)
    dev = _dev(rx)
    rise = int(cp_min) * K // int(nfft)
    key = (K, rise, dev)
    if key not in _noise_tabs:                      # window weights exactly as grid.py:812-815, DFT twiddles e^{2 pi i q/K}
        rc = .5 * (1 - np.sin(np.pi * np.arange(rise - 1, -rise, -2) / (2 * rise)))
        win = np.concatenate([rc[::-1], rc])
        q = np.arange(K)
        tw = np.cos(2 * np.pi * q / K) + 1j * np.sin(2 * np.pi * q / K)
        _noise_tabs[key] = (torch.from_numpy(np.float64(win)).to(dev), torch.from_numpy(np.complex128(tw)).to(dev))
    win, tw = _noise_tabs[key]
    n_g = nds // l_cdm
    hk = hk.to(torch.complex128).contiguous()
    if tuple(hk.shape) != (n, n_g, K, nr, P):
        raise ValueError("hk must be (n, nTg, K, Nr, P)")
    cir = torch.empty((n, n_g * nr * P * 2 * rise), dtype=torch.complex128, device=dev)
    deltas = torch.empty((n, P * nds * nk * nr), dtype=torch.complex128, device=dev)
    kss = _i32(ks_sample, dev)
    if kss is not None and (kss.numel() != nk or int(kss.max()) >= K or int(kss.min()) < 0):
        raise ValueError("ks_sample must hold n_k subcarrier indices")
    check(lib().nrx_chest_noise_f64(ptr(rx), ptr(pilots), ptr(ps), ptr(pk), ptr(kss), _host_i32(dmrs_syms), nds, l_cdm, k_cdm,
                                    nk, L, K, nr, P, ptr(hk), ptr(tw), ptr(win), rise, ptr(cir), ptr(deltas), n, stream()))
    return deltas


def chest_noise_var(rx, pilots, port_ks, dmrs_syms, hk, nfft, cp_min, l_cdm=1, k_cdm=2, pil_set=None):
    """Raw noise variance of estimateChannelLsEx (grid.py:808-837): np.var of the pilot residuals against the
    delay-domain-windowed estimate, per batch item (float64 tensor (n,)) + the number of residuals."""
    deltas = chest_noise_deltas(rx, pilots, port_ks, dmrs_syms, hk, nfft, cp_min, l_cdm, k_cdm, pil_set)
    var, _, _ = noise_level(deltas)
    return var, deltas.shape[1]


def chest_pilot_means(rx, pilots, port_ks, dmrs_syms, l_cdm=1, k_cdm=2, pil_set=None, polar=False):
    """LS estimates at the pilots averaged over each CDM group (grid.py:775-793): (n, nTg, Nr, P, nJ) complex128 whose
    (re, im) are the estimate -- or, polar, (np.unwrap(angle) along the subcarriers, abs) as utils.py:39."""
    rx, pilots, pk, ps, (n, nr, L, K, P, nds, nk) = _chest_tables(rx, pilots, port_ks, dmrs_syms, pil_set)
    out = torch.empty((n, nds // l_cdm, nr, P, nk // k_cdm), dtype=torch.complex128, device=_dev(rx))
    check(lib().nrx_chest_pilot_means_f64(ptr(rx), ptr(pilots), ptr(ps), ptr(pk), _host_i32(dmrs_syms), nds, l_cdm, k_cdm, nk,
                                          L, K, nr, P, 1 if polar else 0, ptr(out), n, stream()))
    return out


def interp_taps(x, idx, w, n_outer, inner, n_in, n_out, in_strides, out_strides, out, polar=False):
    """out[q] = sum_t w[q][t] x[idx[q][t]] per row (both components; polar pairs recombined) -- the tap-table interpolator
    behind the non-default kinds of estimateChannelLsEx.  idx/w: (n_tabs, n_out, T); strides in complex128 elements as
    (outer, inner, sample).  The index range is checked here on the host (the tables are host-built)."""
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    w = np.ascontiguousarray(w, dtype=np.float64)
    if idx.shape != w.shape or idx.ndim != 3 or idx.shape[1] != n_out:
        raise ValueError("tap tables must be (n_tabs, n_out, T)")
    if idx.size and (idx.min() < 0 or idx.max() >= n_in):
        raise ValueError("tap index out of range")
    dev = _dev(x)
    reach_in = (n_outer - 1) * in_strides[0] + (inner - 1) * in_strides[1] + (n_in - 1) * in_strides[2]
    reach_out = (n_outer - 1) * out_strides[0] + (inner - 1) * out_strides[1] + (n_out - 1) * out_strides[2]
    if x.dtype != torch.complex128 or out.dtype != torch.complex128 or not x.is_contiguous() or not out.is_contiguous():
        raise ValueError("interp_taps works on contiguous complex128 tensors")
    if n_outer and (reach_in >= x.numel() or reach_out >= out.numel()):
        raise ValueError("interp_taps: strides reach outside the tensors")
    n_tabs, _, T = idx.shape
    di, dw = torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev)
    check(lib().nrx_interp_taps_f64(ptr(x), ptr(di), ptr(dw), T, n_tabs, n_out * T, n_outer, inner, n_out, in_strides[0],
                                    in_strides[1], in_strides[2], out_strides[0], out_strides[1], out_strides[2],
                                    1 if polar else 0, ptr(out), stream()))
    return out


def xcorr_abs(rx, ref, ref_start, ref_len):
    """Grid.estimateTimingOffset's correlation magnitude per lag (grid.py:612-622): rx (Nr, N), ref (P, M <= N) -> (N,) f64."""
    rx, ref = rx.contiguous(), ref.contiguous()
    if rx.dtype != torch.complex128 or ref.dtype != torch.complex128:
        raise ValueError("xcorr_abs works on complex128")
    dev = _dev(rx)
    out = torch.empty((rx.shape[1],), dtype=torch.float64, device=dev)
    check(lib().nrx_xcorr_abs_f64(ptr(rx), ptr(ref), rx.shape[1], ref.shape[1], int(ref_start), int(ref_len), rx.shape[0],
                                  ref.shape[0], ptr(out), stream()))
    return out


def chest_ls_mmse(rx, pilots, port_ks, dmrs_syms, noise_var, l_cdm=1, k_cdm=2, pil_set=None, sym_mask=None):
    """chest_ls + mmse_equalize fused (complex128, <= 2 DMRS time groups): -> eq (n,P,L,K), llrScales (n,P,L,K).
    ``sym_mask`` (bit l = OFDM symbol l): equalise those symbols only, the others stay uninitialised in eq / llrScales.

    Trusted inputs (the engine validates its tables once): no host-side range checks, no device synchronisation."""
    rx = rx.contiguous()
    if rx.dtype != torch.complex128:
        raise ValueError("chest_ls_mmse is built for complex128")
    n, nr, L, K = rx.shape
    dev = _dev(rx)
    if pilots.dim() == 3:
        pilots = pilots[None]
    sets, P, nds, nk = pilots.shape
    n_g = nds // l_cdm
    nv = torch.as_tensor(noise_var, dtype=torch.float64, device=dev).reshape(-1).contiguous()
    hk = torch.empty((n * n_g * (K + nk // k_cdm) * nr * P,), dtype=torch.complex128, device=dev)   # estimates + CDM-group means
    eq = torch.empty((n, P, L, K), dtype=torch.complex128, device=dev)
    sc = torch.empty((n, P, L, K), dtype=torch.float64, device=dev)
    if sym_mask is None:
        check(lib().nrx_chest_ls_mmse_f64(ptr(rx), ptr(pilots), ptr(pil_set), ptr(port_ks), _host_i32(dmrs_syms), nds, l_cdm,
                                          k_cdm, nk, L, K, nr, P, ptr(nv), 0 if nv.numel() == 1 else 1, ptr(hk), ptr(eq),
                                          ptr(sc), n, stream()))
    else:
        check(lib().nrx_chest_ls_mmse_syms_f64(ptr(rx), ptr(pilots), ptr(pil_set), ptr(port_ks), _host_i32(dmrs_syms), nds, l_cdm,
                                               k_cdm, nk, L, K, nr, P, ptr(nv), 0 if nv.numel() == 1 else 1, ptr(hk), ptr(eq),
                                               ptr(sc), n, int(sym_mask) & 0xffffffff, stream()))
    return eq, sc


def channel_matrix_sub(cir_t, off, nc, K, nfft, k0, n_k):
    """H at subcarriers [k0, k0+n_k) only: (n,nc,n_k,Nr,Nt)."""
    cir_t = cir_t.contiguous()
    n, T, nr, nt, cl = cir_t.shape
    dev = _dev(cir_t)
    H = torch.empty((n, nc, n_k, nr, nt), dtype=torch.complex128, device=dev)
    check(lib().nrx_channel_matrix_sub_f64(ptr(cir_t), n, T, nc, nr, nt, cl, ptr(_i32(off, dev)), K, nfft, k0, n_k, ptr(H),
                                           stream()))
    return H


def svd_precoder(h_block, n_layers):
    """PDSCH.getPrecodingMatrix for one group: h_block (n, ..., Nr, Nt) averaged over the middle axes -> F (n,Nt,Nl)."""
    h_block = h_block.to(torch.complex128).contiguous()
    n, nr, nt = h_block.shape[0], h_block.shape[-2], h_block.shape[-1]
    n_avg = h_block[0].numel() // (nr * nt)
    F = torch.empty((n, nt, n_layers), dtype=torch.complex128, device=_dev(h_block))
    check(lib().nrx_svd_precoder_f64(ptr(h_block), n, n_avg, nr, nt, n_layers, ptr(F), stream()))
    return F


def effective_channel(H, F):
    """H (n,L,K,Nr,Nt) @ F (n,Nt,Nl) or (Nt,Nl) -> (n,L,K,Nr,Nl)."""
    H = H.to(torch.complex128).contiguous()
    F = F.to(torch.complex128).contiguous()
    n, L, K, nr, nt = H.shape
    shared = F.dim() == 2
    nl = F.shape[-1]
    if F.shape[-2] != nt:
        raise ValueError("precoder / channel antenna count mismatch")
    out = torch.empty((n, L, K, nr, nl), dtype=torch.complex128, device=_dev(H))
    check(lib().nrx_effective_channel_f64(ptr(H), ptr(F), 0 if shared else nt * nl, n, L * K, nr, nt, nl, ptr(out), stream()))
    return out


def group_mean(H, k0, nk):
    """Mean channel per PRG: H (n,L,K,Nr,Nt), groups [k0[g], k0[g]+nk[g]) -> (n,G,Nr,Nt) (pdsch.py:1125-1127)."""
    H = H.to(torch.complex128).contiguous()
    n, L, K, nr, nt = H.shape
    dev = _dev(H)
    k0, nk = _i32(k0, dev), _i32(nk, dev)
    G = k0.numel()
    out = torch.empty((n, G, nr, nt), dtype=torch.complex128, device=dev)
    check(lib().nrx_group_mean_f64(ptr(H), n, L, K, nr * nt, ptr(k0), ptr(nk), G, ptr(out), stream()))
    return out


def precode_prg(grid, f, k2g):
    """Grid.precode with a per-PRG list: grid (n,Nl,L,K), f (n,G,Nt,Nl) or (G,Nt,Nl), k2g (K,) int32 (-1: no group)."""
    grid = grid.contiguous()
    sfx, _ = _ct(grid)
    n, nl, L, K = grid.shape
    f = f.to(grid.dtype).contiguous()
    shared = f.dim() == 3
    G, nt = f.shape[-3], f.shape[-2]
    if f.shape[-1] != nl or (not shared and f.shape[0] != n):
        raise ValueError("The last dimension of 'f' (%d) must match the first dimension of the grid (%d)" % (f.shape[-1], nl))
    k2g = _i32(k2g, _dev(grid))
    if k2g.numel() != K:
        raise ValueError("k2g must hold one group index per subcarrier")
    out = torch.empty((n, nt, L, K), dtype=grid.dtype, device=_dev(grid))
    fn = getattr(lib(), 'nrx_precode_prg_' + sfx)
    check(fn(ptr(grid), ptr(f), 0 if shared else G * nt * nl, ptr(k2g), nl, nt, L, K, ptr(out), n, stream()))
    return out


def effective_channel_prg(H, F, k2g):
    """H (n,L,K,Nr,Nt) @ F[k2g[k]] with F (n,G,Nt,Nl) -> (n,L,K,Nr,Nl); zero where k2g < 0."""
    H = H.to(torch.complex128).contiguous()
    F = F.to(torch.complex128).contiguous()
    n, L, K, nr, nt = H.shape
    G, nl = F.shape[-3], F.shape[-1]
    k2g = _i32(k2g, _dev(H))
    out = torch.empty((n, L, K, nr, nl), dtype=torch.complex128, device=_dev(H))
    check(lib().nrx_effective_channel_prg_f64(ptr(H), ptr(F), 0 if F.dim() == 3 else G * nt * nl, ptr(k2g), n, L, K, nr, nt, nl,
                                              ptr(out), stream()))
    return out


def random_bits(n_batch, n_per, seed, device, stream_id=0, batch_offset=0, item_ids=None):
    """(n_batch, n_per) uniform random bits from the counter-based device generator (keyed like :func:`awgn`)."""
    out = torch.empty((n_batch, n_per), dtype=torch.uint8, device=device)
    check(lib().nrx_random_bits(ptr(out), n_per, n_batch, int(seed), int(stream_id), int(batch_offset),
                                ptr(_item_ids(item_ids, n_batch, out.device)), stream()))
    return out


def path_taps(coeff, filter_len=16):
    """Split a (P, cl) coefficient matrix (ChannelModel.getCoeffMatrix) into its (P, flen) non-zero windows + offsets."""
    import numpy as np
    coeff = np.asarray(coeff)
    P, cl = coeff.shape
    offs = np.zeros(P, dtype=np.int32)
    taps = np.zeros((P, filter_len))
    for p in range(P):
        nz = np.nonzero(coeff[p])[0]
        lo = int(nz[0]) if len(nz) else 0
        lo = min(lo, max(cl - filter_len, 0))
        if len(nz) and nz[-1] - lo >= filter_len:
            raise ValueError("coefficient row has more than filter_len consecutive taps")
        w = coeff[p, lo:lo + filter_len]
        taps[p, :len(w)] = w
        offs[p] = lo
    return taps, offs


def apply_td_paths(x, gains1, taps, tap_off, set_lens, hist=None, power=None):
    """ChannelModel.applyToSignal, path form: x (n,Nt,ns), gains1 (n,nc+1,Nr,Nt,P), taps (P,flen), tap_off (P) -> (n,Nr,ns)."""
    if x.dtype == torch.complex64:          # float32 waveform chain (fast mode)
        got = _apply_td_paths_f32(x, gains1, taps, tap_off, set_lens, hist, power)
        if got is not NotImplemented:
            return got
        got = apply_td_paths(x.to(torch.complex128), gains1, taps, tap_off, set_lens, hist=hist, power=power)
        if got is None:
            return None
        return (got[0].to(torch.complex64),) + tuple(got[1:]) if power is not None else got.to(torch.complex64)
    x = x.to(torch.complex128).contiguous()
    gains1 = gains1.to(torch.complex128).contiguous()
    n, nt, ns = x.shape
    if gains1.shape[0] != n or gains1.shape[3] != nt or gains1.shape[1] != len(set_lens):
        raise ValueError("The number of transmit antennas in the signal does not match the channel.")
    nr, P = gains1.shape[2], gains1.shape[4]
    dev = _dev(x)
    taps = taps.to(device=dev, dtype=torch.float64).contiguous()
    host_off = None if isinstance(tap_off, torch.Tensor) and tap_off.is_cuda else np.asarray(tap_off)
    tap_off = _i32(tap_off, dev)
    if taps.shape[0] != P or tap_off.numel() != P:
        raise ValueError("tap table / path count mismatch")
    flen = taps.shape[1]
    if hist is None:                       # (device -> host read; batched callers pass it)
        hist = int(tap_off.max()) + flen - 1
    elif host_off is not None and (host_off.min() < 0 or hist < int(host_off.max()) + flen - 1):
        raise ValueError("hist must cover the longest path: max(tap_off) + flen - 1")
    y = torch.empty((n, nr, ns), dtype=torch.complex128, device=dev)
    if power is not None:
        # ... and the noise level of the output over its CP-stripped samples from the same pass (Waveform.getRePower,
        # waveform.py:107-117, then grid.py:1040-1046): power = (nfft, snr_lin, mult, nv_mult) -> (y, sigma, nv).
        nfft, snr_lin, mult, nv_mult = power
        cap = 3 * n * (-(-max(int(v) for v in set_lens) // 512) + 1) * len(set_lens) * 4
        acc = torch.empty(cap, dtype=torch.float64, device=dev)
        n_part = C.c_int32(0)
        rc = lib().nrx_apply_td_paths_pow_f64(ptr(x), n, nt, ns, ptr(gains1), len(set_lens), nr, P, ptr(taps), ptr(tap_off),
                                              flen, hist, _host_i32(set_lens), ptr(y), int(nfft), ptr(acc), cap,
                                              C.byref(n_part), stream())
        if rc == -3:                 # NRX_E_UNSUPPORTED: no register-tiled instantiation; the caller takes the two entries
            return None
        check(rc)
        snr = torch.as_tensor(snr_lin, dtype=torch.float64, device=dev).reshape(-1).contiguous()
        sigma = torch.empty(n, dtype=torch.float64, device=dev)
        nv = torch.empty(n, dtype=torch.float64, device=dev)
        check(lib().nrx_noise_level_finish_f64(ptr(acc), n_part.value, nr * (len(set_lens) - 1) * int(nfft), n, None, ptr(snr),
                                               0 if snr.numel() == 1 else 1, float(mult), ptr(sigma), ptr(nv), float(nv_mult),
                                               stream()))
        return y, sigma, nv
    check(lib().nrx_apply_td_paths_f64(ptr(x), n, nt, ns, ptr(gains1), len(set_lens), nr, P, ptr(taps), ptr(tap_off), flen,
                                       hist, _host_i32(set_lens), ptr(y), stream()))
    return y


def td_path_spectra(taps, tap_off):
    """nrx_td_path_spectra_f64: (P, flen) taps at columns tap_off -> (P, 1024) complex128 path spectra for apply_td_os (a constant
    of the channel; computed once per link)."""
    if not (isinstance(taps, torch.Tensor) and taps.is_cuda):
        raise ValueError("td_path_spectra: taps must be a GPU tensor")
    dev = _dev(taps)
    taps = taps.to(torch.float64).contiguous()
    P, flen = taps.shape
    tap_off = _i32(tap_off, dev)
    spec = torch.empty((P, 1024), dtype=torch.complex128, device=dev)
    check(lib().nrx_td_path_spectra_f64(ptr(taps), ptr(tap_off), P, flen, ptr(spec), stream()))
    return spec


def apply_td_os(x, gains1, spec, hist, set_lens, power=None):
    """ChannelModel.applyToSignal by overlap-save (nrx_apply_td_os_f64): x (n,Nt,ns) complex128, gains1 (n,nc+1,Nr,Nt,P), spec from
    td_path_spectra, hist = max(tap_off) + flen - 1 -> y (n,Nr,ns), or (y, sigma, nv) with power = (nfft, snr_lin, mult, nv_mult)
    like apply_td_paths.  None when the geometry has no instantiation (Nr != Nt, more than 4 antennas, paths longer than 640
    samples): the caller takes the path form."""
    if x.dtype != torch.complex128:
        return None
    x = x.contiguous()
    gains1 = gains1.to(torch.complex128).contiguous()
    n, nt, ns = x.shape
    if gains1.shape[0] != n or gains1.shape[3] != nt or gains1.shape[1] != len(set_lens):
        raise ValueError("The number of transmit antennas in the signal does not match the channel.")
    nr, P = gains1.shape[2], gains1.shape[4]
    if spec.shape[0] != P:
        raise ValueError("path spectra / path count mismatch")
    dev = _dev(x)
    y = torch.empty((n, nr, ns), dtype=torch.complex128, device=dev)
    acc, cap, nfft = None, 0, 0
    if power is not None:
        nfft = int(power[0])
        cap = 3 * n * len(set_lens) * 2 * nr
        acc = torch.empty(max(cap, 1), dtype=torch.float64, device=dev)
    n_part = C.c_int32(0)
    rc = lib().nrx_apply_td_os_f64(ptr(x), n, nt, ns, ptr(gains1), len(set_lens), nr, P, ptr(spec), int(hist), _host_i32(set_lens), ptr(y),
                                   nfft, ptr(acc), cap, C.byref(n_part), stream())
    if rc == -3:                 # NRX_E_UNSUPPORTED
        return None
    check(rc)
    if power is None:
        return y
    _, snr_lin, mult, nv_mult = power
    snr = torch.as_tensor(snr_lin, dtype=torch.float64, device=dev).reshape(-1).contiguous()
    sigma = torch.empty(n, dtype=torch.float64, device=dev)
    nv = torch.empty(n, dtype=torch.float64, device=dev)
    check(lib().nrx_noise_level_finish_f64(ptr(acc), n_part.value, nr * (len(set_lens) - 1) * nfft, n, None, ptr(snr),
                                           0 if snr.numel() == 1 else 1, float(mult), ptr(sigma), ptr(nv), float(nv_mult), stream()))
    return y, sigma, nv


def _apply_td_paths_f32(x, gains1, taps, tap_off, set_lens, hist, power):
    """apply_td_paths on a complex64 waveform with packed float32 arithmetic (nrx_apply_td_paths_pow_f32); sigma / nv stay
    float64.  NotImplemented when the geometry has no float32 instantiation."""
    x = x.contiguous()
    n, nt, ns = x.shape
    if gains1.shape[0] != n or gains1.shape[3] != nt or gains1.shape[1] != len(set_lens):
        raise ValueError("The number of transmit antennas in the signal does not match the channel.")
    dev = _dev(x)
    gains1 = gains1.to(torch.complex64).contiguous()
    nr, P = gains1.shape[2], gains1.shape[4]
    taps = taps.to(device=dev, dtype=torch.float32).contiguous()
    host_off = None if isinstance(tap_off, torch.Tensor) and tap_off.is_cuda else np.asarray(tap_off)
    tap_off = _i32(tap_off, dev)
    if taps.shape[0] != P or tap_off.numel() != P:
        raise ValueError("tap table / path count mismatch")
    flen = taps.shape[1]
    if hist is None:
        hist = int(tap_off.max()) + flen - 1
    elif host_off is not None and (host_off.min() < 0 or hist < int(host_off.max()) + flen - 1):
        raise ValueError("hist must cover the longest path: max(tap_off) + flen - 1")
    y = torch.empty((n, nr, ns), dtype=torch.complex64, device=dev)
    acc, cap, nfft = None, 0, 0
    if power is not None:
        nfft = int(power[0])
        cap = 3 * n * (-(-max(int(v) for v in set_lens) // 512) + 1) * len(set_lens) * 4
        acc = torch.empty(cap, dtype=torch.float64, device=dev)
    n_part = C.c_int32(0)
    rc = lib().nrx_apply_td_paths_pow_f32(ptr(x), n, nt, ns, ptr(gains1), len(set_lens), nr, P, ptr(taps), ptr(tap_off), flen, hist,
                                          _host_i32(set_lens), ptr(y), nfft, ptr(acc), cap, C.byref(n_part), stream())
    if rc == -3:                 # NRX_E_UNSUPPORTED
        return NotImplemented
    check(rc)
    if power is None:
        return y
    _, snr_lin, mult, nv_mult = power
    snr = torch.as_tensor(snr_lin, dtype=torch.float64, device=dev).reshape(-1).contiguous()
    sigma = torch.empty(n, dtype=torch.float64, device=dev)
    nv = torch.empty(n, dtype=torch.float64, device=dev)
    check(lib().nrx_noise_level_finish_f64(ptr(acc), n_part.value, nr * (len(set_lens) - 1) * nfft, n, None, ptr(snr),
                                           0 if snr.numel() == 1 else 1, float(mult), ptr(sigma), ptr(nv), float(nv_mult), stream()))
    return y, sigma, nv


def fold_precoder(gains1, f):
    """Wideband precoder folded into the path gains: gains1 (n,T,Nr,Nt,P), f (Nt,Nl) | (n,Nt,Nl) -> (n,T,Nr,Nl,P), so that
    apply_td_paths on the Nl layer waveforms equals the filter on the Nt precoded ones (see nrx.h)."""
    gains1 = gains1.to(torch.complex128).contiguous()
    n, T, nr, nt, P = gains1.shape
    f = f.to(device=_dev(gains1), dtype=torch.complex128).contiguous()
    shared = f.dim() == 2
    if f.shape[-2] != nt or (not shared and f.shape[0] != n):
        raise ValueError("fold_precoder: the precoder's shape does not match the gains")
    nl = f.shape[-1]
    out = torch.empty((n, T, nr, nl, P), dtype=torch.complex128, device=_dev(gains1))
    check(lib().nrx_fold_precoder_f64(ptr(gains1), ptr(f), 0 if shared else nt * nl, n, T, nr, nt, nl, P, ptr(out), stream()))
    return out


# ----------------------------------------------------------------------------------------------------- polar
def _idx32(t, n, what):
    if t is None:
        return None
    if t.dtype != torch.int32 or t.numel() != n:
        raise ValueError(f"{what} must be int32 with {n} entries")
    return t.contiguous()


def polar_encode(cbs, N, msg_pos, in_il=None, pc_pos=None):
    """polar.py:527-564 encode: (n_cw, K) bits -> (n_cw, N).  msg_pos/in_il/pc_pos: int32 device index tables."""
    cbs = _u8(cbs)
    n_cw, K = cbs.shape
    n_pc = 0 if pc_pos is None else pc_pos.numel()
    out = torch.empty((n_cw, N), dtype=torch.uint8, device=_dev(cbs))
    check(lib().nrx_polar_encode(ptr(cbs), n_cw, K, N, ptr(_idx32(in_il, K, 'in_il')), ptr(_idx32(msg_pos, K, 'msg_pos')),
                                 ptr(pc_pos), n_pc, ptr(out), stream()))
    return out


def polar_rate_match(coded, gather):
    """polar.py:567-603 rateMatch as one gather: (n_cw, N) -> (n_cw, E)."""
    coded = _u8(coded)
    n_cw, N = coded.shape
    E = gather.numel()
    out = torch.empty((n_cw, E), dtype=torch.uint8, device=_dev(coded))
    check(lib().nrx_polar_rate_match(ptr(coded), n_cw, N, E, ptr(_idx32(gather, E, 'gather')), ptr(out), stream()))
    return out


def polar_rate_recover(llr, N, K, inv_subblock, deinterleave=None):
    """polar.py:882-928 recoverRate: (n_cw, E) float64 LLRs -> (n_cw, N)."""
    if llr.dtype != torch.float64:
        raise ValueError("polar LLRs are float64")
    llr = llr.contiguous()
    n_cw, E = llr.shape
    out = torch.empty((n_cw, N), dtype=torch.float64, device=_dev(llr))
    check(lib().nrx_polar_rate_recover_f64(ptr(llr), n_cw, N, E, K, ptr(_idx32(deinterleave, E, 'deinterleave')),
                                           ptr(_idx32(inv_subblock, N, 'inv_subblock')), ptr(out), stream()))
    return out


def polar_scl_decode(llr, info_mask, n_info, msg_src, list_size=8, crc_poly=None, want_candidates=False, crc_expect=None):
    """polar.py:606-720 + :931-982: (n_cw, N) float64 -> msg (n_cw, K), crc_ok (n_cw,) [, cands (n_cw,L,K), costs].
    ``crc_expect``: int32/uint32 device tensor (n_cw,), the CRC register value a passing candidate ends at (masked CRCs)."""
    if llr.dtype != torch.float64:
        raise ValueError("polar LLRs are float64")
    llr = llr.contiguous()
    n_cw, N = llr.shape
    if info_mask.dtype != torch.uint8 or info_mask.numel() != N:
        raise ValueError(f"info_mask must be uint8 with N={N} entries (bit 0: non-frozen; bits 1..4: rate-0 node size)")
    K = msg_src.numel()
    dev = _dev(llr)
    msg = torch.empty((n_cw, K), dtype=torch.uint8, device=dev)
    ok = torch.empty((n_cw,), dtype=torch.uint8, device=dev)
    cands = torch.empty((n_cw, list_size, K), dtype=torch.uint8, device=dev) if want_candidates else None
    costs = torch.empty((n_cw, list_size), dtype=torch.float64, device=dev) if want_candidates else None
    crc_id = -1 if crc_poly is None else CRC_ID[crc_poly]
    if crc_expect is not None:
        if crc_expect.dtype != torch.int32 or crc_expect.numel() != n_cw or crc_expect.device != dev:
            raise ValueError("crc_expect must be an int32 device tensor with one entry per code word")
        crc_expect = crc_expect.contiguous()
    check(lib().nrx_polar_scl_decode_f64(ptr(llr), n_cw, N, int(list_size), ptr(info_mask.contiguous()), int(n_info),
                                         ptr(_idx32(msg_src, K, 'msg_src')), K, crc_id, ptr(crc_expect), ptr(msg), ptr(ok),
                                         ptr(cands), ptr(costs), stream()))
    return (msg, ok, cands, costs) if want_candidates else (msg, ok)


def csi_sinr(h, w, noise_var):
    """CsiReport.getSINR (csifeedback.py:419-433): h (n, Nr, Nt), codebook w (Ncb, Nt, Nl) complex128 -> (Ncb, n, Nl) f64."""
    h, w = h.contiguous(), w.contiguous()
    if h.dtype != torch.complex128 or w.dtype != torch.complex128:
        raise ValueError("csi_sinr works on complex128")
    if h.dim() != 3 or w.dim() != 3 or h.shape[2] != w.shape[1]:
        raise ValueError("csi_sinr: h must be (n, Nr, Nt) and w (Ncb, Nt, Nl)")
    dev = _dev(h)
    out = torch.empty((w.shape[0], h.shape[0], w.shape[2]), dtype=torch.float64, device=dev)
    check(lib().nrx_csi_sinr_f64(ptr(h), h.shape[0], h.shape[1], h.shape[2], ptr(w), w.shape[0], w.shape[2],
                                 float(noise_var), ptr(out), stream()))
    return out
