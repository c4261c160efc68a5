"""Batched device-tensor front end of the C ABI (include/nrx.h).

Every function takes/returns torch tensors resident on the GPU, validates shapes on the host (a faulting kernel
can take the whole node down), allocates outputs with torch, and enqueues one libnrx call on the current HIP
stream.  Leading batch dimension = Monte-Carlo slots / transport blocks.
"""
import ctypes as C
import torch

from . import _lib
from ._lib import lib, check, ptr, stream

CRC_ID = {'6': 0, '11': 1, '16': 2, '24A': 3, '24B': 4, '24C': 5}
CRC_LEN = {'6': 6, '11': 11, '16': 16, '24A': 24, '24B': 24, '24C': 24}
_FT = {torch.float32: 'f32', torch.float64: 'f64'}


def _dev(t):
    if not t.is_cuda:
        raise ValueError("nrx ops need tensors on the GPU (no CPU fallback)")
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


def ldpc_encode(cbs, cfg, puncture=True):
    """ldpc.py:1033-1090 encode: (n_cb, K) -> (n_cb, N) (or N+2Zc without puncturing)."""
    cbs = _u8(cbs)
    if cbs.dim() != 2 or cbs.shape[1] != cfg.K:
        raise ValueError(f"code blocks must be (n_cb, K={cfg.K}), got {tuple(cbs.shape)}")
    width = cfg.N if puncture else cfg.N + 2 * cfg.Zc
    out = torch.empty((cbs.shape[0], width), dtype=torch.uint8, device=_dev(cbs))
    check(lib().nrx_ldpc_encode(ptr(cbs), cbs.shape[0], C.byref(cfg), 1 if puncture else 0, ptr(out), stream()))
    return out


def ldpc_rate_match(coded, cfg, G, nl, qm, rv=0, nref=0):
    """ldpc.py:1093-1159 rateMatch: (n_tb*C, N) -> (n_tb, sum E_r)."""
    coded = _u8(coded)
    if coded.dim() != 2 or coded.shape[1] != cfg.N or coded.shape[0] % cfg.C:
        raise ValueError(f"coded blocks must be (n_tb*C, N={cfg.N}), got {tuple(coded.shape)}")
    n_tb = coded.shape[0] // cfg.C
    f = nl * qm
    gout = ((G + f - 1) // f) * f
    out = torch.empty((n_tb, gout), dtype=torch.uint8, device=_dev(coded))
    check(lib().nrx_ldpc_rate_match(ptr(coded), n_tb, C.byref(cfg), int(G), nl, qm, rv, nref, ptr(out), stream()))
    return out


def ldpc_rate_recover(llr, cfg, nl, qm, rv=0, nref=0, circ=None):
    """ldpc.py:1330-1418 recoverRate: (n_tb, G) LLRs -> (n_tb*C, N); ``circ`` (n_tb*C, Ncb-F) accumulates in place."""
    if llr.dtype not in _FT:
        raise ValueError("LLRs must be float32 or float64")
    llr = llr.contiguous()
    n_tb, G = llr.shape
    ncb = cfg.N if nref == 0 else min(cfg.N, nref)
    if circ is not None:
        if tuple(circ.shape) != (n_tb * cfg.C, ncb - cfg.F) or circ.dtype != llr.dtype:
            raise ValueError(f"HARQ buffer shape mismatch! It must be a {n_tb * cfg.C}x{ncb - cfg.F} {llr.dtype} tensor!")
    out = torch.empty((n_tb * cfg.C, cfg.N), dtype=llr.dtype, device=_dev(llr))
    fn = getattr(lib(), 'nrx_ldpc_rate_recover_' + _FT[llr.dtype])
    check(fn(ptr(llr), n_tb, G, C.byref(cfg), nl, qm, rv, nref, ptr(circ), ptr(out), stream()))
    return out


_ws_cache = {}


def _decode_ws(cfg, device):
    need = lib().nrx_ldpc_decode_ws_bytes(C.byref(cfg), 1)
    key = (device, cfg.bg)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def ldpc_decode(llr, cfg, n_iter=5, only_info=True, belief=False):
    """ldpc.py:1495-1581 decode: (n_cb, N) LLRs -> (n_cb, K or N+2Zc) hard bits (uint8) or beliefs.

    float64 input runs the bit-exact float64 kernel, float32 input the single-precision throughput kernel."""
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
    if llr.dtype == torch.float64:
        ws = _decode_ws(cfg, dev)
        check(lib().nrx_ldpc_decode_f64(ptr(llr), n_cb, C.byref(cfg), int(n_iter), cols, ptr(hard), ptr(bel), ptr(ws),
                                        ws.numel(), stream()))
    else:
        check(lib().nrx_ldpc_decode_f32(ptr(llr), n_cb, C.byref(cfg), int(n_iter), cols, ptr(hard), ptr(bel), None, 0,
                                        stream()))
    return bel if belief else hard


def ldpc_crc_merge(dec, cfg, want_tb=True):
    """ldpc.py:1584-1619 checkCrcAndMerge (+ TB CRC24A check): (n_tb*C, K) -> tb_out (n_tb,B), cb_ok (n_tb,C), tb_ok."""
    dec = _u8(dec)
    if dec.dim() != 2 or dec.shape[1] != cfg.K or dec.shape[0] % cfg.C:
        raise ValueError(f"decoded blocks must be (n_tb*C, K={cfg.K}), got {tuple(dec.shape)}")
    n_tb = dec.shape[0] // cfg.C
    dev = _dev(dec)
    cb_ok = torch.empty((n_tb, cfg.C), dtype=torch.uint8, device=dev)
    tb_out = torch.empty((n_tb, cfg.B), dtype=torch.uint8, device=dev) if want_tb else None
    tb_ok = torch.empty((n_tb,), dtype=torch.uint8, device=dev) if want_tb else None
    check(lib().nrx_ldpc_crc_merge(ptr(dec), n_tb, C.byref(cfg), ptr(tb_out), ptr(cb_ok), ptr(tb_ok), stream()))
    return tb_out, cb_ok, tb_ok


def count_errors(cb_ok, tb_out, tb_ref, counters):
    """Accumulate (blockErrors, totalBlocks, bitErrors, totalBits) into the int64[4] device tensor ``counters``."""
    n_tb, A = tb_ref.shape
    check(lib().nrx_count_errors(ptr(cb_ok), cb_ok.numel(), ptr(tb_out), ptr(_u8(tb_ref)), n_tb, A,
                                 tb_out.shape[1], ptr(counters), stream()))
    return counters
