"""ctypes binding of libnrx.so (include/nrx.h).  No fallback: if the HIP library is missing this raises.

The wrappers in this package hand torch device tensors to the C ABI through ``tensor.data_ptr()`` and the
current HIP stream; PyTorch is only used for device memory, streams and torch.distributed.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get('NRX_LIB', os.path.join(_HERE, 'libnrx.so'))    # NRX_LIB: developer override

i32, i64, u64, vp = C.c_int32, C.c_int64, C.c_size_t, C.c_void_p


class LdpcCfg(C.Structure):
    """nrx_ldpc_cfg (include/nrx.h)."""
    _fields_ = [(n, i32) for n in ('bg', 'B', 'C', 'Zc', 'iLS', 'K', 'N', 'F', 'cb_len')]

    def __repr__(self):
        return 'LdpcCfg(' + ', '.join(f'{n}={getattr(self, n)}' for n, _ in self._fields_) + ')'


class NrxError(RuntimeError):
    pass


_cfgp = C.POINTER(LdpcCfg)

# name -> (restype, argtypes); every symbol declared in include/nrx.h
SIGNATURES = {
    'nrx_version': (i32, []),
    'nrx_last_error': (i32, [C.c_char_p, i32]),
    'nrx_set_noise_precision': (i32, [i32]),
    'nrx_get_noise_precision': (i32, []),
    'nrx_gold_sequence': (i32, [C.c_uint32, i64, C.c_char_p]),
    'nrx_crc': (i32, [vp, i32, i64, i64, i32, vp, vp]),
    'nrx_ldpc_config': (i32, [i32, i32, _cfgp]),
    'nrx_ldpc_cb_lens': (i32, [i32, i32, i32, i32, C.POINTER(i32)]),
    'nrx_ldpc_segment': (i32, [vp, i32, i32, i32, _cfgp, vp, vp]),
    'nrx_ldpc_encode': (i32, [vp, i32, _cfgp, i32, i32, vp, vp]),
    'nrx_ldpc_rate_match': (i32, [vp, i32, _cfgp, i32, i32, i32, i32, i32, vp, vp]),
    'nrx_ldpc_rate_match_harq': (i32, [vp, i32, _cfgp, i32, i32, i32, vp, i32, vp, vp]),
    'nrx_ldpc_rate_recover_harq_f32': (i32, [vp, i32, i32, _cfgp, i32, i32, vp, vp, i32, vp, vp, vp]),
    'nrx_ldpc_rate_recover_harq_f64': (i32, [vp, i32, i32, _cfgp, i32, i32, vp, vp, i32, vp, vp, vp]),
    'nrx_ldpc_rate_recover_f32': (i32, [vp, i32, i32, _cfgp, i32, i32, i32, i32, vp, vp, vp]),
    'nrx_ldpc_rate_recover_f64': (i32, [vp, i32, i32, _cfgp, i32, i32, i32, i32, vp, vp, vp]),
    'nrx_ldpc_decode_ws_bytes': (u64, [_cfgp, i32]),
    'nrx_ldpc_decode_f32': (i32, [vp, i32, _cfgp, i32, i32, vp, vp, vp, u64, vp]),
    'nrx_ldpc_decode_f64': (i32, [vp, i32, _cfgp, i32, i32, vp, vp, vp, u64, vp]),
    'nrx_ldpc_crc_merge': (i32, [vp, i32, _cfgp, vp, vp, vp, vp]),
    'nrx_ldpc_recover_decode_merge_f64': (i32, [vp, i32, i32, _cfgp, i32, i32, i32, i32, vp, vp, vp]),
    'nrx_ldpc_recover_decode_merge_sel_f64': (i32, [vp, i32, i32, _cfgp, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    'nrx_select_failed': (i32, [vp, i32, vp, vp, vp]),
    'nrx_ldpc_fused_state_bytes': (i64, [_cfgp, i32, i32, i32, i32]),
    'nrx_ldpc_recover_decode_merge_park_f64': (i32, [vp, i32, i32, _cfgp, i32, i32, i32, i32, vp, vp, vp, vp]),
    'nrx_ldpc_resume_decode_merge_sel_f64': (i32, [i32, i32, _cfgp, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp]),
    'nrx_ldpc_cert_bounds': (i32, [_cfgp, i32, vp]),
    'nrx_ldpc_stage_decode_merge_f64': (i32, [vp, i32, i32, _cfgp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    'nrx_ldpc_certified_persistent_f64': (i32, [vp, i32, i32, _cfgp, i32, i32, C.POINTER(i32), i32, i32, vp, vp, vp, vp, vp, vp, u64, vp, i32, i32, vp]),
    'nrx_ldpc_stage_certify_decode_merge_f64': (i32, [vp, i32, i32, _cfgp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, u64, i32, i32, i32, i32, vp]),
    'nrx_ldpc_certify_f64': (i32, [vp, i32, i32, _cfgp, i32, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    'nrx_ldpc_decode_certified_ws_bytes': (i64, [_cfgp, i32]),
    'nrx_ldpc_decode_certified_f64': (i32, [vp, i32, _cfgp, i32, i32, vp, i32, vp, vp, vp, u64, i32, i32, vp]),
    'nrx_debug_cert_sweeps': (i32, [vp, i32]),
    'nrx_debug_clock_probe': (i32, [vp, vp, i32, vp]),
    'nrx_count_errors': (i32, [vp, i32, vp, vp, i32, i32, i32, vp, vp]),
}
f64 = C.c_double
u64_ = C.c_uint64
_i32p = C.POINTER(i32)
for _t in ('f32', 'f64'):
    SIGNATURES.update({
        'nrx_qam_map_' + _t: (i32, [vp, i64, vp, i32, vp, i32, vp, i64, i32, vp]),
        'nrx_pdsch_populate_' + _t: (i32, [vp, i64, vp, i32, vp, vp, vp, i64, vp, i32, i32, vp]),
        'nrx_precode_' + _t: (i32, [vp, vp, i64, i32, i32, i32, vp, i32, vp]),
        'nrx_apply_channel_fd_' + _t: (i32, [vp, vp, i64, i32, i32, i32, vp, i32, vp]),
        'nrx_mmse_equalize_' + _t: (i32, [vp, vp, i64, vp, i32, i32, i32, i32, vp, vp, i32, vp]),
        'nrx_noise_level_' + _t: (i32, [vp, i64, i64, vp, i64, i32, vp, vp, vp, i32, f64, vp, vp, f64, vp]),
        'nrx_add_noise_' + _t: (i32, [vp, vp, vp, i32, i64, vp, i32, vp]),
        'nrx_awgn_' + _t: (i32, [vp, vp, i32, i64, vp, i32, u64_, u64_, i64, vp, vp]),
        'nrx_ofdm_modulate_' + _t: (i32, [vp, i32, i32, i32, _i32p, i32, i32, vp, i64, vp]),
        'nrx_ofdm_modulate_sym_' + _t: (i32, [vp, i32, i32, i32, _i32p, i32, i32, vp, i64, vp, vp]),
        'nrx_ofdm_modulate_precoded_' + _t: (i32, [vp, i32, i32, i32, vp, i64, i32, i32, _i32p, i32, i32, vp, i64, vp, vp]),
        'nrx_ofdm_demodulate_' + _t: (i32, [vp, i64, i64, vp, i32, i32, i32, i32, i32, _i32p, i32, f64, vp, vp]),
        'nrx_ofdm_demodulate_awgn_' + _t: (i32, [vp, i64, i64, vp, i32, i32, i32, i32, i32, _i32p, i32, vp, i32, u64_, u64_,
                                                 i64, vp, vp, vp]),
        'nrx_chest_ls_' + _t: (i32, [vp, vp, vp, vp, _i32p, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp]),
    })
SIGNATURES['nrx_ofdm_demodulate_f32o64'] = (i32, [vp, i64, i64, vp, i32, i32, i32, i32, i32, _i32p, i32, f64, vp, vp])
SIGNATURES['nrx_ofdm_demodulate_awgn_f32o64'] = (i32, [vp, i64, i64, vp, i32, i32, i32, i32, i32, _i32p, i32, vp, i32, u64_, u64_,
                                                      i64, vp, vp, vp])
for _t in ('f32', 'f64', 'f64o32'):
    SIGNATURES['nrx_qam_demap_' + _t] = (i32, [vp, i64, vp, vp, i32, vp, i32, vp, i32, vp, i64, i32, i32, f64, vp])
    SIGNATURES['nrx_qam_demap_cb_' + _t] = (i32, [vp, i64, vp, vp, i32, vp, i32, vp, i32, i32, i32, vp, i64, i32, f64, vp])
    SIGNATURES['nrx_qam_demap_rr_' + _t] = (i32, [vp, i64, vp, vp, i32, vp, i32, vp, i32, _cfgp, i32, i32, vp, i32, f64, vp])
SIGNATURES.update({
    'nrx_cdl_gains_f64': (i32, [vp, vp, vp, f64, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    'nrx_cdl_gains_items_f64': (i32, [vp, vp, vp, f64, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    'nrx_fold_precoder_f64': (i32, [vp, vp, i64, i32, i32, i32, i32, i32, i32, vp, vp]),
    'nrx_cir_f64': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    'nrx_chan_setup_f64': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    'nrx_channel_matrix_f64': (i32, [vp, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp]),
    'nrx_apply_td_f64': (i32, [vp, i32, i32, i64, vp, i32, i32, i32, _i32p, vp, vp]),
    'nrx_apply_td_paths_f64': (i32, [vp, i32, i32, i64, vp, i32, i32, i32, vp, vp, i32, i32, _i32p, vp, vp]),
    'nrx_apply_td_paths_pow_f64': (i32, [vp, i32, i32, i64, vp, i32, i32, i32, vp, vp, i32, i32, _i32p, vp, i32, vp, i64,
                                         _i32p, vp]),
    'nrx_apply_td_paths_pow_f32': (i32, [vp, i32, i32, i64, vp, i32, i32, i32, vp, vp, i32, i32, _i32p, vp, i32, vp, i64,
                                         _i32p, vp]),
    'nrx_td_path_spectra_f64': (i32, [vp, vp, i32, i32, vp, vp]),
    'nrx_apply_td_os_f64': (i32, [vp, i32, i32, i64, vp, i32, i32, i32, vp, i32, _i32p, vp, i32, vp, i64, _i32p, vp]),
    'nrx_chan_setup_paths_f64': (i32, [vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    'nrx_td_path_spectra_bins_f64': (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
    'nrx_mmse_equalize_paths_f64': (i32, [vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, C.c_uint32, vp, vp, i32, vp]),
    'nrx_noise_level_finish_f64': (i32, [vp, i32, i64, i32, vp, vp, i32, f64, vp, vp, f64, vp]),
    'nrx_random_bits': (i32, [vp, i64, i32, u64_, u64_, i64, vp, vp]),
    'nrx_channel_matrix_sub_f64': (i32, [vp, i32, i32, i32, i32, i32, i32, vp, i32, i32, i32, i32, vp, vp]),
    'nrx_svd_precoder_f64': (i32, [vp, i32, i32, i32, i32, i32, vp, vp]),
    'nrx_effective_channel_f64': (i32, [vp, vp, i64, i32, i32, i32, i32, i32, vp, vp]),
    'nrx_chest_ls_mmse_f64': (i32, [vp, vp, vp, vp, _i32p, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, i32, vp]),
    'nrx_chest_ls_mmse_syms_f64': (i32, [vp, vp, vp, vp, _i32p, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, i32, C.c_uint32, vp]),
    'nrx_chest_ls_ex_f64': (i32, [vp, vp, vp, vp, _i32p, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp]),
    'nrx_chest_noise_f64': (i32, [vp, vp, vp, vp, vp, _i32p, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, i32, vp]),
    'nrx_chest_pilot_means_f64': (i32, [vp, vp, vp, vp, _i32p, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp]),
    'nrx_interp_taps_f64': (i32, [vp, vp, vp, i32, i32, i64, i64, i32, i32, i64, i64, i64, i64, i64, i64, i32, vp, vp]),
    'nrx_csi_sinr_f64': (i32, [vp, i32, i32, i32, vp, i32, i32, f64, vp, vp]),
    'nrx_xcorr_abs_f64': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    'nrx_group_mean_f64': (i32, [vp, i32, i32, i32, i32, vp, vp, i32, vp, vp]),
    'nrx_precode_prg_f32': (i32, [vp, vp, i64, vp, i32, i32, i32, i32, vp, i32, vp]),
    'nrx_precode_prg_f64': (i32, [vp, vp, i64, vp, i32, i32, i32, i32, vp, i32, vp]),
    'nrx_effective_channel_prg_f64': (i32, [vp, vp, i64, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    'nrx_ldpc_decode_rows_f32': (i32, [vp, i32, _cfgp, i32, i32, vp, vp, u64, vp]),
    'nrx_ldpc_decode_rows_f64': (i32, [vp, i32, _cfgp, i32, i32, vp, vp, u64, vp]),
    'nrx_ldpc_decode_rows_sel_f64': (i32, [vp, i32, _cfgp, i32, i32, vp, vp, u64, vp, vp, vp]),
    'nrx_polar_encode': (i32, [vp, i32, i32, i32, vp, vp, vp, i32, vp, vp]),
    'nrx_polar_rate_match': (i32, [vp, i32, i32, i32, vp, vp, vp]),
    'nrx_polar_rate_recover_f64': (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    'nrx_polar_scl_decode_f64': (i32, [vp, i32, i32, i32, vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp]),
})

_lib = None


def lib():
    """Load libnrx.so (once).  Fails loudly when it has not been built: there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise NrxError(f"{_LIB_PATH} not found -- build it with `python -m neoradium_amd.build` "
                           "(hipcc --offload-arch=gfx950).  neoradium_amd has no CPU fallback.")
        l = C.CDLL(_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def last_error():
    buf = C.create_string_buffer(512)
    lib().nrx_last_error(buf, 512)
    return buf.value.decode(errors='replace')


def check(rc):
    """Map an NRX_E_* return code to the exception type the reference raises for the same mistake."""
    if rc == 0:
        return
    msg = last_error()
    if rc in (-1, -2):      # NRX_E_ARG / NRX_E_SHAPE: the reference raises ValueError for bad arguments
        raise ValueError(msg)
    if rc == -3:
        raise NotImplementedError(msg)
    raise NrxError(f"libnrx error {rc}: {msg}")


_empty_anchor = {}


def ptr(t):
    """Device pointer of a (contiguous) torch tensor, or NULL for None.

    An empty tensor has no storage (data_ptr() == 0) but is not "no buffer": libnrx rejects NULL for mandatory
    arguments before it looks at the counts, so an empty tensor is passed as the address of a small per-device anchor
    (never dereferenced: a zero count returns before any launch)."""
    if t is None:
        return None
    assert t.is_contiguous(), "nrx buffers must be contiguous"
    if t.numel() == 0:
        import torch
        a = _empty_anchor.get(t.device)
        if a is None:
            a = _empty_anchor[t.device] = torch.zeros(16, dtype=torch.uint8, device=t.device)
        return a.data_ptr()
    return t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def ldpc_config(bg, B):
    cfg = LdpcCfg()
    check(lib().nrx_ldpc_config(int(bg), int(B), C.byref(cfg)))
    return cfg


def ldpc_cb_lens(G, Cn, nl, qm):
    arr = (i32 * Cn)()
    check(lib().nrx_ldpc_cb_lens(int(G), int(Cn), int(nl), int(qm), arr))
    return list(arr)


def gold_sequence(c_init, n):
    """utils.py:70-94 goldSequence -> numpy int8 array of n bits (host)."""
    import numpy as np
    buf = C.create_string_buffer(max(int(n), 1))
    check(lib().nrx_gold_sequence(int(c_init) & 0xFFFFFFFF, int(n), buf))
    return np.frombuffer(buf.raw, dtype=np.uint8, count=int(n)).astype(np.int8)
