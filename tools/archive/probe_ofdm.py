"""Dev tool: time the precoded OFDM modulator and the AWGN + demodulator at the metric configuration (256 slots, 4 layers ->
4 ports, 273 PRB, nFFT 4096) and, with a -DNRX_OFDM_PROBE side build (NRX_LIB), print the s_memtime phase split.

    NRX_VARIANT_UNIT=nrx_ofdm tools/build_variant.sh ofdmprobe -DNRX_OFDM_PROBE
    NRX_LIB=exp_libs/libnrx_ofdmprobe.so python3 tools/archive/probe_ofdm.py
"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neoradium_amd import ops
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K, nfft, L = 3276, 4096, 14
cp = [352] + [288] * 13
g = torch.Generator(device=dev); g.manual_seed(1)
grid = torch.randn((n, 4, L, K, 2), device=dev, generator=g, dtype=torch.float64)
grid = torch.view_as_complex(grid)
f = torch.view_as_complex(torch.randn((n, 4, 4, 2), device=dev, generator=g, dtype=torch.float64))
sig = torch.full((n,), 0.01, dtype=torch.float64, device=dev)
toff = torch.full((n,), 7, dtype=torch.int32, device=dev)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


wave = ops.ofdm_modulate(grid, nfft, cp, window_len=0, pad=400, f=f)
res = {}
res['mod_nowin_ms'] = timed(lambda: ops.ofdm_modulate(grid, nfft, cp, window_len=0, pad=400, f=f))
res['mod_win_ms'] = timed(lambda: ops.ofdm_modulate(grid, nfft, cp, window_len=144, pad=400, f=f))
res['mod_layers_win_ms'] = timed(lambda: ops.ofdm_modulate(grid, nfft, cp, window_len=144, pad=400))      # what the engine runs
res['demod_awgn_ms'] = timed(lambda: ops.ofdm_demodulate(wave, nfft, cp, K, t_off=toff, awgn=(sig, 1, 2, 0)))
res['demod_plain_ms'] = timed(lambda: ops.ofdm_demodulate(wave, nfft, cp, K, t_off=toff))
print(json.dumps(res))
if os.environ.get('NRX_LIB'):
    lib = ctypes.CDLL(os.environ['NRX_LIB'])
    if hasattr(lib, 'nrx_debug_ofdm_probe'):
        out = (ctypes.c_ulonglong * 16)()
        lib.nrx_debug_ofdm_probe(out, 1)
        ops.ofdm_modulate(grid, nfft, cp, window_len=144, pad=400, f=f)
        ops.ofdm_demodulate(wave, nfft, cp, K, t_off=toff, awgn=(sig, 1, 2, 0))
        torch.cuda.synchronize()
        lib.nrx_debug_ofdm_probe(out, 0)
        v = [int(x) for x in out]
        for base, name, names in ((0, 'modulator', ['fill (loads + precode -> LDS)', 'barrier', 'FFT', 'write-out']),
                                  (8, 'demodulator', ['fill (loads + AWGN -> LDS)', 'barrier', 'FFT', 'write-out'])):
            waves = v[base + 5]
            print(name, 'wave-tasks', waves)
            for k, nm in enumerate(names):
                print(f"   {nm:34s} {v[base + k] / max(waves, 1):10.0f} cycles per wave-task")
