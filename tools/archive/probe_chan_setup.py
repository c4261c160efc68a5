"""Dev tool: phase split of the fused channel set-up kernel (NRX_LIB = a -DNRX_CS_PROBE side build of nrx_chan) at the metric
configuration: 256 slots, 4x4, 24 paths, 15 gain instants.

    NRX_VARIANT_UNIT=nrx_chan tools/build_variant.sh csprobe -DNRX_CS_PROBE
    NRX_LIB=exp_libs/libnrx_csprobe.so python3 tools/archive/probe_chan_setup.py
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from neoradium_amd import ops
dev = torch.device('cuda:0')
n, T, nr, nt, P, cl = 256, 15, 4, 4, 24, 341
g = torch.Generator(device=dev); g.manual_seed(3)
gains = torch.view_as_complex(torch.randn((n, T, nr, nt, P, 2), device=dev, generator=g, dtype=torch.float64))
coeff = torch.randn((P, cl), device=dev, generator=g, dtype=torch.float64)
run = lambda: ops.chan_setup(gains, coeff, 14, 3276, 4096, 0, 12)
assert run() is not None
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b))
print(f"chan_setup {best:.3f} ms")
if os.environ.get('NRX_LIB'):
    lib = ctypes.CDLL(os.environ['NRX_LIB'])
    if hasattr(lib, 'nrx_debug_cs_probe'):
        out = (ctypes.c_ulonglong * 6)()
        lib.nrx_debug_cs_probe(out, 1)
        run(); torch.cuda.synchronize()
        lib.nrx_debug_cs_probe(out, 0)
        v = [int(x) for x in out]
        for k, nm in enumerate(['tap matrix -> LDS', 'offset pass', 'argmax', 'twiddles', 'matrix pass']):
            print(f"   {nm:20s} {v[k] / max(v[5], 1):10.0f} cycles per workgroup")
