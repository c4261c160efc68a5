"""Dev tool: time of the time-domain channel filter (nrx_apply_td_paths_f64) at the metric configuration, 256 slots."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import neoradium_amd as nr
from neoradium_amd import ops
link = bench.build_link(nr, decoder='f64')
dev = link.dev
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device=dev); g.manual_seed(1)
ns = link.slot_len[0] + link.max_delay
x = torch.complex(torch.randn((B, 4, ns), dtype=torch.float64, device=dev, generator=g), torch.randn((B, 4, ns), dtype=torch.float64, device=dev, generator=g))
times = torch.from_numpy(link.gain_times(np.arange(B))).to(dev)
gains = ops.cdl_gains(link.A, link.nu, times, A_los=link.Alos, nu_los=link.nulos)
sl = [int(v) for v in link.sym_lens[0]]
f = lambda: ops.apply_td_paths(x, gains, link.taps, link.tap_off, sl, hist=link.td_hist)
y = f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): y = f()
e1.record(); torch.cuda.synchronize()
print(f"apply_td_paths B={B} lib={os.environ.get('NRX_LIB','tree')}: {e0.elapsed_time(e1)/5:.3f} ms  checksum {float(y.abs().sum()):.6e}", flush=True)
if os.environ.get('NRX_LIB') and 'probe' in os.environ['NRX_LIB']:
    import ctypes
    lib = ctypes.CDLL(os.environ['NRX_LIB'])
    out = (ctypes.c_ulonglong * 3)()
    lib.nrx_debug_td_probe(out)
    t, r, n = [int(v) for v in out]
    print(f"clock probe: {n} workgroups, mean lifetime {t / n:.0f} s_memtime cycles = {r / n / 100:.2f} us  =>  {t / (r / 100.0):.0f} MHz during the kernel")
