"""Dev tool: BLER vs SNR of the bench configuration on the GPU engine (throughput mode)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import neoradium_amd as nr
from bench import build_link
link = build_link(nr)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for snr in [float(x) for x in (sys.argv[2:] or [18, 20, 22, 24, 26, 28, 30, 32, 34])]:
    c = link.run(0, n, snr, seed=1).cpu().numpy()
    print(f"SNR {snr:5.1f} dB  BLER {c[0]}/{c[1]} = {100*c[0]/c[1]:6.2f}%   BER {100*c[2]/c[3]:.3f}%", flush=True)
