"""Dev tool: decoder kernel time vs number of code blocks (occupancy / tail study)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neoradium_amd import ops, _lib
dev = torch.device('cuda:0')
cfg = _lib.ldpc_config(1, 606504 + 24)
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = torch.Generator(device=dev); g.manual_seed(1)
for n_cb in [int(x) for x in (sys.argv[2:] or [256, 512, 1024, 2048, 4096, 9216])]:
    llr = (2 / 0.78**2 + (2 / 0.78) * torch.randn((n_cb, cfg.N), device=dev, generator=g)).float()
    ops.ldpc_decode(llr, cfg, n_iter); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ops.ldpc_decode(llr, cfg, n_iter)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"n_cb {n_cb:6d}: {ms:8.3f} ms   {n_cb/72/ms*1e3:8.1f} slots/s   {ms*1e3/ (n_cb*n_iter) * 256:7.2f} us per CB-iteration-per-CU", flush=True)
