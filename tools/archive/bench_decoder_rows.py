import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from neoradium_amd import ops, _lib
dev = torch.device('cuda:0')
cfg = _lib.ldpc_config(1, 606504 + 24)
g = torch.Generator(device=dev); g.manual_seed(1)
n_cb = 4608
llr = (2 / 0.78**2 + (2 / 0.78) * torch.randn((n_cb, cfg.N), device=dev, generator=g)).float()
llr[:, 13104:] = 0
for rows in [int(x) for x in sys.argv[1:]]:
    ops.ldpc_decode(llr, cfg, 50, rows=rows); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): out = ops.ldpc_decode(llr, cfg, 50, rows=rows)
    e1.record(); torch.cuda.synchronize()
    print(f"rows {rows}: {e0.elapsed_time(e1)/3:.3f} ms  checksum {int(out.sum())}", flush=True)
