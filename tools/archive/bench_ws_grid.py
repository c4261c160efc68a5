"""Dev tool: the float64 workspace decoder (all 46 rows, 18 432 code blocks, 50 iterations) against the number of workgroups
in flight (NRX_LDPC_WS_GRID): each holds two 494 KB workspace slices, 256 of them = 253 MB, about the size of the MALL."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neoradium_amd import ops, _lib
dev = torch.device('cuda:0')
cfg = _lib.ldpc_config(1, 606504 + 24)
g = torch.Generator(device=dev); g.manual_seed(1)
n_cb = 72 * 256
llr = (2 / 0.78**2 + (2 / 0.78) * torch.randn((n_cb, cfg.N), device=dev, generator=g, dtype=torch.float64))
ref = None
for grid in (256, 240, 224, 208, 192, 176, 160, 128):
    os.environ['NRX_LDPC_WS_GRID'] = str(grid)
    run = lambda: ops.ldpc_decode(llr, cfg, 50)
    out = run(); torch.cuda.synchronize()
    if ref is None:
        ref = out
    best = 1e9
    for _ in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    print(json.dumps(dict(grid=grid, workspace_MB=round(grid * 2 * 46 * 384 * 28 / 1e6), ms=round(best, 2), identical=bool(torch.equal(out, ref)))), flush=True)
