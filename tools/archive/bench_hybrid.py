"""Dev tool: float64 decoding with more than 15 rows at Zc = 384 (18 432 code blocks, 50 iterations): the hybrid of
nrx_ldpc_dec3.hip against the workspace kernel (NRX_LDPC_NOHYBRID), bits compared."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neoradium_amd import ops, _lib
dev = torch.device('cuda:0')
cfg = _lib.ldpc_config(1, 606504 + 24)
g = torch.Generator(device=dev); g.manual_seed(1)
n_cb = 72 * 256
llr = (2 / 0.78**2 + (2 / 0.78) * torch.randn((n_cb, cfg.N), device=dev, generator=g, dtype=torch.float64))


def timed(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


for rows in (46, 30, 20):
    x = llr.clone()
    x[:, (22 + rows - 4) * 384:] = 0.0
    os.environ.pop('NRX_LDPC_NOHYBRID', None)
    hy = ops.ldpc_decode(x, cfg, 50, rows=rows)
    t_hy = timed(lambda: ops.ldpc_decode(x, cfg, 50, rows=rows))
    os.environ['NRX_LDPC_NOHYBRID'] = '1'
    wk = ops.ldpc_decode(x, cfg, 50, rows=rows)
    t_wk = timed(lambda: ops.ldpc_decode(x, cfg, 50, rows=rows))
    del os.environ['NRX_LDPC_NOHYBRID']
    print(json.dumps(dict(rows=rows, hybrid_ms=round(t_hy, 2), workspace_ms=round(t_wk, 2), identical=bool(torch.equal(hy, wk)))), flush=True)
