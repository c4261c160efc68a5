"""Time the run-time-Zc hybrid float64 decoder (nrx_ldpc_dec4.hip, more than 15 rows at lifting sizes other than 384) against the
workspace kernel (NRX_LDPC_NOHYBRID) and, at Zc = 384, against the specialised hybrid (nrx_ldpc_dec3.hip).  One JSON line per case."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neoradium_amd import ops, _lib

dev = 'cuda:0'


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts)


out = []
IT = 20
for bg, zc, rows, n_cb in [(1, 384, 46, 4608), (1, 352, 46, 4608), (1, 352, 31, 4608), (1, 256, 46, 4608), (1, 128, 46, 9216), (1, 64, 31, 18432),
                           (2, 384, 42, 4608), (2, 256, 21, 9216), (2, 256, 42, 9216), (2, 128, 22, 18432), (2, 64, 42, 18432)]:
    kb, core, ncols = (22, 26, 68) if bg == 1 else (10, 14, 52)
    ils = next(k for k, b in enumerate((2, 3, 5, 7, 9, 11, 13, 15)) if zc % b == 0 and (zc // b) & (zc // b - 1) == 0)
    cfg = _lib.LdpcCfg()
    cfg.bg, cfg.Zc, cfg.iLS, cfg.K, cfg.N, cfg.F, cfg.C, cfg.B, cfg.cb_len = bg, zc, ils, kb * zc, (ncols - 2) * zc, 0, 1, 0, 0
    g = torch.Generator(device=dev); g.manual_seed(zc + rows)
    x = 2 / 0.9 ** 2 + (2 / 0.9) * torch.randn((n_cb, cfg.N), device=dev, dtype=torch.float64, generator=g)
    x[:, (core - 2 + rows - 4) * zc:] = 0.0
    r = dict(bg=bg, zc=zc, rows=rows, n_cb=n_cb, iters=IT)
    for k in ('NRX_LDPC_NOHYBRID', 'NRX_LDPC_NOCHIP384'):
        os.environ.pop(k, None)
    if zc == 384 and bg == 1:
        r['specialised_hybrid_ms'] = round(timed(lambda: ops.ldpc_decode(x, cfg, IT, rows=rows)), 3)
        os.environ['NRX_LDPC_NOCHIP384'] = '1'
    r['hybrid_ms'] = round(timed(lambda: ops.ldpc_decode(x, cfg, IT, rows=rows)), 3)
    a = ops.ldpc_decode(x, cfg, IT, rows=rows)
    os.environ['NRX_LDPC_NOHYBRID'] = '1'
    r['workspace_ms'] = round(timed(lambda: ops.ldpc_decode(x, cfg, IT, rows=rows)), 3)
    b = ops.ldpc_decode(x, cfg, IT, rows=rows)
    for k in ('NRX_LDPC_NOHYBRID', 'NRX_LDPC_NOCHIP384'):
        os.environ.pop(k, None)
    r['identical'] = bool(torch.equal(a, b))
    r['speedup'] = round(r['workspace_ms'] / r['hybrid_ms'], 2)
    print(json.dumps(r), flush=True)
    out.append(r)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)
