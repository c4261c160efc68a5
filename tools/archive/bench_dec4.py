"""Time the run-time-Zc on-chip float64 decoder (nrx_ldpc_dec4.hip) against the workspace kernel (NRX_LDPC_NOCHIP64) and, at
Zc = 384, against the specialised kernel (nrx_ldpc_dec3.hip).  Prints one JSON line per case."""
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
for bg, zc, rows, n_cb in [(1, 384, 13, 9216), (1, 352, 13, 9216), (1, 320, 13, 9216), (1, 256, 13, 9216), (1, 208, 15, 9216),
                           (1, 128, 13, 18432), (1, 64, 13, 36864), (2, 384, 15, 9216), (2, 256, 12, 9216), (2, 64, 15, 36864)]:
    kb, core, ncols = (22, 26, 68) if bg == 1 else (10, 14, 52)
    ils = next(k for k, b in enumerate((2, 3, 5, 7, 9, 11, 13, 15)) if zc % b == 0 and (zc // b) & (zc // b - 1) == 0)
    cfg = _lib.LdpcCfg()
    cfg.bg, cfg.Zc, cfg.iLS, cfg.K, cfg.N, cfg.F, cfg.C, cfg.B, cfg.cb_len = bg, zc, ils, kb * zc, (ncols - 2) * zc, 0, 1, 0, 0
    g = torch.Generator(device=dev); g.manual_seed(zc)
    x = 2 / 0.8 ** 2 + (2 / 0.8) * torch.randn((n_cb, cfg.N), device=dev, dtype=torch.float64, generator=g)
    x[:, (core - 2 + rows - 4) * zc:] = 0.0
    r = dict(bg=bg, zc=zc, rows=rows, n_cb=n_cb, iters=50)
    os.environ.pop('NRX_LDPC_NOCHIP64', None)
    if zc == 384 and bg == 1:
        r['specialised_ms'] = round(timed(lambda: ops.ldpc_decode(x, cfg, 50, rows=rows)), 3)
        os.environ['NRX_LDPC_NOCHIP384'] = '1'
    r['chipz_ms'] = round(timed(lambda: ops.ldpc_decode(x, cfg, 50, rows=rows)), 3)
    a = ops.ldpc_decode(x, cfg, 50, rows=rows)
    os.environ.pop('NRX_LDPC_NOCHIP384', None)
    os.environ['NRX_LDPC_NOCHIP64'] = '1'
    r['workspace_ms'] = round(timed(lambda: ops.ldpc_decode(x, cfg, 50, rows=rows)), 3)
    b = ops.ldpc_decode(x, cfg, 50, rows=rows)
    del os.environ['NRX_LDPC_NOCHIP64']
    r['identical'] = bool(torch.equal(a, b))
    r['speedup'] = round(r['workspace_ms'] / r['chipz_ms'], 2)
    print(json.dumps(r), flush=True)
    out.append(r)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)
