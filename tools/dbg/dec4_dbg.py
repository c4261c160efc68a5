import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neoradium_amd import ops, _lib
dev = 'cuda:0'
for bg, zc, rows in [(1, 128, 15), (1, 64, 15), (1, 352, 15), (1, 352, 8), (1, 320, 15), (1, 208, 15), (2, 64, 15), (2, 256, 12), (1, 2, 15), (1, 36, 7)]:
    kb, core, ncols = (22, 26, 68) if bg == 1 else (10, 14, 52)
    ils = next(k for k, b in enumerate((2, 3, 5, 7, 9, 11, 13, 15)) if zc % b == 0 and (zc // b) & (zc // b - 1) == 0)
    cfg = _lib.LdpcCfg()
    cfg.bg, cfg.Zc, cfg.iLS, cfg.K, cfg.N, cfg.F, cfg.C, cfg.B, cfg.cb_len = bg, zc, ils, kb * zc, (ncols - 2) * zc, 0, 1, 0, 0
    rng = np.random.default_rng(zc)
    n_cb = 5
    llr = 2 / 0.8 ** 2 + (2 / 0.8) * rng.standard_normal((n_cb, cfg.N))
    llr[:, (core - 2 + rows - 4) * zc:] = 0.0
    x = torch.from_numpy(llr).to(dev)
    for it in (0, 1, 2, 9):
        got = ops.ldpc_decode(x, cfg, it, rows=rows)
        os.environ['NRX_LDPC_NOCHIP64'] = '1'
        ref = ops.ldpc_decode(x, cfg, it, rows=rows)
        del os.environ['NRX_LDPC_NOCHIP64']
        d = (got != ref).cpu().numpy()
        percol = d.reshape(n_cb, kb, zc).sum(axis=(0, 2))
        print(bg, zc, rows, 'it', it, 'mismatch', int(d.sum()), 'per col', percol.tolist() if d.sum() else '', flush=True)
        if d.sum() and it >= 1:
            z_bad = np.nonzero(d.reshape(n_cb, kb, zc).sum(axis=(0, 1)))[0]
            print('   bad z:', z_bad[:40].tolist(), '... n', len(z_bad))
            break
