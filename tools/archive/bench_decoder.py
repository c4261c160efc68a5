"""Decoder-only microbenchmark (dev tool): metric-size code blocks (BG1, Zc=384), random noisy LLRs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from neoradium_amd import ops, _lib

def main():
    n_slots = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dev = torch.device('cuda:0')
    cfg = _lib.ldpc_config(1, 606504 + 24)
    n_cb = cfg.C * n_slots
    g = torch.Generator(device=dev); g.manual_seed(1)
    # all-zero codeword over BPSK-like LLRs near the waterfall: mean 2/s^2, std 2/s
    s = 0.78
    llr = (2 / s**2 + (2 / s) * torch.randn((n_cb, cfg.N), device=dev, generator=g)).float()
    llr[:, cfg.K - 2*cfg.Zc - cfg.F: cfg.K - 2*cfg.Zc] = 1e20
    for dt in (torch.float32, torch.float64):
        x = llr.to(dt)
        out = ops.ldpc_decode(x, cfg, n_iter)
        torch.cuda.synchronize()
        reps = 3
        t0 = time.time()
        for _ in range(reps):
            out = ops.ldpc_decode(x, cfg, n_iter)
        torch.cuda.synchronize()
        dt_s = (time.time() - t0) / reps
        ber = out.float().mean().item()
        ev = n_cb * n_iter * 316 * cfg.Zc
        print(f"{dt}: {n_cb} CBs x {n_iter} it: {dt_s*1e3:.2f} ms  -> {n_slots/dt_s:.1f} slots/s (decoder only), "
              f"{ev/dt_s/1e12:.3f} T edge-visits/s, residual BER {ber:.2e}")

main()
