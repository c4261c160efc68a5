"""Dev tool: float64 decoder, on-chip kernel (nrx_ldpc_dec3.hip) against the workspace kernel (NRX_LDPC_NOCHIP64=1 in a
second process): time per 256-slot launch at the metric configuration and a checksum of the hard bits."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neoradium_amd import ops, _lib
dev = torch.device('cuda:0')
cfg = _lib.ldpc_config(1, 606504 + 24)
g = torch.Generator(device=dev); g.manual_seed(1)
n_cb = int(sys.argv[1]) if len(sys.argv) > 1 else 72 * 256
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 15
sig = float(sys.argv[3]) if len(sys.argv) > 3 else 0.78
llr = (2 / sig**2 + (2 / sig) * torch.randn((n_cb, cfg.N), device=dev, generator=g, dtype=torch.float64))
llr[:, 13104:] = 0
llr[:, 8448 - 768 - 24:8448 - 768] = 1e20           # filler positions (F = 24): the +1e5 quirk path runs
NIT = int(os.environ.get('NRX_BENCH_ITERS', '50'))
fused = os.environ.get('NRX_BENCH_FUSED') is not None
if fused:       # the fused entry on the same LLRs: (n_tb, G) in per-code-block de-interleaved order = the first E columns of llr
    E = 13104
    g_in = llr[:, :8448 - 768 - 24].clone() if False else None
    # rate-recovered row = [sys (K-2Zc-F) | fillers F | parity]; the circular buffer (no fillers) holds its first E entries
    sys_len = cfg.K - 2 * cfg.Zc - cfg.F
    buf = torch.cat([llr[:, :sys_len], llr[:, sys_len + cfg.F:]], dim=1)[:, :E].contiguous()
    x = buf.reshape(n_cb // cfg.C, cfg.C * E).contiguous()
    run = lambda: ops.ldpc_recover_decode_merge(x, cfg, 4, 6, NIT, rows=rows)
else:
    run = lambda: ops.ldpc_decode(llr, cfg, NIT, rows=rows)
out = run(); torch.cuda.synchronize()
if fused:
    out = out[0]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): out = run()
if fused:
    out = out[0]
e1.record(); torch.cuda.synchronize()
import hashlib
h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
print(f"iters={NIT} fused={fused} NOCHIP64={os.environ.get('NRX_LDPC_NOCHIP64')} n_cb {n_cb} rows {rows}: {e0.elapsed_time(e1)/3:.3f} ms  ones {int(out.sum())} sha {h}", flush=True)
