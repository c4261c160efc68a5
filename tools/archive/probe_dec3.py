"""Dev tool: phase timing of the on-chip float64 decoder from the s_memtime stamps of a -DNRX_DEC3_PROBE side build.

    tools/build_variant.sh probe -DNRX_DEC3_PROBE && NRX_LIB=exp_libs/libnrx_probe.so python3 tools/archive/probe_dec3.py [n_cb] [rows]

Prints average cycles per layer and wave for: pass 1 (LDS reads + t = r - m), min-sum, pass 2 (+ write drain + the next
layer's mask loads), barrier -- for the wide (degree 19) and narrow layers -- next to the instruction counts of the phases."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neoradium_amd import ops, _lib
dev = torch.device('cuda:0')
cfg = _lib.ldpc_config(1, 606504 + 24)
g = torch.Generator(device=dev); g.manual_seed(1)
n_cb = int(sys.argv[1]) if len(sys.argv) > 1 else 72 * 256
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 15
llr = (2 / 0.78**2 + (2 / 0.78) * torch.randn((n_cb, cfg.N), device=dev, generator=g, dtype=torch.float64))
llr[:, 13104:] = 0
lib = ctypes.CDLL(os.environ['NRX_LIB'])
out = (ctypes.c_ulonglong * 14)()
run = lambda: ops.ldpc_decode(llr, cfg, 50, rows=rows)
if len(sys.argv) > 3 and sys.argv[3] == 'fused':      # the headline entry: rate recovery in the fill, CRC + merge in the tail
    G, nl, qm = 943488, 4, 6
    raw = (2 / 0.78**2 + (2 / 0.78) * torch.randn((n_cb // cfg.C, G), device=dev, generator=g, dtype=torch.float64))
    rows = ops.ldpc_active_rows(cfg, max(_lib.ldpc_cb_lens(G, cfg.C, nl, qm)))
    run = lambda: ops.ldpc_recover_decode_merge(raw, cfg, nl, qm, 50, rows=rows)
run(); torch.cuda.synchronize()
assert lib.nrx_debug_dec3_probe(out, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
assert lib.nrx_debug_dec3_probe(out, 0) == 0
v = [int(x) for x in out]
waves, layers = v[8], v[9]
n_wide = 4
lw = layers * n_wide / rows
ln = layers * (rows - n_wide) / rows
names = ['pass1 (reads + t=r-m)', 'min-sum', 'pass2 (+drain, mask loads)', 'barrier']
res = {'ms': e0.elapsed_time(e1), 'waves': waves, 'layers_stamped': layers, 'rows': rows}
print(f"launch {res['ms']:.3f} ms (instrumented), {waves} waves, {layers} layer visits")
tot = 0
for k in range(4):
    w, n = v[k] / lw, v[4 + k] / ln
    res[names[k]] = {'wide_cycles_per_layer': w, 'narrow_cycles_per_layer': n}
    tot += v[k] + v[4 + k]
    print(f"{names[k]:28s} wide {w:8.1f}   narrow {n:8.1f}  cycles per layer per wave")
per_iter = tot / waves / (layers / waves / rows)
res['cycles_per_iteration_per_wave'] = per_iter
print(f"sum over phases: {per_iter:.0f} cycles per iteration per wave")
print(json.dumps(res))
if len(v) > 12 and v[12]:
    print(f"fill (+ barrier) {v[10] / v[12]:9.0f}   tail (+ barrier) {v[11] / v[12]:9.0f}  cycles per code-block round per wave; "
          f"{v[12] / waves:.1f} rounds per wave")
