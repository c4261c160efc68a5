"""Dev probe: how much of the front end hides behind the decoder when both run on separate HIP streams?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import neoradium_amd as nr
from neoradium_amd import ops
import bench
link = bench.build_link(nr)
B, n = 256, 4
dev = link.dev
cfg = link.cfg
rows = link.cw[0]['rows']
# a real rate-recovered LLR batch for the decoder loop
cap = {}
orig = ops.ldpc_decode
def grab(rr, c, it, rows=None):
    cap['rr'] = rr
    return orig(rr, c, it, rows=rows)
ops.ldpc_decode = grab
link.run(0, B, 31.0, seed=1); torch.cuda.synchronize()
rr = cap['rr'].clone()
zeros = torch.zeros((rr.shape[0], cfg.K), dtype=torch.uint8, device=dev)
ops.ldpc_decode = lambda rr_, c, it, rows=None: zeros            # front end only
def front(k): link.run(k * B, B, 31.0, seed=1)
def dec(): orig(rr, cfg, 50, rows=rows)
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
front(1); dec()
t_front = timed(lambda: [front(k) for k in range(n)]) / n
t_dec = timed(lambda: [dec() for _ in range(n)]) / n
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    for k in range(n):
        with torch.cuda.stream(sb):
            dec()
        with torch.cuda.stream(sa):
            front(k)
t_both = timed(both) / n
print(f"front end {t_front:.2f} ms, decoder {t_dec:.2f} ms, sum {t_front + t_dec:.2f} ms, two streams {t_both:.2f} ms per {B} slots "
      f"-> {B / t_both * 1e3:.0f} slots/s if fully pipelined")
