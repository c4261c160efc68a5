import sys, time, json, torch
sys.path.insert(0, '.')
import bench, neoradium_amd as nr
res = {}
for name, kw in (('f32dec', dict(decoder='f32')), ('f32dec_f32wave', dict(decoder='f32', waveform='f32'))):
    l = bench.build_link(nr, **kw)
    for _ in range(2): l.run(0, 256, 31.0, seed=1)
    torch.cuda.synchronize(); t = time.time(); c = None
    for i in range(10): c = l.run(256 * (i + 1), 256, 31.0, seed=1, counters=c)
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
    res[name] = dict(ms=dt * 1e3, slots_s=256 / dt, counters=c.cpu().tolist())
    print(name, res[name], flush=True)
