import sys, torch
sys.path.insert(0, '.')
import bench, neoradium_amd as nr
l = bench.build_link(nr, decoder='f32', waveform='f32')
c = None
for i in range(4): c = l.run(256 * i, 256, 31.0, seed=1, counters=c)
torch.cuda.synchronize()
