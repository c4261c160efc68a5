import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import neoradium_amd as nr
nr.random.setSeed(123)
car = nr.Carrier(numRbs=25, spacing=15)
bwp = car.curBwp
p = nr.PDSCH(bwp, numLayers=1, nID=car.cellId, modulation='QPSK')
p.setDMRS(configType=1, additionalPos=1)
ch = nr.TdlChannel(bwp, 'A', delaySpread=30, carrierFreq=4e9, dopplerShift=5)
link = nr.PdschLink(p, ch, 0.3, baseGraphNo=2, numIter=20, freqDomain=False, chanEst="LS")
for B in (64, 512, 4096):
    link.run(0, B, 2.0, seed=1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(3): link.run((k + 1) * B, B, 2.0, seed=1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"cfg1 B={B}: {B/dt:.0f} slots/s, {dt*1e3:.2f} ms per step", flush=True)
