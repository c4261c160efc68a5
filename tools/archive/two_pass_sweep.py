import sys, os, json, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import neoradium_amd as nr
import bench
for fp in (8, 10, 12, (8, 16), (10, 20), (8, 16, 28), (10, 16, 24, 34), (6, 10, 16, 26)):
    link = bench.build_link(nr, decoder='f64', firstPassIter=fp)
    B = 256
    for snr in (31.0,):
        link.run(0, B, snr, seed=123); link.run(B, B, snr, seed=123)
        torch.cuda.synchronize()
        c = torch.zeros(4, dtype=torch.int64, device=link.dev)
        t0 = time.perf_counter()
        for k in range(6):
            link.run((2 + k) * B, B, snr, seed=123, counters=c)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(fp, snr, round(6 * B / dt), 'slots/s', round(1e3 * dt / 6, 2), 'ms', c.cpu().tolist(), flush=True)
    del link
    torch.cuda.empty_cache()
