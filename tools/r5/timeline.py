"""Round-5 tool (CPU): account for every microsecond of a traced run.  Reads a rocprofv3 `*_kernel_trace.csv`, orders the dispatches
by start time, splits them into steps at a marker kernel (default: random_bits_kernel = first launch of a PdschLink step) and prints,
for the LAST `--steps` steps: busy time per kernel, the gaps (GPU idle between the end of one dispatch and the start of the next) with
the kernels either side, and the step's wall time (first start of this step -> first start of the next).

    python tools/r5/timeline.py gpurun_out/r5/trace/*/*_kernel_trace.csv [--steps 8] [--marker random_bits] [--json out.json]
"""
import argparse
import collections
import csv
import json
import re


def short(name):
    m = re.search(r'chip64_kernel<([^>]*)>', name)
    if m:
        return 'chip64<' + m.group(1).replace(' ', '') + '>'
    m = re.search(r'([A-Za-z_0-9]+)(<[^(]*)?\(', name)
    n = m.group(1) if m else name
    if 'at::native' in name:
        n = 'torch:' + (re.search(r'(\w+Functor|\w+_kernel_cuda|arange|index_select|scatter_gather|reduce_kernel)', name) or [None, name[:30]])[1]
    return n[:48]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--marker', default='random_bits')
    ap.add_argument('--json', default='')
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.csv)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if a.marker in r[2]]
    starts = starts[-(a.steps + 1):] if len(starts) > a.steps else starts
    steps = []
    for si in range(len(starts) - 1):
        seg = rows[starts[si]:starts[si + 1]]
        wall = rows[starts[si + 1]][0] - seg[0][0]
        busy = collections.OrderedDict()
        gaps = []
        end = seg[0][0]
        for (s, e, n) in seg:
            busy[n] = busy.get(n, 0) + (e - s)
            if s > end:
                gaps.append((s - end, prev, n))
            end, prev = max(end, e), n
        tail = rows[starts[si + 1]][0] - end
        if tail > 0:
            gaps.append((tail, prev, '(next step)'))
        steps.append(dict(wall_ms=wall / 1e6, busy_ms=sum(busy.values()) / 1e6, idle_ms=sum(g[0] for g in gaps) / 1e6, n_dispatch=len(seg),
                          busy={k: v / 1e6 for k, v in busy.items()}, gaps=sorted(gaps, reverse=True)[:8]))
    n = max(len(steps), 1)
    print(f"{len(steps)} steps: wall {sum(s['wall_ms'] for s in steps) / n:.3f} ms, busy {sum(s['busy_ms'] for s in steps) / n:.3f} ms, "
          f"idle {sum(s['idle_ms'] for s in steps) / n:.3f} ms, dispatches {sum(s['n_dispatch'] for s in steps) / n:.0f}")
    tot = collections.OrderedDict()
    for s in steps:
        for k, v in s['busy'].items():
            tot[k] = tot.get(k, 0) + v / n
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f"  {v:8.3f} ms  {k}")
    for i, s in enumerate(steps):
        print(f"step {i}: wall {s['wall_ms']:.3f} busy {s['busy_ms']:.3f} idle {s['idle_ms']:.3f}; largest gaps (us): "
              + ', '.join(f"{g[0] / 1e3:.0f} [{g[1]} -> {g[2]}]" for g in s['gaps'][:5]))
    if a.json:
        json.dump(dict(steps=steps, mean_busy_ms=tot), open(a.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
