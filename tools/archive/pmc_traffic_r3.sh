# HBM traffic of the bench's decoder launch (fused float64 on-chip kernel): FETCH_SIZE and WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md, HBM section) -> gpurun_out/r3_decoder_traffic.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_r3_fetch -- $PY $R/bench.py --no-cpu --no-fast --no-allrows --no-twopass --steps 1 --warmup 1 > $R/gpurun_out/pmc_r3_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_r3_write -- $PY $R/bench.py --no-cpu --no-fast --no-allrows --no-twopass --steps 1 --warmup 1 > $R/gpurun_out/pmc_r3_write.log 2>&1
$PY - <<PY
import csv, glob, json, collections, re
def per_kernel(pat, name):
    acc = collections.defaultdict(list)
    for f in glob.glob(pat):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name:
                m = re.search(r'(\\w+_kernel(?:<[^(]*>)?)', r['Kernel_Name'])
                acc[m.group(1) if m else r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
fe = per_kernel("$R/gpurun_out/pmc_r3_fetch/*/*counter_collection.csv", 'FETCH_SIZE')
wr = per_kernel("$R/gpurun_out/pmc_r3_write/*/*counter_collection.csv", 'WRITE_SIZE')
rows = []
for k in sorted(fe, key=lambda k: -fe[k][0]):
    rows.append(dict(kernel=k[-70:], launches=fe[k][1], FETCH_SIZE_KB=round(fe[k][0], 1), WRITE_SIZE_KB=round(wr.get(k, (0, 0))[0], 1)))
dec = [r for r in rows if 'chip64' in r['kernel']]
out = dict(batch_slots=256, per_kernel=rows[:30])
if dec:
    out.update(kernel=dec[0]['kernel'], rows=15, FETCH_SIZE_KB_per_launch=dec[0]['FETCH_SIZE_KB'], WRITE_SIZE_KB_per_launch=dec[0]['WRITE_SIZE_KB'], fetch_correction=2.0,
               algorithmic_bytes_per_launch={'read': 256 * 72 * 13104 * 8, 'write': 256 * 72 * 8400 + 256 * 72},
               note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of bench.py --steps 1 --warmup 1 (256 slots per launch, tools/archive/pmc_traffic_r3.sh); gfx950 tallies "
                    "each 128-byte read request at 64 bytes (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is doubled (calibrated in round 2 on this kernel's 8-byte-per-lane loads)")
json.dump(out, open("$R/gpurun_out/r3_decoder_traffic.json", 'w'), indent=1)
for r in rows[:10]: print(r)
PY
