# HBM traffic of the bench's decoder launch (fused float64 on-chip kernel): FETCH_SIZE and WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md, HBM section) -> gpurun_out/r2_decoder_traffic.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_r2_fetch -- python3 $R/bench.py --no-cpu --no-fast --no-allrows --steps 1 --warmup 1 > $R/gpurun_out/pmc_r2_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_r2_write -- python3 $R/bench.py --no-cpu --no-fast --no-allrows --steps 1 --warmup 1 > $R/gpurun_out/pmc_r2_write.log 2>&1
python3 - <<PY
import csv, glob, json, collections
def per_kernel(pat, name):
    acc = collections.defaultdict(list)
    for f in glob.glob(pat):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name:
                acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
fe = per_kernel("$R/gpurun_out/pmc_r2_fetch/*/*counter_collection.csv", 'FETCH_SIZE')
wr = per_kernel("$R/gpurun_out/pmc_r2_write/*/*counter_collection.csv", 'WRITE_SIZE')
rows = []
for k in sorted(fe, key=lambda k: -fe[k][0]):
    rows.append(dict(kernel=k[-70:], launches=fe[k][1], FETCH_SIZE_KB=round(fe[k][0], 1), WRITE_SIZE_KB=round(wr.get(k, (0, 0))[0], 1)))
dec = [r for r in rows if 'chip64' in r['kernel']]
out = dict(batch_slots=256, per_kernel=rows[:30])
if dec:
    out.update(kernel=dec[0]['kernel'], rows=15, FETCH_SIZE_KB_per_launch=dec[0]['FETCH_SIZE_KB'], WRITE_SIZE_KB_per_launch=dec[0]['WRITE_SIZE_KB'], fetch_correction=2.0)
json.dump(out, open("$R/gpurun_out/r2_decoder_traffic.json", 'w'), indent=1)
for r in rows[:14]: print(r)
tot_f = sum(r['FETCH_SIZE_KB'] * r['launches'] for r in rows) ; tot_w = sum(r['WRITE_SIZE_KB'] * r['launches'] for r in rows)
print('all kernels of the 2 steps: FETCH', tot_f / 1e6, 'GB (uncorrected)  WRITE', tot_w / 1e6, 'GB')
PY
