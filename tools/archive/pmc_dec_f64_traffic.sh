cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_unf_fetch -- python3 $R/tools/archive/bench_decoder_f64.py 18432 15 > $R/gpurun_out/pmc_unf_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_unf_write -- python3 $R/tools/archive/bench_decoder_f64.py 18432 15 > $R/gpurun_out/pmc_unf_write.log 2>&1
python3 - <<PY
import csv, glob
for nm, pat in (('FETCH_SIZE', "$R/gpurun_out/pmc_unf_fetch/*/*counter_collection.csv"), ('WRITE_SIZE', "$R/gpurun_out/pmc_unf_write/*/*counter_collection.csv")):
    v = [float(r['Counter_Value']) for f in glob.glob(pat) for r in csv.DictReader(open(f)) if 'chip64' in r['Kernel_Name'] and r['Counter_Name'] == nm]
    print(nm, 'KB per launch', sum(v) / len(v), 'launches', len(v))
print('algorithmic input: 18432 x 35 x 384 x 8 B =', 18432 * 35 * 384 * 8 / 1e9, 'GB; output 18432 x 8448 B =', 18432 * 8448 / 1e6, 'MB')
PY
