# HBM traffic of the certified schedule's launches (stage kernels with the certificate in their tail, the final continuation) and of
# the fixed-schedule decoder: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section) -> gpurun_out/r4/r4_cert_traffic.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
cd $R
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r4/pmc_fetch -- $PY $R/tools/r4/cert_gpu_check.py --snr 31 --slots 256 --batches 1 --stages 8 16 > $R/gpurun_out/r4/pmc_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r4/pmc_write -- $PY $R/tools/r4/cert_gpu_check.py --snr 31 --slots 256 --batches 1 --stages 8 16 > $R/gpurun_out/r4/pmc_write.log 2>&1
$PY - <<PY
import csv, glob, json, collections, re
def per_kernel(pat, name):
    acc = collections.defaultdict(list)
    for f in glob.glob(pat):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name and 'chip64' in r['Kernel_Name']:
                m = re.search(r'chip64_kernel<([^>]*)>', r['Kernel_Name'])
                acc[m.group(1) if m else r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
fe = per_kernel("$R/gpurun_out/r4/pmc_fetch/*/*counter_collection.csv", 'FETCH_SIZE')
wr = per_kernel("$R/gpurun_out/r4/pmc_write/*/*counter_collection.csv", 'WRITE_SIZE')
rows = [dict(kernel='ldpc_dec_chip64_kernel<' + k + '>', launches=fe[k][1], read_GB=round(2 * fe[k][0] * 1024 / 1e9, 3), written_GB=round(wr.get(k, (0, 0))[0] * 1024 / 1e9, 3))
        for k in sorted(fe)]
out = dict(batch_slots=256, snr_db=31.0, stages=[8, 16], per_kernel=rows,
           template_arguments="<BG, Zc index, rows, fused, blocks per workgroup, MODE, rows in registers>: MODE 0 = the fixed 50-iteration schedule, "
                              "5 = stage from the LLRs + certificate, 7 = continued stage + certificate, 2 = final continuation",
           note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of tools/r4/cert_gpu_check.py (one 256-slot batch through the fixed and the "
                "certified schedule); FETCH_SIZE doubled (gfx950 tallies a 128-byte request at 64 bytes, calibrated in round 2)")
json.dump(out, open("$R/gpurun_out/r4/r4_cert_traffic.json", 'w'), indent=1)
for r in rows: print(r)
PY
