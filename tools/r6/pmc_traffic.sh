# Round 6: tools/r5/pmc_traffic.sh with a third mode -- bash tools/r6/pmc_traffic.sh pers: the certified schedule as ONE persistent launch ->
#   gpurun_out/r5/r5_cert_persistent_traffic.json   (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes; fixed | cert as in round 5)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:-fixed}
mkdir -p $R/gpurun_out/r5
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
cd $R
FLAG=$([ "$MODE" = fixed ] && echo --fixed); [ "$MODE" = pers ] && FLAG=--persistent
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/r5/pmc_${MODE}_$C -- $PY $R/tools/r5/cert_steps.py $FLAG --steps 1 --warmup 1 > $R/gpurun_out/r5/pmc_${MODE}_$C.log 2>&1
done
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
fe = per_kernel("$R/gpurun_out/r5/pmc_${MODE}_FETCH_SIZE/*/*counter_collection.csv", 'FETCH_SIZE')
wr = per_kernel("$R/gpurun_out/r5/pmc_${MODE}_WRITE_SIZE/*/*counter_collection.csv", 'WRITE_SIZE')
rows = []
for k in sorted(fe, key=lambda k: -(2 * fe[k][0] + wr.get(k, (0, 0))[0])):
    rows.append(dict(kernel=k[-80:], launches=fe[k][1], FETCH_SIZE_KB=round(fe[k][0], 1), WRITE_SIZE_KB=round(wr.get(k, (0, 0))[0], 1),
                     read_GB=round(2 * fe[k][0] * 1024 / 1e9, 3), written_GB=round(wr.get(k, (0, 0))[0] * 1024 / 1e9, 3)))
out = dict(batch_slots=256, schedule="$MODE", per_kernel=rows[:32],
           note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of tools/r5/cert_steps.py --steps 1 --warmup 1 (256 slots per launch, "
                "tools/r5/pmc_traffic.sh); read_GB = 2 x FETCH_SIZE (gfx950 tallies each 128-byte read request at 64 bytes, MI355X_MICROARCH.md HBM section; "
                "calibrated in round 2 on the decoder's 8-byte-per-lane loads), averaged over the launches of the run (warm-up + timed)")
dec = [r for r in rows if 'chip64' in r['kernel']]
if "$MODE" == "fixed" and dec:
    out.update(kernel=dec[0]['kernel'], rows=15, FETCH_SIZE_KB_per_launch=dec[0]['FETCH_SIZE_KB'], WRITE_SIZE_KB_per_launch=dec[0]['WRITE_SIZE_KB'],
               fetch_correction=2.0, algorithmic_bytes_per_launch={'read': 256 * 72 * 13104 * 8, 'write': 256 * 72 * 8400 + 256 * 72})
json.dump(out, open("$R/gpurun_out/r5/r5_" + ("decoder" if "$MODE" == "fixed" else ("cert_persistent" if "$MODE" == "pers" else "cert")) + "_traffic.json", 'w'), indent=1)
for r in rows[:12]: print(r)
PY
rm -rf $R/gpurun_out/r5/pmc_${MODE}_FETCH_SIZE $R/gpurun_out/r5/pmc_${MODE}_WRITE_SIZE
