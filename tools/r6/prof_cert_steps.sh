# Round 6: rocprofv3 kernel stats of the certified schedule's bench steps (3 timed steps of 256 slots, one persistent launch per step)
#   bash tools/r6/prof_cert_steps.sh TAG [extra cert_steps.py arguments]   -> gpurun_out/r6/steps_TAG_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1; shift
mkdir -p $R/gpurun_out/r6
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6/prof_$T -- $PY $R/tools/r5/cert_steps.py --persistent --steps 3 --warmup 1 "$@" > $R/gpurun_out/r6/prof_$T.log 2>&1
grep -v simple_timer $R/gpurun_out/r6/prof_$T.log | tail -1 | cut -c1-300
cp $(ls $R/gpurun_out/r6/prof_$T/*/*kernel_stats.csv | head -1) $R/gpurun_out/r6/steps_${T}_kernel_stats.csv
rm -rf $R/gpurun_out/r6/prof_$T
$PY - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/r6/steps_${T}_kernel_stats.csv")))
for r in rows[:10]:
    print(f"{float(r['AverageNs'])/1e6:9.3f} ms x{r['Calls']:>3}  {r['Name'][:110]}")
PY
