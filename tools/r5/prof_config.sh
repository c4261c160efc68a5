# Round 5: rocprofv3 kernel stats of `bench.py --config CFG --steps 3 --warmup 1` -> gpurun_out/r5/CFG_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=${1:-cfg3}
mkdir -p $R/gpurun_out/r5
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
cd $R
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5/prof_$C -- $PY $R/bench.py --config $C --steps 3 --warmup 1 > $R/gpurun_out/r5/prof_$C.log 2>&1
grep -v simple_timer $R/gpurun_out/r5/prof_$C.log | tail -1 | cut -c1-200
cp $(ls $R/gpurun_out/r5/prof_$C/*/*kernel_stats.csv | head -1) $R/gpurun_out/r5/${C}_kernel_stats.csv
rm -rf $R/gpurun_out/r5/prof_$C
$PY - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/r5/${C}_kernel_stats.csv")))
for r in rows[:18]:
    print(f"{float(r['TotalDurationNs'])/1e6/4:9.3f} ms/step x{int(r['Calls'])/4:5.1f}  {r['Name'][:110]}")
PY
