# Round 5: kernel-trace timeline of the bench's certified leg (8 timed steps at 31 dB) and of the fixed schedule -> gpurun_out/r5/
#   bash tools/r5/trace_cert.sh [TAG]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-a}
mkdir -p $R/gpurun_out/r5
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
cd $R
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5/trace_cert_$T -- $PY $R/tools/r5/cert_steps.py > $R/gpurun_out/r5/trace_cert_$T.log 2>&1
grep -v simple_timer $R/gpurun_out/r5/trace_cert_$T.log | tail -2 | cut -c1-300
f=$(ls $R/gpurun_out/r5/trace_cert_$T/*/*kernel_trace.csv | head -1)
$PY tools/r5/timeline.py $f --steps 8 --json $R/gpurun_out/r5/timeline_cert_$T.json > $R/gpurun_out/r5/timeline_cert_$T.txt
cp $(ls $R/gpurun_out/r5/trace_cert_$T/*/*kernel_stats.csv | head -1) $R/gpurun_out/r5/cert_steps_${T}_kernel_stats.csv
rm -rf $R/gpurun_out/r5/trace_cert_$T
head -40 $R/gpurun_out/r5/timeline_cert_$T.txt
