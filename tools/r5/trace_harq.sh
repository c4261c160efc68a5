# Round 5: kernel-trace timeline of bench.py --config cfg5 (HARQ-IR rounds) -> gpurun_out/r5/timeline_harq.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
cd $R
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5/trace_harq -- $PY $R/bench.py --config cfg5 --steps 6 --warmup 2 > $R/gpurun_out/r5/trace_harq.log 2>&1
grep -v simple_timer $R/gpurun_out/r5/trace_harq.log | tail -1 | cut -c1-400
f=$(ls $R/gpurun_out/r5/trace_harq/*/*kernel_trace.csv | head -1)
$PY tools/r5/timeline.py $f --steps 6 --marker random_bits --json $R/gpurun_out/r5/timeline_harq.json > $R/gpurun_out/r5/timeline_harq.txt
cp $(ls $R/gpurun_out/r5/trace_harq/*/*kernel_stats.csv | head -1) $R/gpurun_out/r5/harq_round_kernel_stats.csv
rm -rf $R/gpurun_out/r5/trace_harq
head -45 $R/gpurun_out/r5/timeline_harq.txt
