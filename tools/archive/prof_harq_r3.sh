# rocprofv3 kernel statistics of the batched HARQ-IR loop (cfg5, float64) -> gpurun_out/prof_harq_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_harq -- $PY $R/tools/r4/bench_harq.py --decoder f64 > $R/gpurun_out/prof_harq.log 2>&1
f=$(ls $R/gpurun_out/prof_harq/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/prof_harq_kernel_stats.csv
