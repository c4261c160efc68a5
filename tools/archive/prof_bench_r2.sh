# rocprofv3 kernel trace of the default (float64) bench step; summary -> gpurun_out/prof_$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r2a}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$T -- python3 $R/bench.py --no-cpu --no-fast --no-allrows --steps 3 --warmup 1 > $R/gpurun_out/prof_$T.log 2>&1
tail -1 $R/gpurun_out/prof_$T.log | cut -c1-300
f=$(ls $R/gpurun_out/prof_$T/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/prof_${T}_kernel_stats.csv
head -40 $f | cut -c1-200
