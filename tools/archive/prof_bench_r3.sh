# rocprofv3 kernel trace of the default (float64) bench step; summary -> gpurun_out/prof_$1_kernel_stats.csv
# (the interpreter is resolved to its real path first: no exec hop after the profiler's preload has initialised the GPU;
#  --no-cpu is mandatory under the profiler: the CPU leg starts a process pool)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r3a}
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$T -- $PY $R/bench.py --no-cpu --no-fast --no-allrows --no-twopass --steps 3 --warmup 1 > $R/gpurun_out/prof_$T.log 2>&1
tail -1 $R/gpurun_out/prof_$T.log | cut -c1-300
f=$(ls $R/gpurun_out/prof_$T/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/prof_${T}_kernel_stats.csv
head -32 $f | cut -c1-160
