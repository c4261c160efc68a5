# rocprofv3 kernel trace of the float32 fast mode ($1 = f64 | f32: the waveform chain); summary -> gpurun_out/prof_fast_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fast -- $PY $R/bench.py --decoder f32 --waveform ${1:-f64} --no-cpu --no-fast --no-allrows --no-twopass --steps 3 --warmup 1 > $R/gpurun_out/prof_fast.log 2>&1
f=$(ls $R/gpurun_out/prof_fast/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/prof_fast_kernel_stats.csv
