cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fastw -- $PY $R/tools/tmp/fastw1.py > $R/gpurun_out/prof_fastw.log 2>&1
f=$(ls $R/gpurun_out/prof_fastw/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/prof_fastw_kernel_stats.csv
