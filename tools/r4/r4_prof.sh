# rocprofv3 kernel trace of a python tool of this repo; summary -> gpurun_out/r4/prof_$TAG_kernel_stats.csv
#   bash tools/r4/r4_prof.sh TAG script.py args...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1; shift
S=$1; shift
mkdir -p $R/gpurun_out/r4
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
cd $R
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/prof_$T -- $PY $R/$S "$@" > $R/gpurun_out/r4/prof_$T.log 2>&1
grep -v simple_timer $R/gpurun_out/r4/prof_$T.log | tail -4 | cut -c1-400
f=$(ls $R/gpurun_out/r4/prof_$T/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/r4/prof_${T}_kernel_stats.csv
head -24 $f | cut -c1-200
