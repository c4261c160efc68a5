cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1g -- python3 $R/bench.py --no-cpu --no-exact --steps 3 --warmup 1 > $R/gpurun_out/prof_r1g.log 2>&1
tail -1 $R/gpurun_out/prof_r1g.log | cut -c1-200
