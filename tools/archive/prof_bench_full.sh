cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1h -- python3 $R/bench.py --no-cpu --no-exact --steps 3 --warmup 1 > $R/gpurun_out/prof_r1h.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_r1h_fetch -- python3 $R/bench.py --no-cpu --no-exact --steps 1 --warmup 1 > $R/gpurun_out/pmc_r1h_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_r1h_write -- python3 $R/bench.py --no-cpu --no-exact --steps 1 --warmup 1 > $R/gpurun_out/pmc_r1h_write.log 2>&1
tail -1 $R/gpurun_out/prof_r1h.log | cut -c1-300
