# round 4 evidence run: kernel stats of the default float64 bench step and of the certified schedule, the certificate check across
# the waterfall (and broken on purpose), all under gpurun_out/r4/ (copy what is to be judged into profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/prof_bench -- $PY $R/bench.py --no-cpu --no-fast --no-allrows --no-twopass --no-cert --no-configs --steps 3 --warmup 1 > $R/gpurun_out/r4/prof_bench.log 2>&1
cp $(ls -t $R/gpurun_out/r4/prof_bench/*/*kernel_stats.csv | head -1) $R/gpurun_out/r4/r4_v2_bench_b256_f64_kernel_stats.csv
head -6 $R/gpurun_out/r4/r4_v2_bench_b256_f64_kernel_stats.csv | cut -c1-150
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/prof_certb -- $PY $R/tools/r4/cert_gpu_check.py --snr 31 --slots 256 --batches 3 --stages 8 16 > $R/gpurun_out/r4/prof_certb.log 2>&1
cp $(ls -t $R/gpurun_out/r4/prof_certb/*/*kernel_stats.csv | head -1) $R/gpurun_out/r4/r4_cert_inkernel_b256_kernel_stats.csv
head -8 $R/gpurun_out/r4/r4_cert_inkernel_b256_kernel_stats.csv | cut -c1-150
timeout -k 10 500 $PY tools/r4/cert_gpu_check.py --snr 28 29 30 31 32 33 34 35 --slots 256 --batches 2 --stages 8 16 --json gpurun_out/r4/r4_cert_check_waterfall_inkernel.json 2>/dev/null | cut -c1-330
timeout -k 10 200 $PY tools/r4/cert_gpu_check.py --snr 30 31 --slots 128 --stages 8 16 --flags 7 --json gpurun_out/r4/r4_cert_check_broken_inkernel.json 2>/dev/null | cut -c1-330
timeout -k 10 1100 $PY bench.py > gpurun_out/r4/bench_run3.json 2> gpurun_out/r4/bench_run3.err
tail -c 600 gpurun_out/r4/bench_run3.json
