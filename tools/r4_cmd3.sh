set -e
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r4/prof_cert -o cert -- python tools/cert_gpu_check.py --snr 31 --slots 64 --batches 3 --stages 8 16 > gpurun_out/r4/prof_cert.log 2>&1
tail -3 gpurun_out/r4/prof_cert.log
find gpurun_out/r4/prof_cert -name "*kernel_stats*" | head
