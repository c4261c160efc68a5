mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4/test_gpu_all.log 2>&1
tail -15 gpurun_out/r4/test_gpu_all.log
