timeout -k 10 800 python -m pytest tests/test_gpu_cert.py -x -q > gpurun_out/r4/test_gpu_cert.log 2>&1
tail -30 gpurun_out/r4/test_gpu_cert.log
