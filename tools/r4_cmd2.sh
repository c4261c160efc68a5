timeout -k 10 900 python -m pytest tests/test_gpu_ldpc.py -x -q -k "uninitialised" 2>&1 | tail -8
timeout -k 10 600 python -m pytest tests/test_gpu_cert.py -x -q 2>&1 | tail -3
