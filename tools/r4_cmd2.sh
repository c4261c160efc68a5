timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_classes.py -x -q 2>&1 | tail -5
