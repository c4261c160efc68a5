timeout -k 10 900 python -m pytest tests/test_gpu_cert.py -x -q -k "class_surface" 2>&1 | tail -15
