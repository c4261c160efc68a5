timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
