mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_cert.py tests/test_gpu_ldpc.py -x -q -k "not uninitialised" 2>&1 | tail -3
echo "== in-kernel"; timeout -k 10 300 python tools/cert_gpu_check.py --snr 31 33 --slots 256 --batches 4 --stages 8 16 2>&1 | grep -v amdgpu | cut -c1-420
echo "== stand-alone"; timeout -k 10 300 python tools/cert_gpu_check.py --snr 31 --slots 256 --batches 4 --stages 8 16 --standalone 2>&1 | grep -v amdgpu | cut -c1-420
