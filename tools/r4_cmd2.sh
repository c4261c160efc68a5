set -e
mkdir -p gpurun_out/r4
timeout -k 10 300 python tools/cert_gpu_check.py --snr 31 --slots 32 --stages 8 16 > gpurun_out/r4/cert_check_a.log 2>&1
cat gpurun_out/r4/cert_check_a.log
bash tools/r4_prof.sh cert3 tools/cert_gpu_check.py --snr 31 --slots 64 --batches 3 --stages 8 16 | grep -E "certify|chip64|^31" 
