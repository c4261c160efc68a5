set -e
mkdir -p gpurun_out/r4
python tools/llr_stats.py --snr 29 31 33 35 --slots 4 --keep 96 --out gpurun_out/r4/llr_sample.npz > gpurun_out/r4/llr_stats.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench/occ_test.hip -o /tmp/occ_test && /tmp/occ_test > gpurun_out/r4/occ_test.txt 2>&1
cat gpurun_out/r4/llr_stats.log gpurun_out/r4/occ_test.txt
