# Round 5: fixed-schedule steps of the in-tree library against exp_libs/libnrx_$1.so (tools/build_variant.sh), three alternating runs each,
# after the decoder's parity tests.   bash tools/r5/ab_steps.sh VARIANT
R=$GRAFT_REPO_ROOT
V=$1
mkdir -p $R/gpurun_out/r5
cd $R
python -m pytest tests/test_gpu_ldpc.py -m gpu -x -q > gpurun_out/r5/ab_tests.log 2>&1 || { tail -20 gpurun_out/r5/ab_tests.log; exit 1; }
tail -1 gpurun_out/r5/ab_tests.log
rm -f gpurun_out/r5/ab_$V.log
for rep in 1 2 3; do
  for lib in neoradium_amd/libnrx.so exp_libs/libnrx_$V.so; do
    echo "lib=$lib" >> gpurun_out/r5/ab_$V.log
    NRX_LIB=$R/$lib python tools/r5/cert_steps.py --fixed --steps 8 --warmup 2 >> gpurun_out/r5/ab_$V.log 2>&1 || exit 1
  done
done
grep -v amdgpu.ids gpurun_out/r5/ab_$V.log | cut -c1-150
