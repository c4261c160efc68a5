# Round 5: NRX_DEC3_PREFETCH (reads ahead of the barrier): parity, then fixed-schedule steps A/B against
#   exp_libs/libnrx_pf0.so = tools/build_variant.sh pf0 -DNRX_DEC3_PREFETCH=0   (the in-tree library built with -DNRX_DEC3_PREFETCH=1 for this comparison)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5
cd $R
python -m pytest tests/test_gpu_ldpc.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r5/pf_tests.log 2>&1 || { tail -20 gpurun_out/r5/pf_tests.log; exit 1; }
tail -2 gpurun_out/r5/pf_tests.log
rm -f gpurun_out/r5/pf_steps.log
for rep in 1 2 3; do
  for lib in neoradium_amd/libnrx.so exp_libs/libnrx_pf0.so; do
    echo "lib=$lib" >> gpurun_out/r5/pf_steps.log
    NRX_LIB=$R/$lib python tools/r5/cert_steps.py --fixed --steps 8 --warmup 2 >> gpurun_out/r5/pf_steps.log 2>&1 || exit 1
  done
done
grep -v amdgpu.ids gpurun_out/r5/pf_steps.log
