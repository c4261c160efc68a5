# Round 5: non-temporal streaming accesses in the decoder (NRX_DEC3_NT / NRX_CERT_NT): parity, step times A/B, HBM traffic of the certified schedule.
#   exp_libs/libnrx_nt0.so = tools/build_variant.sh nt0 -DNRX_DEC3_NT=0 -DNRX_CERT_NT=0 (the stage kernels without the hint; the standalone
#   certificate kernel keeps whatever the in-tree build has)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5
cd $R
python -m pytest tests/test_gpu_ldpc.py tests/test_gpu_cert.py -m gpu -x -q > gpurun_out/r5/nt_tests.log 2>&1 || { tail -20 gpurun_out/r5/nt_tests.log; exit 1; }
tail -2 gpurun_out/r5/nt_tests.log
for rep in 1 2; do
  for lib in neoradium_amd/libnrx.so exp_libs/libnrx_nt0.so; do
    for f in "" --fixed; do
      echo "lib=$lib $f" >> gpurun_out/r5/nt_steps.log
      NRX_LIB=$R/$lib python tools/r5/cert_steps.py $f --steps 8 --warmup 2 >> gpurun_out/r5/nt_steps.log 2>&1 || exit 1
    done
  done
done
cat gpurun_out/r5/nt_steps.log
bash tools/r5/pmc_traffic.sh cert
