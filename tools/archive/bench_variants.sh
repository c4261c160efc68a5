# GPU side: time the on-chip float64 decoder of several side builds (exp_libs/libnrx_<name>.so; tools/build_variant.sh)
# usage: bash tools/archive/bench_variants.sh name1 name2 ...   (name "tree" = the in-tree library)
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = tree ]; then unset NRX_LIB; else export NRX_LIB=$R/exp_libs/libnrx_$v.so; fi
  echo "== $v" >> $R/gpurun_out/r3_variants.txt
  python3 $R/tools/archive/bench_decoder_f64.py 18432 15 2>&1 | grep iters >> $R/gpurun_out/r3_variants.txt
  NRX_BENCH_FUSED=1 python3 $R/tools/archive/bench_decoder_f64.py 18432 15 2>&1 | grep iters >> $R/gpurun_out/r3_variants.txt
done
cat $R/gpurun_out/r3_variants.txt
