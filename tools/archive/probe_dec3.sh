# GPU side of tools/archive/probe_dec3.py (the side library is built in the container: tools/build_variant.sh probe -DNRX_DEC3_PROBE)
R=$GRAFT_REPO_ROOT
NRX_LIB=$R/exp_libs/libnrx_probe.so python3 $R/tools/archive/probe_dec3.py 18432 15 > $R/gpurun_out/r3_probe_dec3.txt 2>&1
cat $R/gpurun_out/r3_probe_dec3.txt
