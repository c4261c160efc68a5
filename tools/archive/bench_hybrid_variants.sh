for v in hyb_12_2 hyb_8_3 hyb_10_3; do echo $v; NRX_LIB=$GRAFT_REPO_ROOT/exp_libs/libnrx_$v.so timeout -k 10 200 python tools/archive/bench_hybrid.py 2>&1 | grep rows.*46 || exit 1; done
