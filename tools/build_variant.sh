#!/bin/bash
# Developer tool: build a side library exp_libs/libnrx_<name>.so whose on-chip float64 decoder (nrx_ldpc_dec3.hip) is
# compiled with extra flags (e.g. -DNRX_DEC3_PROBE); every other object comes from the in-tree build.  Load it with
# NRX_LIB=exp_libs/libnrx_<name>.so.   usage: tools/build_variant.sh <name> [flags ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/exp_libs
FLAGS="-std=c++20 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sdwa-peephole=0 -fPIC -Wno-comment -Wno-unused-value"
SRC=${NRX_VARIANT_SRC:-$R/neoradium_amd/csrc/nrx_ldpc_dec3.hip}
/opt/rocm/bin/hipcc $FLAGS "$@" -I$R/neoradium_amd/csrc -c $SRC -o $R/exp_libs/dec3_$name.o
objs=$(ls $R/neoradium_amd/csrc/obj/*.o | grep -v nrx_ldpc_dec3.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/exp_libs/dec3_$name.o -o $R/exp_libs/libnrx_$name.so
echo $R/exp_libs/libnrx_$name.so
