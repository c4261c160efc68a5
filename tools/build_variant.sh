#!/bin/bash
# Developer tool: build a side library exp_libs/libnrx_<name>.so in which ONE translation unit is compiled with extra flags
# (default: the on-chip float64 decoder, nrx_ldpc_dec3.hip; NRX_VARIANT_UNIT=nrx_chan etc. picks another, NRX_VARIANT_SRC another
# source file for it); every other object comes from the in-tree build.  Load it with NRX_LIB=exp_libs/libnrx_<name>.so.
#   usage: tools/build_variant.sh <name> [flags ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/exp_libs
FLAGS="-std=c++20 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sdwa-peephole=0 -fPIC -Wno-comment -Wno-unused-value"
UNIT=${NRX_VARIANT_UNIT:-nrx_ldpc_dec3}
SRC=${NRX_VARIANT_SRC:-$R/neoradium_amd/csrc/$UNIT.hip}
/opt/rocm/bin/hipcc $FLAGS "$@" -I$R/neoradium_amd/csrc -c $SRC -o $R/exp_libs/${UNIT}_$name.o
objs=$(ls $R/neoradium_amd/csrc/obj/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/exp_libs/${UNIT}_$name.o -o $R/exp_libs/libnrx_$name.so
echo $R/exp_libs/libnrx_$name.so
