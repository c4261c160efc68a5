# PMC look at every kernel of the float64 step (two passes of SQ counters); raw CSVs -> gpurun_out/pmc_fe2_{a,b}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--no-cpu --no-fast --no-allrows --steps 1 --warmup 1 --batch 64"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_fe2_a -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_fe2_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_fe2_b -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_fe2_b.log 2>&1
ls $R/gpurun_out/pmc_fe2_a/*/ $R/gpurun_out/pmc_fe2_b/*/ | head
