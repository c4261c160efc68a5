# PMC look at the front-end kernels (one pass, SQ counters)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_fe -- python3 $R/bench.py --no-cpu --no-exact --steps 1 --warmup 1 --batch 64 > $R/gpurun_out/pmc_fe.log 2>&1
ls $R/gpurun_out/pmc_fe/*/ | head
