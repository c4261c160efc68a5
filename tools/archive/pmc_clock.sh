# effective shader clock under the decoder: GRBM_GUI_ACTIVE (shader-clock cycles the GPU was busy) / kernel duration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_clock -- python3 $R/tools/archive/bench_decoder2.py 50 4608 > $R/gpurun_out/pmc_clock.log 2>&1
ls -R $R/gpurun_out/pmc_clock | head -20
