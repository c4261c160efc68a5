# PMC passes over the on-chip float64 decoder (tools/archive/bench_decoder_f64.py, 4608 code blocks, 15 rows) -> gpurun_out/pmc_dec64_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_dec64_$n -- python3 $R/tools/archive/bench_decoder_f64.py 4608 15 > $R/gpurun_out/pmc_dec64_$n.log 2>&1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob("$R/gpurun_out/pmc_dec64_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'chip64' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
disp = max(cnt.values()) if cnt else 1
for k in sorted(tot): print(f"{k:28s} {tot[k]/max(cnt[k],1):16.1f}  (x{cnt[k]})")
PY
