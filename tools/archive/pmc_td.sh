# PMC passes over the time-domain channel filter (tools/archive/bench_td.py, 256 slots) -> gpurun_out/pmc_td_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_td_$n -- $PY $R/tools/archive/bench_td.py 256 > $R/gpurun_out/pmc_td_$n.log 2>&1
done
$PY - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob("$R/gpurun_out/pmc_td_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'apply_td_paths4' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
for k in sorted(tot): print(f"{k:28s} {tot[k]/max(cnt[k],1):18.1f}  (x{cnt[k]})")
PY
