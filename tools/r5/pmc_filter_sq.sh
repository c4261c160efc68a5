# Round 5: SQ counters of the overlap-save channel filter (tools/r5/bench_filter.py) -> gpurun_out/r5/filter_sq.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
cd $R
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/r5/pmc_fsq$i -- $PY $R/tools/r5/bench_filter.py --reps 2 > $R/gpurun_out/r5/pmc_fsq$i.log 2>&1
done
$PY - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/r5/pmc_fsq*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'apply_td_os' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: sum(v) / len(v) for k, v in acc.items()}
json.dump(out, open("$R/gpurun_out/r5/filter_sq.json", 'w'), indent=1)
for k, v in sorted(out.items()): print(f"{k:28s} {v:.4g}")
PY
rm -rf $R/gpurun_out/r5/pmc_fsq1 $R/gpurun_out/r5/pmc_fsq2
