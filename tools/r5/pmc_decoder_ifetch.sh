# Round 5: instruction-fetch / stall counters of the fixed-schedule decoder launch, in-tree library against the wave-specialised
# variant (exp_libs/libnrx_wspec.so: tools/build_variant.sh wspec -DNRX_DEC3_WSPEC=1) -> gpurun_out/r5/decoder_ifetch.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
cd $R
for V in intree wspec; do
  [ $V = wspec ] && export NRX_LIB=$R/exp_libs/libnrx_wspec.so || unset NRX_LIB
  timeout -k 10 400 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/r5/pmc_if_$V -- $PY $R/tools/r5/cert_steps.py --fixed --steps 1 --warmup 1 > $R/gpurun_out/r5/pmc_if_$V.log 2>&1
done
unset NRX_LIB
$PY - <<PY
import csv, glob, json, collections
out = {}
for v in ('intree', 'wspec'):
    acc = collections.defaultdict(list)
    for f in glob.glob("$R/gpurun_out/r5/pmc_if_%s/*/*counter_collection.csv" % v):
        for r in csv.DictReader(open(f)):
            if 'chip64' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    out[v] = {k: sum(x) / len(x) for k, x in acc.items()}
json.dump(out, open("$R/gpurun_out/r5/decoder_ifetch.json", 'w'), indent=1)
for k in sorted(out['intree']): print(f"{k:22s} {out['intree'][k]:.4g}  {out['wspec'].get(k, float('nan')):.4g}")
PY
rm -rf $R/gpurun_out/r5/pmc_if_intree $R/gpurun_out/r5/pmc_if_wspec
