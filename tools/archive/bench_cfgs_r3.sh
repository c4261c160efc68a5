# Throughput of the other BASELINE.json configurations, machine-readable -> gpurun_out/r3_cfg{1,2,3,4,5}.json
# (cfg2 / cfg3: tools/archive/bench_configs.py, exact float64 chain; cfg4: polar SCL + PDCCH blind decoding; cfg5: batched HARQ-IR)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python3 $R/tools/archive/bench_configs.py --steps 3 2> /dev/null | grep '^{' > $O/r3_cfg23.jsonl
grep '"cfg1"' $O/r3_cfg23.jsonl > $O/r3_cfg1.json
grep '"cfg2"' $O/r3_cfg23.jsonl > $O/r3_cfg2.json
grep '"cfg3' $O/r3_cfg23.jsonl > $O/r3_cfg3.json
python3 $R/tests/tools/bench_polar.py 2> /dev/null | grep '^{' > $O/r3_cfg4.json
python3 $R/tools/archive/bench_pdcch.py 2> /dev/null | grep '^{' >> $O/r3_cfg4.json
python3 $R/tools/r4/bench_harq.py --decoder f64 2> /dev/null | grep '^{' > $O/r3_cfg5.json
python3 $R/tools/r4/bench_harq.py --decoder f32 2> /dev/null | grep '^{' >> $O/r3_cfg5.json
cat $O/r3_cfg1.json $O/r3_cfg2.json $O/r3_cfg3.json $O/r3_cfg4.json $O/r3_cfg5.json | cut -c1-400
