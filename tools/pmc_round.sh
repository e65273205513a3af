#!/bin/bash
# Runs on the GPU box: rocprofv3 counter passes over bench.py (counters in their OWN runs, no tracing beside them):
# FETCH_SIZE, WRITE_SIZE -> HBM bytes per kernel launch (profiles/summarize_pmc.py), SQ counters for the three big kernels.
set -u
tag=${1:-pmc}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/rp_$c -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $out/bench_$c.json 2> $out/$c.err
  f=$(find $out/rp_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmc_$c.csv; rm -rf $out/rp_$c
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/rp_sq -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $out/bench_sq.json 2> $out/sq.err
f=$(find $out/rp_sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmc_sq.csv; rm -rf $out/rp_sq
cd $GRAFT_REPO_ROOT
python3 profiles/summarize_pmc.py $out/pmc_FETCH_SIZE.csv $out/pmc_WRITE_SIZE.csv $out/pmc_traffic.json 10000000 150 $out/pmc_sq.csv | head -80
python3 - $out/pmc_sq.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("crass::", "")
    if any(x in k for x in ("k_filter_fast", "k_survivor_lanes", "k_anchor_filter_dev", "k_dm_verify")):
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
