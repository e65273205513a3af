#!/bin/bash
# Runs on the GPU box: rocprofv3 counter passes over bench.py (counters in their OWN runs, no tracing beside them):
# FETCH_SIZE, WRITE_SIZE -> HBM bytes per kernel launch (profiles/summarize_pmc.py), SQ counters for the big kernels.
# usage: pmc_round.sh <tag> [config] [reads] [read_len]     (default: config 2 = the headline 100 M x 150 bp job)
set -u
tag=${1:-pmc}
cfg=${2:-2}
reads=${3:-100000000}
rlen=${4:-150}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
B="--config $cfg --steps 3 --warmup 1 --cpu-sample 0 --single-shots 0 --e2e-reads 0"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --pmc $c --output-format csv -d $out/rp_$c -o r -- python3 $GRAFT_REPO_ROOT/bench.py $B > $out/bench_$c.json 2> $out/$c.err
  f=$(find $out/rp_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmc_$c.csv; rm -rf $out/rp_$c
done
timeout 420 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/rp_sq -o r -- python3 $GRAFT_REPO_ROOT/bench.py $B > $out/bench_sq.json 2> $out/sq.err
f=$(find $out/rp_sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmc_sq.csv; rm -rf $out/rp_sq
cd $GRAFT_REPO_ROOT
python3 profiles/summarize_pmc.py $out/pmc_FETCH_SIZE.csv $out/pmc_WRITE_SIZE.csv $out/pmc_traffic.json $reads $rlen $out/pmc_sq.csv | head -120
