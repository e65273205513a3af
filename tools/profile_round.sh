#!/bin/bash
# Runs on the GPU box: bench.py (full, with the CPU baseline), then the same command under rocprofv3
# --kernel-trace --stats; leaves bench JSON + kernel stats CSV under gpurun_out/$1/.
set -u
tag=${1:-prof}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 400 python bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rp -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --cpu-sample 0 --e2e-reads 0 > $out/prof_bench.json 2> $out/prof.err
f=$(find $out/rp -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv
rm -rf $out/rp
ls -la $out
